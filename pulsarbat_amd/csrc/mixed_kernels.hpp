// mixed_kernels.hpp -- transforms of 7-smooth length (radices 9, 7, 5, 3, 8, 4, 2) for the lengths that
// pulsarbat.utils.next_fast_len / prev_fast_len hand out (reference pulsarbat/utils.py:68-130, transforms.py:364-382).
//
// k_colmix.  A length N = N1 * N2 with N1 = P * Q any 7-smooth number runs the planar pipeline of pbhip.hip with BOTH
// column roles played by this one kernel:
//   role A (P > 1, what k_radix_p does for P <= 16): the P-point transform over n1 = Q a + b, rows one chunk = N / P apart,
//           then the twiddle W_N1^{b c};
//   role B (what k_colq does for powers of two): the Q-point transform over b inside row block c, then W_N^{n2 k1},
//           k1 = c + P d.
// The inverse direction mirrors both (conjugate twiddle first, inverse transform after).
// One tile = L rows x W columns (W a power of two: as many 8- or 16-byte elements as keep the tile within 64 KiB, at least
// 64 bytes per row, the last group of a row short when W does not divide the row length) in LDS, transformed IN PLACE by
// decimation-in-frequency stages of radix r_1, r_2, ... (each: L / r butterflies per column, twiddle W_Lj^{i u} from a W_L table
// in LDS), which leaves X[k] at the mixed-radix digit-reversed position; the stores read LDS through a permutation table, so
// rows leave in natural order.  Per-thread float64 recurrence for the inter-pass twiddle as in k_colq.  Two workgroups per
// CU; the next tile's samples are requested before the stages of the current one.
//
// k_rowmix (end of the file).  With N2 = 2^k (k >= 5) the rows belong to the power-of-two tile engine (k_row / k_rowp); with
// fewer factors of two the rows are mixed-radix as well and this kernel is the fused row pass.
#pragma once
#include "fft_core.hpp"

namespace PBH_NS {

// (kMixMaxStages: fft_core.hpp -- the plan structure of pbhip_internal.hpp needs it without this header)
constexpr int kMixMaxLen = 1024;   // rows of a tile at most (64-byte pieces beyond 512 rows of complex64)

struct MixParams {
    const cf* ld;       // loads: series s at s * ld_plane
    int64_t ld_plane;
    cf* st;             // stores: series s at s * st_plane, element index (time) - st_shift
    int64_t st_plane;
    int S, L;           // series, transform length (rows of a tile)
    int64_t rstride;    // elements between consecutive rows
    int nblock;         // row blocks per series (role B: P blocks of Q rows), block c at c * bstride
    int64_t bstride;
    int64_t ncolgrp;    // column groups (W columns each) per block; the last one is short when W does not divide ncols
    int64_t ncols;      // columns of a block (role A: N / P, role B: N2)
    int wlog2;          // W = 2^wlog2 columns per tile, L * W points within kMixTileBytes
    // twiddle exponent of (row k, column position x): ((x / xdiv) * (c * y0mul + ystep * k) % nmod) * mult, looked up in tw
    int64_t xdiv, ystep, nmod, mult;
    int y0mul;
    BigTwiddle tw;
    int nstage;
    int radix[kMixMaxStages];
    const cf* wl;                  // W_L^p = exp(-2 pi i p / L), p < L
    const unsigned short* perm;    // natural row k sits at LDS row perm[k] after the stages
    int64_t keep0, keep1, st_shift;   // only element indices in [keep0, keep1) are stored (crop of the last inverse pass)
    unsigned* counter;  // tile hand-out (zeroed before the launch): tiles beyond a workgroup's first two come in launch-wide order,
                        // so the tiles in flight across the chip stay a tight window of neighbouring column groups (k_colq)
};

// Odd small transforms through the symmetry of the roots: with s_a = v[a] + v[P-a], d_a = v[a] - v[P-a],
//   X[c], X[P-c] = (v0 + sum_a cos(2 pi a c / P) s_a)  -/+  i dir (sum_a sin(2 pi a c / P) d_a),   dir = -1 forward.
// Half the multiplications of the direct form (dft_odd, fft_core.hpp) and few values live at a time: the direct 7-point
// form alone needed more registers than the 128 this kernel has.
template <int P, int DIR>
__device__ __forceinline__ void dft_odd_sym(cf (&v)[P]) {
    constexpr int H = (P - 1) / 2;
    cf sp[H], dm[H];
    cf x0 = v[0];
#pragma unroll
    for (int a = 1; a <= H; ++a) {
        sp[a - 1] = cadd(v[a], v[P - a]);
        dm[a - 1] = csub(v[a], v[P - a]);
        x0 = cadd(x0, sp[a - 1]);
    }
    const cf v0 = v[0];
    v[0] = x0;
#pragma unroll
    for (int c = 1; c <= H; ++c) {
        cf m = v0, n = make_cf(0, 0);
#pragma unroll
        for (int a = 1; a <= H; ++a) {
            const int j = (a * c) % P;   // compile-time after unrolling
            const real wr = OddRoots<P>::c(j), wi = OddRoots<P>::s(j);
            m.x += wr * sp[a - 1].x;
            m.y += wr * sp[a - 1].y;
            n.x += wi * dm[a - 1].x;
            n.y += wi * dm[a - 1].y;
        }
        // -/+ i dir n with dir = DIR:  forward (DIR = -1): X[c] = m - i n -> (m.x + n.y, m.y - n.x)
        if (DIR < 0) {
            v[c] = make_cf(m.x + n.y, m.y - n.x);
            v[P - c] = make_cf(m.x - n.y, m.y + n.x);
        } else {
            v[c] = make_cf(m.x - n.y, m.y + n.x);
            v[P - c] = make_cf(m.x + n.y, m.y - n.x);
        }
    }
}
// Composite odd radix 9 = 3 x 3: two register levels per LDS round trip, in place (written for any R = P * P) with
// n = P n1 + n2, k = k1 + P k2: the P-point transforms over n1 leave A[k1][n2] at position P k1 + n2; times W_R^{n2 k1} (read
// from the W_L table: R divides L; one address for all lanes); the transforms over n2 leave X[k1 + P k2] at position
// P k1 + k2 -- the caller stores position p at output index (p / P) + P (p % P).
template <int P, int DIR>
__device__ __forceinline__ void dft_square(cf (&v)[P * P], const cf* wl, int L) {
    const int ts = L / (P * P);
#pragma unroll
    for (int n2 = 0; n2 < P; ++n2) {
        cf t[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) t[n1] = v[P * n1 + n2];
        dft_odd_sym<P, DIR>(t);
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            if (n2 * k1 != 0) {
                const cf w = wl[n2 * k1 * ts];
                t[k1] = cmul(t[k1], DIR < 0 ? w : cconj(w));
            }
            v[P * k1 + n2] = t[k1];
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        cf t[P];
#pragma unroll
        for (int n2 = 0; n2 < P; ++n2) t[n2] = v[P * k1 + n2];
        dft_odd_sym<P, DIR>(t);
#pragma unroll
        for (int k2 = 0; k2 < P; ++k2) v[P * k1 + k2] = t[k2];
    }
}
// MixDft<R>::run leaves output index out_index(p) at register position p
template <int R, int DIR>
struct MixDft {
    static __device__ __forceinline__ void run(cf (&v)[R], const cf*, int) { Dft<R, DIR>::run(v); }
    static constexpr int out_index(int p) { return p; }
};
template <int DIR>
struct MixDft<3, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[3], const cf*, int) { dft_odd_sym<3, DIR>(v); }
    static constexpr int out_index(int p) { return p; }
};
template <int DIR>
struct MixDft<5, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[5], const cf*, int) { dft_odd_sym<5, DIR>(v); }
    static constexpr int out_index(int p) { return p; }
};
template <int DIR>
struct MixDft<7, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[7], const cf*, int) { dft_odd_sym<7, DIR>(v); }
    static constexpr int out_index(int p) { return p; }
};
template <int DIR>
struct MixDft<9, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[9], const cf* wl, int L) { dft_square<3, DIR>(v, wl, L); }
    static constexpr int out_index(int p) { return p / 3 + 3 * (p % 3); }
};

template <int R, int DIR>
__device__ __forceinline__ void mix_stage(cf* lds, const cf* wl, int L, int Lj, int wlog2, int tid, int nthreads) {
    const int m = Lj / R, tws = L / Lj;
    const int nbf = (L / R) << wlog2;
    const float inv_m = 1.0f / (float)m;
    for (int b = tid; b < nbf; b += nthreads) {
        const int col = b & ((1 << wlog2) - 1), q = b >> wlog2;
        // q / m without the integer division: q < 1024 and m <= 512, so (q + 0.5) / m is at least 1e-3 away from an integer
        // and the float32 product (error < 1e-4) truncates to the exact quotient
        const int blk = (int)(((float)q + 0.5f) * inv_m), i = q - blk * m;
        cf* base = lds + (((blk * Lj + i)) << wlog2) + col;
        const int es = m << wlog2;
        cf v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = base[u * es];
        MixDft<R, DIR>::run(v, wl, L);
        if (m > 1) {
            const int iw = i * tws;
#pragma unroll
            for (int q2 = 1; q2 < R; ++q2) {
                const int u = MixDft<R, DIR>::out_index(q2);
                if (u != 0) {
                    const cf w = wl[iw * u];   // i u < Lj: the index stays below L
                    v[q2] = cmul(v[q2], DIR < 0 ? w : cconj(w));
                }
            }
        }
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) base[MixDft<R, DIR>::out_index(q2) * es] = v[q2];
    }
}

// Two workgroups per CU: a tile is at most 64 KiB of LDS (L * W <= 8192 complex64 points) and the kernel stays within 128
// VGPRs, so one workgroup's loads and stores overlap the other's stages -- the write path alone needs ~7 us for a tile, as
// long as its stages take (a single workgroup per CU with 128-KiB tiles and deferred stores measured 1.5x slower).
constexpr int kMixTileBytes = 65536;
template <int DIR>
__global__ __launch_bounds__(512, 4) void k_colmix(MixParams p) {
    constexpr int NT = 512;
    constexpr int NI = kMixTileBytes / (int)sizeof(cf) / NT;   // rows per thread at most
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int L = p.L, W = 1 << p.wlog2;             // W columns per tile (whole 64- or 128-byte pieces), a power of two <= 512
    cf* wl = lds + (size_t)L * W;
    unsigned short* perm = reinterpret_cast<unsigned short*>(wl + L);

    const int tid = threadIdx.x, f = tid & (W - 1), tau = tid >> p.wlog2;
    const int RL = NT >> p.wlog2;                    // rows in flight per sweep over the tile
    for (int k = tid; k < L; k += NT) {
        wl[k] = p.wl[k];
        perm[k] = p.perm[k];
    }
    // (the stage radices through LDS: indexing the by-value parameter array with the stage counter would put it in scratch)
    int* srad = reinterpret_cast<int*>(smem + (((size_t)L * W + L) * sizeof(cf) + (size_t)L * sizeof(unsigned short) + 15) / 16 * 16);
    if (tid < kMixMaxStages) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < kMixMaxStages; ++j) r = (tid == j) ? p.radix[j] : r;
        srad[tid] = r;
    }
    const int64_t ntile = (int64_t)p.S * p.nblock * p.ncolgrp;
    // tile T = (series s, row block c, column group g), g fastest; its first element is (time) index c * bstride + g * W
    auto tile_origin = [&](int64_t T, int64_t& s, int& c, int64_t& x0) -> int64_t {
        const int64_t g = T % p.ncolgrp, rest = T / p.ncolgrp;
        c = (int)(rest % p.nblock);
        s = rest / p.nblock;
        x0 = g * W;
        return (int64_t)c * p.bstride + x0;
    };
    // Buffer descriptors bound a tile to its L rows: the rows a thread's last sweep reaches beyond them load zeros and drop
    // their stores without a branch (tile extent < 2^31 bytes: rows are at most N / 2 elements apart and N <= 2^27).
    const uint32_t tile_bytes = (uint32_t)(((int64_t)(L - 1) * p.rstride + W) * (int64_t)sizeof(cf));
    const int voff = (int)(((int64_t)tau * p.rstride + f) * (int64_t)sizeof(cf));
    const int sweep = (int)((int64_t)RL * p.rstride * (int64_t)sizeof(cf));   // bytes between a thread's consecutive rows
    const int ni = (L + RL - 1) / RL;               // sweeps over the tile (uniform)
    const int64_t tstep = (int64_t)RL * p.rstride;

    // columns of a short last group: out-of-range offset (loads return zero, stores are dropped)
    auto col_off = [&](int64_t x0) -> int { return x0 + f < p.ncols ? voff : (int)0x80000000; };
    int64_t T = blockIdx.x;
    if (T >= ntile) return;
    int64_t Tn = T + gridDim.x;
    unsigned* slot = reinterpret_cast<unsigned*>(srad + kMixMaxStages);
    cf v[NI];
    {
        int64_t s, x0; int c;
        const int64_t e0 = tile_origin(T, s, c, x0);
        const rsrc_t rd = make_rsrc(p.ld + s * p.ld_plane + e0, tile_bytes);
        const int vo = col_off(x0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (i < ni) v[i] = buf_load(rd, vo, i * sweep);
    }
    while (true) {
        int64_t s, x0; int c;
        const int64_t e0 = tile_origin(T, s, c, x0);
        // inter-pass twiddle of this thread's rows tau + RL i: z_i = zb * zs^i (float64 recurrence)
        // (forward: the seeds are used after the stages and looked up there -- eight registers less across the stage
        //  butterflies, which is what the forward kernel spilled)
        const int64_t xx = (x0 + f) / p.xdiv, y0 = (int64_t)c * p.y0mul;
        double2 zb = make_double2(1, 0), zs = make_double2(1, 0);
        if (DIR > 0) {
            zb = big_tw(p.tw, ((xx * (y0 + p.ystep * tau)) % p.nmod) * p.mult);
            zs = big_tw(p.tw, ((xx * ((p.ystep * RL) % p.nmod)) % p.nmod) * p.mult);
        }
        {
            double2 z = zb;
            cf* dl = lds + tau * W + f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if (i < ni) {
                    cf a = v[i];
                    if (DIR > 0) {
                        a = cmul(a, make_cf((real)z.x, (real)-z.y));
                        z = zmul(z, zs);
                    }
                    if (tau + RL * i < L) dl[i * RL * W] = a;
                }
            }
        }
        unsigned fetched = (unsigned)(Tn + gridDim.x);   // the tile after next: requested now, parked in LDS after the stages
        if (tid == 0 && p.counter) fetched = 2 * gridDim.x + atomicAdd(p.counter, 1u);
        if (Tn < ntile) {   // the next tile's samples travel while this one is transformed
            int64_t s2, x2; int c2;
            const int64_t e2 = tile_origin(Tn, s2, c2, x2);
            const rsrc_t rd = make_rsrc(p.ld + s2 * p.ld_plane + e2, tile_bytes);
            const int vo = col_off(x2);
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (i < ni) v[i] = buf_load(rd, vo, i * sweep);
        }
        __syncthreads();
        int Lj = L;
        for (int j = 0; j < p.nstage; ++j) {
            const int r = srad[j];
            switch (r) {
                case 2: mix_stage<2, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                case 3: mix_stage<3, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                case 4: mix_stage<4, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                case 5: mix_stage<5, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                case 7: mix_stage<7, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                case 9: mix_stage<9, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
                default: mix_stage<8, DIR>(lds, wl, L, Lj, p.wlog2, tid, NT); break;
            }
            Lj /= r;
            __syncthreads();
        }
        if (DIR < 0) {
            int fl = f, tl = tau;
            asm volatile("" : "+v"(fl), "+v"(tl));   // (recomputed here, not carried across the stages)
            const int64_t xo = (x0 + fl) / p.xdiv;
            zb = big_tw(p.tw, ((xo * (y0 + p.ystep * tl)) % p.nmod) * p.mult);
            zs = big_tw(p.tw, ((xo * ((p.ystep * RL) % p.nmod)) % p.nmod) * p.mult);
        }
        {
            // stores: element (time) index t = e0 + f + k rstride goes to st[t - st_shift] when keep0 <= t < keep1
            const rsrc_t ro = make_rsrc(p.st + s * p.st_plane + e0 - p.st_shift, tile_bytes);
            const int64_t ot0 = e0 + f + (int64_t)tau * p.rstride;
            const int vst = col_off(x0);
            double2 z = zb;
            constexpr int CH = NI < 8 ? NI : 8;   // a chunk of rows at a time: permutation entries, then the values, then the stores
#pragma unroll
            for (int i0 = 0; i0 < NI; i0 += CH) {
                if (i0 < ni) {
                    int pk[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int k = tau + RL * (i0 + i);
                        pk[i] = ((int)perm[k < L ? k : L - 1] << p.wlog2) + f;
                    }
                    cf a[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) a[i] = lds[pk[i]];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        cf o = a[i];
                        if (DIR < 0) {
                            o = cmul(o, make_cf((real)z.x, (real)z.y));
                            z = zmul(z, zs);
                        }
                        const int64_t t = ot0 + (i0 + i) * tstep;
                        buf_store(ro, (t >= p.keep0 && t < p.keep1) ? vst : (int)0x80000000, (i0 + i) * sweep, o);   // out of range = dropped
                    }
                }
            }
        }
        if (Tn >= ntile) break;
        if (tid == 0) slot[0] = fetched;
        __syncthreads();   // every read of the tile is done before the next one is written over it
        T = Tn;
        Tn = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)slot[0]);
        __syncthreads();   // ... and everyone has the slot before thread 0 writes it again
    }
}

// ---- fused row pass of the 7-smooth plans whose power-of-two part is too small for the 2^k engine's rows ------------------
// N = N1 * N2 with N2 = 2^k * odd (k = 3 or 4: 8- or 16-element pieces for the column passes) and at most 1024 points: the
// rows are transformed by mixed-radix stages too.  A tile is FR consecutive rows of the planar work buffer -- ONE contiguous
// run of FR * N2 elements -- in LDS as it lies in memory.  Forward decimation-in-frequency stages leave every row in
// digit-reversed order; the chirp is stored in that order (ChirpParams::row_perm), so the product is element by element;
// the inverse runs the same stages backwards as decimation-in-time (conjugate twiddle first, inverse butterfly after),
// which takes the digit-reversed order back to the natural one: no permutation anywhere.
struct RowMixParams {
    cf* data;            // planar rows, in place: row g at g * N2
    const cf* chirp;     // plan order [chan][row][position], pre-scaled by 1/N
    int64_t nrows;       // S * N1
    int N1, npol, N2;
    int FR;              // rows per tile: FR * N2 <= the tile budget
    int nstage;
    int radix[kMixMaxStages];   // forward order; radices 9, 7, 5, 3, 8, 4, 2
    const cf* wl;        // W_N2^p
};

// one stage over every row of the tile: butterfly b = (row, q); INV = false: transform then twiddle (DIF), true: conjugate
// twiddle then inverse transform (DIT, undoing the matching forward stage)
template <int R, bool INV>
__device__ __forceinline__ void rowmix_stage(cf* lds, const cf* wl, int N2, int Lj, int rows, int tid, int nthreads) {
    const int m = Lj / R, tws = N2 / Lj, per_row = N2 / R;
    const int nbf = rows * per_row;
    const float inv_m = 1.0f / (float)m, inv_pr = 1.0f / (float)per_row;
    for (int b = tid; b < nbf; b += nthreads) {
        const int row = (int)(((float)b + 0.5f) * inv_pr), q = b - row * per_row;   // (exact: b < 2^13, see mix_stage)
        const int blk = (int)(((float)q + 0.5f) * inv_m), i = q - blk * m;
        cf* base = lds + row * N2 + blk * Lj + i;
        cf v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = base[u * m];
        const int iw = i * tws;
        if (INV && m > 1) {
#pragma unroll
            for (int u = 1; u < R; ++u) v[u] = cmul(v[u], cconj(wl[iw * u]));
        }
        if (INV) MixDft<R, +1>::run(v, wl, N2);
        else MixDft<R, -1>::run(v, wl, N2);
        // (register position q2 holds output index out_index(q2): the composite radix 9 leaves its outputs transposed)
        if (!INV && m > 1) {
#pragma unroll
            for (int q2 = 1; q2 < R; ++q2) {
                const int u = MixDft<R, -1>::out_index(q2);
                if (u != 0) v[q2] = cmul(v[q2], wl[iw * u]);
            }
        }
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) base[MixDft<R, -1>::out_index(q2) * m] = v[q2];
    }
}
// The last forward stage, the chirp and the first inverse stage in one round: the last stage's butterflies take R ADJACENT
// elements and have no twiddles, and the first inverse stage undoes exactly them -- so a thread transforms its R elements
// forward, multiplies by the chirp values of the same R positions (contiguous in memory: read straight from the chirp row,
// no staging) and transforms them back.  Two LDS round trips and two barriers less per tile than stage / product / stage.
template <int R>
__device__ __forceinline__ void rowmix_mid(cf* lds, const cf* wl, const cf* __restrict__ chirp, int N2, int rows, int tid,
                                           int nthreads) {
    const int nbf = rows * (N2 / R);
    for (int b = tid; b < nbf; b += nthreads) {
        cf* base = lds + b * R;           // row * N2 + q * R with q < N2 / R: butterfly b starts at element b * R of the tile
        const cf* cb = chirp + b * R;
        cf v[R], ch[R];
#pragma unroll
        for (int u = 0; u < R; ++u) ch[u] = cb[u];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = base[u];
        MixDft<R, -1>::run(v, wl, N2);
        cf w[R];
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) {   // register q2 holds output index out_index(q2): back to natural order for the inverse
            w[MixDft<R, -1>::out_index(q2)] = cmul(v[q2], ch[MixDft<R, -1>::out_index(q2)]);
        }
        MixDft<R, +1>::run(w, wl, N2);
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) base[MixDft<R, -1>::out_index(q2)] = w[q2];
    }
}
__device__ __forceinline__ void rowmix_mid_r(int r, cf* lds, const cf* wl, const cf* chirp, int N2, int rows, int tid, int nthreads) {
    switch (r) {
        case 2: rowmix_mid<2>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 3: rowmix_mid<3>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 4: rowmix_mid<4>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 5: rowmix_mid<5>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 7: rowmix_mid<7>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 8: rowmix_mid<8>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        case 9: rowmix_mid<9>(lds, wl, chirp, N2, rows, tid, nthreads); break;
        default: break;
    }
}

template <bool INV>
__device__ __forceinline__ void rowmix_stage_r(int r, cf* lds, const cf* wl, int N2, int Lj, int rows, int tid, int nthreads) {
    switch (r) {
        case 2: rowmix_stage<2, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 3: rowmix_stage<3, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 4: rowmix_stage<4, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 5: rowmix_stage<5, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 7: rowmix_stage<7, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 8: rowmix_stage<8, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        case 9: rowmix_stage<9, INV>(lds, wl, N2, Lj, rows, tid, nthreads); break;
        default: break;
    }
}

// FWD_ONLY: the forward stages alone (stand-alone transforms, pbh_fft_c2c): the rows stay in digit-reversed order, which the
// output pass undoes (k_fft_out, row_perm)
template <bool FWD_ONLY>
__global__ __launch_bounds__(512, 4) void k_rowmix(RowMixParams p) {
    constexpr int NT = 512;
    constexpr int NI = kMixTileBytes / (int)sizeof(cf) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int N2 = p.N2, FR = p.FR, tid = threadIdx.x;
    cf* wl = lds + FR * N2;
    int* srad = reinterpret_cast<int*>(wl + N2);   // radix and sub-length of every stage
    int* slen = srad + kMixMaxStages;
    for (int k = tid; k < N2; k += NT) wl[k] = p.wl[k];
    if (tid < kMixMaxStages) {
        int r = 1, len = N2;
#pragma unroll
        for (int j = 0; j < kMixMaxStages; ++j) {
            const int rj = j < p.nstage ? p.radix[j] : 1;
            if (tid == j) { r = rj; slen[tid] = len; }
            len /= rj;
        }
        srad[tid] = r;
    }
    // a tile is FR consecutive rows of ONE series (the last tile of a series is short): its chirp values are then one
    // contiguous run too, read with the same flat index as the samples
    const int64_t tps = (p.N1 + FR - 1) / FR, ntile = (p.nrows / p.N1) * tps;
    auto tile_geom = [&](int64_t t, int64_t& dbase, int64_t& cbase) -> int {
        const int64_t srs = t / tps, r0 = (t - srs * tps) * FR;
        dbase = (srs * p.N1 + r0) * N2;
        cbase = ((srs / p.npol) * p.N1 + r0) * N2;
        const int64_t left = p.N1 - r0;
        return (int)(left < FR ? left : FR);
    };
    int64_t t = blockIdx.x;
    if (t >= ntile) return;
    cf v[NI];
    {
        int64_t db, cb;
        const int cnt = tile_geom(t, db, cb) * N2;
        const cf* src = p.data + db;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + NT * i;
            v[i] = src[e < cnt ? e : cnt - 1];   // (clamped, not predicated: no branches around the loads)
        }
    }
    while (true) {
        int64_t db, cb;
        const int rows = tile_geom(t, db, cb), cnt = rows * N2;
        cf* dst = p.data + db;
        __syncthreads();   // the previous tile's stores have read LDS
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + NT * i;
            if (e < cnt) lds[e] = v[i];
        }
        const int64_t tn = t + gridDim.x;
        if (tn < ntile) {   // the next tile's samples travel while this one is transformed
            int64_t db2, cb2;
            const int cn = tile_geom(tn, db2, cb2) * N2;
            const cf* src = p.data + db2;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int e = tid + NT * i;
                v[i] = src[e < cn ? e : cn - 1];
            }
        }
        __syncthreads();
        const int nfwd = FWD_ONLY ? p.nstage : p.nstage - 1;
        for (int j = 0; j < nfwd; ++j) {
            rowmix_stage_r<false>(srad[j], lds, wl, N2, slen[j], rows, tid, NT);
            __syncthreads();
        }
        if constexpr (!FWD_ONLY) {
            // last forward stage x chirp (stored in the rows' digit-reversed order) x first inverse stage, in one round
            rowmix_mid_r(srad[p.nstage - 1], lds, wl, p.chirp + cb, N2, rows, tid, NT);
            __syncthreads();
            for (int j = p.nstage - 2; j >= 0; --j) {
                rowmix_stage_r<true>(srad[j], lds, wl, N2, slen[j], rows, tid, NT);
                __syncthreads();
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e = tid + NT * i;
            if (e < cnt) dst[e] = lds[e];
        }
        if (tn >= ntile) break;
        t = tn;
    }
}

}  // namespace PBH_NS
