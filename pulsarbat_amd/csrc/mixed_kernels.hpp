// mixed_kernels.hpp -- column transforms of 7-smooth length (radices 2, 3, 4, 5, 7) for the lengths that
// pulsarbat.utils.next_fast_len / prev_fast_len hand out (reference pulsarbat/utils.py:68-130, transforms.py:364-382).
//
// A length N = N1 * N2 with N2 = 2^k (the row pass, power-of-two tile engine) and N1 = P * Q any 7-smooth number runs the
// planar pipeline of pbhip.hip with BOTH column roles played by this one kernel:
//   role A (P > 1, what k_radix_p does for P <= 16): the P-point transform over n1 = Q a + b, rows one chunk = N / P apart,
//           then the twiddle W_N1^{b c};
//   role B (what k_colq does for powers of two): the Q-point transform over b inside row block c, then W_N^{n2 k1},
//           k1 = c + P d.
// The inverse direction mirrors both (conjugate twiddle first, inverse transform after).
//
// One tile = L rows x COLS columns (COLS * sizeof(cf) = 128 bytes: one line per row) in LDS, transformed IN PLACE by
// decimation-in-frequency stages of radix r_1, r_2, ... (each: L / r butterflies per column, twiddle W_Lj^{i u} from a W_L table
// in LDS), which leaves X[k] at the mixed-radix digit-reversed position; the stores read LDS through a permutation table, so
// rows leave in natural order.  Per-thread float64 recurrence for the inter-pass twiddle as in k_colq.  The pass is bound
// by HBM like the other column passes as long as the LDS work of its stages (about 1.3 us per stage and tile) stays under
// the ~12 us a CU has per 256 KiB of traffic; the next tile's samples are requested before the stages of the current one.
#pragma once
#include "fft_core.hpp"

namespace PBH_NS {

constexpr int kMixMaxStages = 14;
constexpr int kMixMaxLen = 1024;   // rows of a tile: 128 KiB of full 128-byte lines in both precisions

struct MixParams {
    const cf* ld;       // loads: series s at s * ld_plane
    int64_t ld_plane;
    cf* st;             // stores: series s at s * st_plane, element index (time) - st_shift
    int64_t st_plane;
    int S, L;           // series, transform length (rows of a tile)
    int64_t rstride;    // elements between consecutive rows
    int nblock;         // row blocks per series (role B: P blocks of Q rows), block c at c * bstride
    int64_t bstride;
    int64_t ncolgrp;    // COLS-column groups per block
    // twiddle exponent of (row k, column position x): ((x / xdiv) * (c * y0mul + ystep * k) % nmod) * mult, looked up in tw
    int64_t xdiv, ystep, nmod, mult;
    int y0mul;
    BigTwiddle tw;
    int nstage;
    int radix[kMixMaxStages];
    const cf* wl;                  // W_L^p = exp(-2 pi i p / L), p < L
    const unsigned short* perm;    // natural row k sits at LDS row perm[k] after the stages
    int64_t keep0, keep1, st_shift;   // only element indices in [keep0, keep1) are stored (crop of the last inverse pass)
};

template <int R, int DIR>
__device__ __forceinline__ void mix_stage(cf* lds, const cf* wl, int L, int Lj, int cols, int tid, int nthreads) {
    const int m = Lj / R, tws = L / Lj;
    const int nbf = (L / R) * cols;
    for (int b = tid; b < nbf; b += nthreads) {
        const int col = b % cols, q = b / cols;
        const int blk = q / m, i = q - blk * m;
        cf* base = lds + (blk * Lj + i) * cols + col;
        cf v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = base[u * m * cols];
        Dft<R, DIR>::run(v);
        if (m > 1) {
#pragma unroll
            for (int u = 1; u < R; ++u) {
                const cf w = wl[i * u * tws];   // i u < Lj: the index stays below L
                v[u] = cmul(v[u], DIR < 0 ? w : cconj(w));
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) base[u * m * cols] = v[u];
    }
}

template <int DIR>
__global__ __launch_bounds__(512) void k_colmix(MixParams p) {
    constexpr int COLS = 128 / (int)sizeof(cf);
    constexpr int NT = 512, RL = NT / COLS;          // RL rows in flight per pass over the tile
    constexpr int NI = kMixMaxLen / RL;              // rows per thread at the longest length
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    cf* wl = lds + (size_t)p.L * COLS;
    unsigned short* perm = reinterpret_cast<unsigned short*>(wl + p.L);

    const int tid = threadIdx.x, f = tid % COLS, tau = tid / COLS;
    const int L = p.L;
    for (int k = tid; k < L; k += NT) {
        wl[k] = p.wl[k];
        perm[k] = p.perm[k];
    }
    const int64_t ntile = (int64_t)p.S * p.nblock * p.ncolgrp;
    auto tile_base = [&](int64_t T, int64_t& s, int& c, int64_t& x) -> int64_t {
        const int64_t g = T % p.ncolgrp, rest = T / p.ncolgrp;
        c = (int)(rest % p.nblock);
        s = rest / p.nblock;
        x = g * COLS + f;
        return (int64_t)c * p.bstride + x;    // element index (time) of row 0 of this thread's column
    };

    int64_t T = blockIdx.x;
    if (T >= ntile) return;
    cf v[NI];
    {
        int64_t s, x; int c;
        const int64_t e0 = tile_base(T, s, c, x);
        const cf* src = p.ld + s * p.ld_plane + e0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int k = tau + RL * i;
            if (k < L) v[i] = src[(int64_t)k * p.rstride];
        }
    }
    while (true) {
        int64_t s, x; int c;
        const int64_t e0 = tile_base(T, s, c, x);
        // inter-pass twiddle of this thread's rows tau + RL i: z_i = zb * zs^i (float64 recurrence)
        const int64_t xx = x / p.xdiv, y0 = (int64_t)c * p.y0mul;
        const double2 zb = big_tw(p.tw, ((xx * (y0 + p.ystep * tau)) % p.nmod) * p.mult);
        const double2 zs = big_tw(p.tw, ((xx * ((p.ystep * RL) % p.nmod)) % p.nmod) * p.mult);
        {
            double2 z = zb;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int k = tau + RL * i;
                if (k < L) {
                    cf a = v[i];
                    if (DIR > 0) a = cmul(a, make_cf((real)z.x, (real)-z.y));
                    lds[k * COLS + f] = a;
                }
                if (DIR > 0) z = zmul(z, zs);
            }
        }
        const int64_t Tn = T + gridDim.x;
        if (Tn < ntile) {   // the next tile's samples travel while this one is transformed
            int64_t s2, x2; int c2;
            const int64_t e2 = tile_base(Tn, s2, c2, x2);
            const cf* src = p.ld + s2 * p.ld_plane + e2;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int k = tau + RL * i;
                if (k < L) v[i] = src[(int64_t)k * p.rstride];
            }
        }
        __syncthreads();
        int Lj = L;
        for (int j = 0; j < p.nstage; ++j) {
            const int r = p.radix[j];
            switch (r) {
                case 2: mix_stage<2, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
                case 3: mix_stage<3, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
                case 4: mix_stage<4, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
                case 5: mix_stage<5, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
                case 7: mix_stage<7, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
                default: mix_stage<8, DIR>(lds, wl, L, Lj, COLS, tid, NT); break;
            }
            Lj /= r;
            __syncthreads();
        }
        {
            cf* dst = p.st + s * p.st_plane - p.st_shift;
            double2 z = zb;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int k = tau + RL * i;
                if (k < L) {
                    cf a = lds[(int)perm[k] * COLS + f];
                    if (DIR < 0) a = cmul(a, make_cf((real)z.x, (real)z.y));
                    const int64_t t = e0 + (int64_t)k * p.rstride;
                    if (t >= p.keep0 && t < p.keep1) dst[t] = a;
                }
                if (DIR < 0) z = zmul(z, zs);
            }
        }
        if (Tn >= ntile) break;
        T = Tn;
        __syncthreads();   // every read of the tile is done before the next one is written over it
    }
}

}  // namespace PBH_NS
