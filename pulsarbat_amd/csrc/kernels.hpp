// kernels.hpp -- the hot-path kernels (gfx950).  Included by exactly one TU per kernel family.
//
//   k_col<M, OP>   column pass: F = 2^14/M strided FFTs of length M per workgroup
//                  OP_FWD_TW : forward FFT over n1, then * W_N^{n2 k1}          (pass 1)
//                  OP_TW_INV : * conj W_N^{m k1}, inverse FFT over k1, crop    (pass 3)
//   k_row<M>       fused row pass: forward FFT over n2, * chirp, inverse FFT   (pass 2)
//   k_small<M>     whole transform in one tile when nsample = M <= 2^14
//
// Together they replace `ifft(fft(z.data, axis=0) * chirp, axis=0)[start:stop]`
// (pulsarbat/transforms/dedispersion.py:125-133).  With N = N1*N2, n = N2 n1 + n2,
// k = k1 + N1 k2:   X[k1 + N1 k2] = sum_n2 W_N2^{n2 k2} W_N^{n2 k1} sum_n1 x[N2 n1 + n2] W_N1^{n1 k1}.
// Pass 2 holds, for one k1, every k2 -- exactly the set the inverse transform's first level
// needs -- so forward level 2, the chirp product and inverse level 1 fuse into one kernel and
// the whole convolution is 3 HBM round trips instead of 4.
#pragma once
#include "fft_core.hpp"

namespace PBH_NS {

enum ColOp { OP_FWD_TW = 0, OP_TW_INV = 1 };

enum Layout { LAYOUT_INTERLEAVED = 0, LAYOUT_PLANAR = 1, LAYOUT_BLOCK = 2 };

// Column addressing.  A "column" is one (n2, series) pair; element (row, n2, s):
//   interleaved: row * (N2*S) + n2 * S + s          (the reference's (nsample, nchan, npol) block)
//   planar     : s * plane + row * N2 + n2          (one contiguous series per plane)
struct ColSide {
    int layout;
    int64_t plane;       // planar: elements between series
    int64_t row_stride;  // elements between consecutive rows: N2*S (interleaved) or N2 (planar)
};

struct ColParams {
    const cf* in;
    cf* out;
    ColSide is, os;
    int enum_layout;  // which side's column order the tiles enumerate (that side is contiguous);
                      // LAYOUT_BLOCK: a tile is CB adjacent n2 x SB adjacent series (SB*CB = F), both
                      // sides see SB*8 / CB*8 byte pieces; tiles sharing 128-B lines sit on one XCD
    int SB;           // LAYOUT_BLOCK: series per tile
    int lane_order;   // LAYOUT_BLOCK: 0 = series fastest across lanes, 1 = n2 fastest
    int S;            // series = nchan * npol
    int N2;           // columns per series
    int64_t ncols;    // S * N2
    int ntile;        // ceil(ncols / F)
    BigTwiddle tw;    // W_N, N = M * N2
    const cf* tw16k;  // stage twiddles
    int64_t crop_start, crop_stop;  // OP_TW_INV: keep time index t in [start, stop), t = row*N2 + n2
    int64_t out_shift;              // subtracted from the output offset (crop_start * S when interleaved)
};

__device__ __forceinline__ int64_t col_addr(const ColSide& sd, int64_t row, int n2, int s, int S, int N2) {
    return sd.layout == LAYOUT_INTERLEAVED ? (row * N2 + n2) * (int64_t)S + s
                                           : (int64_t)s * sd.plane + row * N2 + n2;
}

// Column (n2, series) handled by lane f of tile `tile`.
template <int F>
__device__ __forceinline__ bool col_decode(const ColParams& p, int64_t tile, int f, int& n2, int& s) {
    n2 = 0;
    s = 0;
    if (p.enum_layout == LAYOUT_BLOCK) {
        // super-group = tiles that share input lines (all series groups of one n2 group) and output
        // lines (the n2 groups of one aligned 16-column block): NH * (16/CB) tiles, dealt to ONE XCD
        // (tiles b, b+8, b+16, ... run on one XCD) so its L2 merges their partial-line accesses.
        const int SB = p.SB, CB = F / SB;
        const int NH = p.S / SB;
        const int GG = CB >= 16 ? 1 : 16 / CB;
        const int SG = NH * GG;
        const int b = (int)tile;
        int j, G;
        if ((p.ntile / SG) % 8 == 0) {
            const int xcd = b & 7, r = b >> 3;
            j = r % SG;
            G = (r / SG) * 8 + xcd;
        } else {
            j = b % SG;
            G = b / SG;
        }
        const int h = j % NH, gg = j / NH;
        const int g = G * GG + gg;
        const int fs = p.lane_order ? f / CB : f % SB;
        const int fc = p.lane_order ? f % CB : f / SB;
        n2 = g * CB + fc;
        s = h * SB + fs;
        return true;
    }
    const int64_t q = tile * F + f;
    if (q >= p.ncols) return false;
    if (p.enum_layout == LAYOUT_INTERLEAVED) {
        n2 = (int)(q / p.S);
        s = (int)(q - (int64_t)n2 * p.S);
    } else {
        s = (int)(q / p.N2);
        n2 = (int)(q - (int64_t)s * p.N2);
    }
    return true;
}

// Generic column pass (any pair of layouts): one tile per workgroup.  XS = split (re, then im)
// LDS exchange: 64 KiB instead of 128 KiB and a 128-VGPR cap, so TWO workgroups share a CU and
// one's loads/stores overlap the other's butterflies.
template <int M, int OP, int R, bool XS>
__global__ __launch_bounds__(kTilePoints / R, XS ? (2 * kTilePoints / R) / 256 : 1) void k_col(ColParams p) {
    constexpr int F = kTilePoints / M;  // columns per tile
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;           // row stride between a thread's points
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    int n2, s;
    const bool valid = col_decode<F>(p, blockIdx.x, f, n2, s);

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    // inter-pass twiddle W_N^{n2 * k1}, k1 = tau + i*M/R: base * step^i, float64 recurrence
    const double2 zb = big_tw(p.tw, (int64_t)n2 * tau);
    const double2 zs = big_tw(p.tw, (int64_t)n2 * MR);

    cf v[R];
    {
        const cf* src = p.in + col_addr(p.is, tau, n2, s, p.S, p.N2);
        const int64_t step = (int64_t)MR * p.is.row_stride;
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = valid ? src[i * step] : make_cf(0, 0);
    }
    cf* dst = p.out + col_addr(p.os, tau, n2, s, p.S, p.N2) - p.out_shift;
    const int64_t ostep = (int64_t)MR * p.os.row_stride;

    if constexpr (OP == OP_TW_INV) {
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = cmul(v[i], make_cf((real)z.x, (real)-z.y));
            z = zmul(z, zs);
        }
        fft_tile<M, 1, R, +1, F, PAD, XS>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int64_t t = (int64_t)(tau + i * MR) * p.N2 + n2;
            if (valid && t >= p.crop_start && t < p.crop_stop) dst[i * ostep] = v[i];
        }
    } else {
        fft_tile<M, 1, R, -1, F, PAD, XS>(v, lds, tau, f, w);
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            cf r = cmul(v[i], make_cf((real)z.x, (real)z.y));
            z = zmul(z, zs);
            if (valid) dst[i * ostep] = r;
        }
    }
}

// ---- persistent column pass for the planar, in-place case (planar5's passes 2 and 4) ------------------
// One workgroup per CU (128 KiB of LDS admits one) walks over tiles (series s, 16-column group g).
// History: a first persistent version that stored each tile as soon as it was ready and
// walked tiles with a static stride was SLOWER than one-tile workgroups (1.11 vs 0.98 ms); the two
// things that made the persistent form win are in k_colq's header below.
struct ColpParams {
    cf* data;          // planar work buffer: series s at s * plane, row r at r * N2 (stores; loads too unless ld set)
    int64_t plane;
    int S, N2;
    BigTwiddle tw;     // W_N, N = M * N2
    const cf* tw16k;
    int64_t crop_start, crop_stop;  // OP_TW_INV: keep time index t in [start, stop), t = row*N2 + n2
    int order;         // tile order: 0 = column groups of one series consecutive, 1 = series fastest
    unsigned* counter; // k_colq: dynamic tile hand-out (zeroed before the launch); null = static stride
    // series-major device I/O (pbh_dedisperse_layout): loads may come from a different planar array
    // (the caller's input, pitch ld_plane) and stores may go to one (the caller's output, pitch
    // `plane`, time index shifted by st_shift = crop_start so that sample `start` lands at element 0)
    const cf* ld = nullptr;
    int64_t ld_plane = 0;
    int64_t st_shift = 0;
    int P = 1;  // column transform split P x M (k_radix_p did the radix-P stage): a series is P blocks of M rows
    // DET (OP_TW_INV, F = 16 ... 256 columns, P = 1): the pass stores no voltages.  Each tile leaves |z|^2 summed over every
    // 16 of its columns, one float per row, in det_part[(series * N2/16 + g16) * M + tau * R + i] (row = tau + (M/R) i); a column group that holds
    // a scrunch boundary (time index == crop_start mod det_ns) inside it leaves the columns before the boundary there and
    // the rest in det_side[(series * (N2/det_ns) + boundary) * M + ...].  k_detect_reduce sums them per output sample.
    real* det_part = nullptr;
    real* det_side = nullptr;
    int det_ns = 0;
};

#ifndef PBH_DET_PUMP
#define PBH_DET_PUMP 2   // loads requested per tick of the detecting column pass (it has no stores to pace them with)
#endif
#ifndef PBH_F64
// sum over the 16 lanes of a DPP row of 32 values per lane, transposed on the way: lane f ends with the totals of values
// 2f and 2f + 1.  Step k halves the values a lane is responsible for (the half chosen by one bit of f) and adds the partner's
// contribution to that half; partners: f^8 (row_ror:8), f^7 (row_half_mirror), f^2, f^1 (quad_perm).  30 DPP adds
// instead of the 128 of an all-reduce, and the result is one 8-byte store per lane.
template <int CTRL>
__device__ __forceinline__ float dpp_from(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int N, int CTRL>
__device__ __forceinline__ void treduce_step(const float* in, float* out, bool bit) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float keep = bit ? in[j + N] : in[j];
        const float send = bit ? in[j] : in[j + N];
        out[j] = keep + dpp_from<CTRL>(send);
    }
}
__device__ __forceinline__ float2 treduce16x32(const float* a, int f) {
    float b[16], c[8], d[4], e[2];
    treduce_step<16, 0x128>(a, b, (f & 8) != 0);
    treduce_step<8, 0x141>(b, c, (f & 4) != 0);
    treduce_step<4, 0x4E>(c, d, (f & 2) != 0);
    treduce_step<2, 0xB1>(d, e, (f & 1) != 0);
    return make_float2(e[0], e[1]);
}
// the same over the 8 lanes of HALF a DPP row (partners f^7, f^2, f^1): lane f & 7 ends with the totals of values 4f .. 4f + 3
__device__ __forceinline__ float4 treduce8x32(const float* a, int f) {
    float c[16], d[8], e[4];
    treduce_step<16, 0x141>(a, c, (f & 4) != 0);
    treduce_step<8, 0x4E>(c, d, (f & 2) != 0);
    treduce_step<4, 0xB1>(d, e, (f & 1) != 0);
    return make_float4(e[0], e[1], e[2], e[3]);
}
#endif

// ---- persistent column pass with deferred, interleaved stores --------------------------------------
// The outputs of tile i are not stored when they are ready: a burst of 32 stores per thread blocks
// the wave until the write path has taken 128 KiB (writes are the slow direction: the burst alone
// was ~45 % of an iteration).  They stay in registers and are stored during tile i+1's butterflies,
// two at a time at fft_tile's tick points, each followed by the load of the same slot of tile i+2 --
// the slot's registers pass from "output waiting to be stored" to "input on its way", so the budget
// is two tiles of registers (the one being transformed, and the out/in slots).  Out-of-range
// descriptors (0 bytes) turn the first iteration's stores and the last iterations' loads into
// no-ops without branches.
// DET: 0 = voltages stored; 1 = |z|^2 summed over every 16 columns of a series (intensity, Stokes I); 2 (round 4) = the tile is
// 8 columns x BOTH polarisations of a channel (lanes 0-7 / 8-15 of a DPP row: 64-byte pieces of two planes instead of one
// 128-byte piece), and what is summed over the 8 columns is |a|^2, |b|^2, Re conj(a) b, Im conj(a) b -- all four Stokes
// parameters without storing the voltages and without a second tile waiting in registers (the round-3 attempt spilled).
template <int M, int OP, int R, int DET = 0>
__global__ __launch_bounds__(kTilePoints / R) void k_colq(ColpParams p) {
    constexpr int F = kTilePoints / M;
    static_assert(!DET || (OP == OP_TW_INV && F % 16 == 0 && M >= 2 * R && R == 32), "the detect form reduces over the 16 lanes of a DPP row");
    static_assert(DET != 2 || F == 16, "the pol-pair detect form: 16-column tiles (8 columns x 2 pols)");
    constexpr int CW = DET == 2 ? F / 2 : F;   // columns n2 of a tile
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const uint32_t ngrp = (uint32_t)(p.N2 / CW);
    // (DET 2: the "series" of a tile is a channel = a polarisation pair; lanes 8-15 read the second plane)
    const uint32_t ntile = ngrp * (uint32_t)(DET == 2 ? p.S / 2 : p.S) * (uint32_t)p.P;   // (series, row block, column group)
    const int pitch = p.N2;   // elements between consecutive rows of the planar arrays (a padded pitch was measured: see oop_mode)
    const int64_t ldp0 = p.ld ? p.ld_plane : p.plane;
    const int voff = DET == 2 ? (int)(((int64_t)tau * pitch + (f & 7) + (int64_t)(f >> 3) * ldp0) * (int64_t)sizeof(cf))
                              : (tau * pitch + f) * (int)sizeof(cf);
    const int stepb = MR * pitch * (int)sizeof(cf);
    const uint32_t tile_bytes = (uint32_t)(((int64_t)(M - 1) * pitch + CW + (DET == 2 ? ldp0 : 0)) * (int64_t)sizeof(cf));
    const int shift = p.tw.shift;
    const int64_t lomask = (1LL << shift) - 1;
    const uint32_t c0 = (uint32_t)p.crop_start, c1 = (uint32_t)p.crop_stop;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    // tile t = (series s, row block c, column group g), g fastest
    auto group_of = [&](uint32_t t) -> int { return (int)(t % ngrp); };
    auto block_of = [&](uint32_t t) -> int { return (int)((t / ngrp) % (uint32_t)p.P); };
    auto series_of = [&](uint32_t t) -> int { return (int)(t / (ngrp * (uint32_t)p.P)); };
    const int64_t blk = (int64_t)M * pitch;   // elements per row block
    const cf* ldb = p.ld ? p.ld : p.data;
    const int64_t ldp = p.ld ? p.ld_plane : p.plane;
    auto tile_rsrc = [&](uint32_t t) {   // where tile t is loaded from
        return t < ntile ? make_rsrc(ldb + (int64_t)series_of(t) * (DET == 2 ? 2 : 1) * ldp + block_of(t) * blk + (int64_t)group_of(t) * CW, tile_bytes)
                         : make_rsrc(p.data, 0);
    };
    auto store_rsrc = [&](uint32_t t) {  // where its outputs go (the base may lie before the array when the
                                         // first columns are cropped: only in-range offsets are ever used)
        return make_rsrc(p.data + (int64_t)series_of(t) * p.plane + block_of(t) * blk + (int64_t)group_of(t) * F - p.st_shift,
                         tile_bytes);
    };
    // inter-pass twiddle W_N^{n2 k1}, k1 = c + P*(tau + i*MR): base and step of the float64 recurrence
    auto load_tables = [&](int n2, int c, double2& bh, double2& bl, double2& sh, double2& sl) {
        const int64_t pb = tw_reduce(p.tw, (int64_t)n2 * (c + (int64_t)p.P * tau));
        const int64_t ps = tw_reduce(p.tw, (int64_t)n2 * p.P * MR);
        bh = p.tw.hi[pb >> shift];
        bl = p.tw.lo[pb & lomask];
        sh = p.tw.hi[ps >> shift];
        sl = p.tw.lo[ps & lomask];
    };

    // Tile order.  The first two tiles of a workgroup are static (b, b + G); after that tiles are handed
    // out by an atomic counter when p.counter is set, one iteration ahead through an LDS slot, so the
    // tiles in flight across the chip stay a tight window of consecutive column groups (what hardware
    // dispatch of one-tile workgroups gives, and what DRAM pages like) however the workgroups drift.
    const uint32_t G = gridDim.x;
    unsigned* slot = reinterpret_cast<unsigned*>(smem + lds_tile_bytes<PAD>());
    uint32_t t = blockIdx.x;
    if (t >= ntile) return;
    uint32_t tn = t + G;
    int g = group_of(t);
    const rsrc_t rd0 = tile_rsrc(t);     // first tile, loaded up front
    rsrc_t rdo = make_rsrc(p.data, 0);   // where the outputs waiting in `out` go (none yet)
    uint32_t tto = 0;                    // time index of that tile's first sample in this thread (N < 2^31)
    const uint32_t rowstep = (uint32_t)MR * (uint32_t)p.N2;
    double2 zbh, zbl, zsh, zsl;
    const int fcol = DET == 2 ? (f & 7) : f;   // column of this lane within the tile
    load_tables(g * CW + fcol, block_of(t), zbh, zbl, zsh, zsl);
    cf v[R], out[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        v[i] = buf_load(rd0, voff, i * stepb);
        out[i] = make_cf(0, 0);
    }

    while (true) {
        launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
        const int n2 = g * CW + fcol;
        const double2 zb = zmul(zbh, zbl), zs = zmul(zsh, zsl);
        const bool more = tn < ntile;
        // the tile after next: requested now, its index parked in LDS after the butterflies (the atomic's
        // round trip must not be waited for here)
        unsigned fetched = tn + G;
        if (tid == 0 && p.counter) fetched = 2 * G + atomicAdd(p.counter, 1u);
        const rsrc_t rd2 = tile_rsrc(tn);          // tile whose samples are requested during this iteration
        cf in2[R];                                 // ... into the registers the stored outputs leave
        int cnt = 0;
        auto pump = [&](int n) {                   // store n waiting outputs, request n samples
#pragma unroll
            for (int k = 0; k < n; ++k) {
                if (cnt < R) {
                    if constexpr (DET != 0) {
                    } else if constexpr (OP == OP_TW_INV) {
                        // cropped samples: an out-of-range offset makes the hardware drop the store (no branch)
                        const uint32_t tt = tto + (uint32_t)cnt * rowstep;
                        buf_store(rdo, (tt >= c0 && tt < c1) ? voff : (int)0x80000000, cnt * stepb, out[cnt]);
                    } else {
                        buf_store(rdo, voff, cnt * stepb, out[cnt]);
                    }
                    in2[cnt] = buf_load(rd2, voff, cnt * stepb);
                    ++cnt;
                }
            }
        };
        auto hk = [&](auto st, auto q) {
            if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                __builtin_amdgcn_sched_barrier(0x38E);  // VALU / SALU / LDS may move across, global memory ops stay put
                pump(DET ? PBH_DET_PUMP : 2);
                __builtin_amdgcn_sched_barrier(0x38E);
            }
        };
        if constexpr (OP == OP_TW_INV) {
            double2 z = zb;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                v[i] = cmul(v[i], make_cf((real)z.x, (real)-z.y));
                z = zmul(z, zs);
            }
            fft_tile<M, 1, R, +1, F, PAD, false, false>(v, lds, tau, f, w, hk);
        } else {
            fft_tile<M, 1, R, -1, F, PAD, false, false>(v, lds, tau, f, w, hk);
        }
        pump(R);  // whatever the ticks did not reach (short transforms)
        if (tid == 0) slot[0] = fetched;
        __syncthreads();
        const uint32_t tnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[0]);  // wave-uniform: descriptors stay in SGPRs
        // tables of the next tile: after every request above, before nothing that has to wait for them
        const int gn = more ? group_of(tn) : g;
        if (more) load_tables(gn * CW + fcol, block_of(tn), zbh, zbl, zsh, zsl);
#ifndef PBH_F64
        if constexpr (DET == 2) {
            // lanes 0-7: polarisation a, lanes 8-15: polarisation b of the same 8 columns; the partner's value comes over DPP
            // (row_ror:8).  a-lanes carry |a|^2 and Re conj(a) b, b-lanes |b|^2 and Im conj(a) b; both are summed over the
            // 8 columns by a transposing reduction (lane fl ends with rows 4 fl .. 4 fl + 3): two 16-byte stores per lane.
            const int fl = f & 7;
            const bool polb = (f & 8) != 0;
            float pw[R], qw[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float wx = dpp_from<0x128>(v[i].x), wy = dpp_from<0x128>(v[i].y);
                pw[i] = v[i].x * v[i].x + v[i].y * v[i].y;
                qw[i] = polb ? (wx * v[i].y - wy * v[i].x) : (v[i].x * wx + v[i].y * wy);
            }
            const int c8 = g * 8;
            const int bmod = (int)(p.crop_start % p.det_ns), bcol = bmod & 7;
            const bool split = bcol != 0 && c8 % p.det_ns == bmod - bcol;   // tile-uniform: a scrunch boundary inside this group
            const int64_t at = (int64_t)tau * R + 4 * fl;
            const int64_t chan = series_of(t);
            const int64_t ng8 = p.N2 / 8, nb = p.N2 / p.det_ns;
            float4* dp = reinterpret_cast<float4*>(p.det_part + ((chan * 4 + (polb ? 1 : 0)) * ng8 + g) * M + at);
            float4* dq = reinterpret_cast<float4*>(p.det_part + ((chan * 4 + 2 + (polb ? 1 : 0)) * ng8 + g) * M + at);
            if (split) {
                float lo[R];
#pragma unroll
                for (int i = 0; i < R; ++i) { lo[i] = fl >= bcol ? 0.0f : pw[i]; pw[i] -= lo[i]; }
                *dp = treduce8x32(lo, fl);
                const float4 hp = treduce8x32(pw, fl);
#pragma unroll
                for (int i = 0; i < R; ++i) { lo[i] = fl >= bcol ? 0.0f : qw[i]; qw[i] -= lo[i]; }
                *dq = treduce8x32(lo, fl);
                const float4 hq = treduce8x32(qw, fl);
                const int64_t sb = c8 / p.det_ns;
                *reinterpret_cast<float4*>(p.det_side + ((chan * 4 + (polb ? 1 : 0)) * nb + sb) * M + at) = hp;
                *reinterpret_cast<float4*>(p.det_side + ((chan * 4 + 2 + (polb ? 1 : 0)) * nb + sb) * M + at) = hq;
            } else {
                *dp = treduce8x32(pw, fl);
                *dq = treduce8x32(qw, fl);
            }
        } else if constexpr (DET == 1) {
            float pw[R];
#pragma unroll
            for (int i = 0; i < R; ++i) pw[i] = v[i].x * v[i].x + v[i].y * v[i].y;
            // a tile is F / 16 groups of 16 columns, one per DPP row of the wave; the partial sums are per group
            const int fl = f & 15, c16 = g * F + (f & ~15);      // lane's column within its group; the group's first column
            const int bmod = (int)(p.crop_start % p.det_ns), bcol = bmod & 15;
            const int first = (g * F) % p.det_ns;                // tile-uniform: does one of this tile's groups hold a boundary?
            const bool split = bcol != 0 && first <= bmod - bcol && bmod - bcol < first + F;
            const int64_t at = (int64_t)tau * R + 2 * fl;
            float2* dst = reinterpret_cast<float2*>(p.det_part + ((int64_t)series_of(t) * (p.N2 / 16) + c16 / 16) * M + at);
            if (split) {
                const bool mine = c16 % p.det_ns == bmod - bcol;   // this lane's group is the one
                float lo[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    lo[i] = (mine && fl >= bcol) ? 0.0f : pw[i];
                    pw[i] -= lo[i];
                }
                *dst = treduce16x32(lo, fl);
                const float2 hi = treduce16x32(pw, fl);
                if (mine)
                    *reinterpret_cast<float2*>(p.det_side + ((int64_t)series_of(t) * (p.N2 / p.det_ns) + c16 / p.det_ns) * M + at) = hi;
            } else {
                *dst = treduce16x32(pw, fl);
            }
        } else
#endif
        if constexpr (OP == OP_TW_INV) {
#pragma unroll
            for (int i = 0; i < R; ++i) out[i] = v[i];
        } else {
            double2 z = zb;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                out[i] = cmul(v[i], make_cf((real)z.x, (real)z.y));
                z = zmul(z, zs);
            }
        }
        rdo = store_rsrc(t);
        tto = (uint32_t)tau * (uint32_t)p.N2 + (uint32_t)n2;
        if (!more) break;
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = in2[i];
        __syncthreads();   // everyone has read the slot before thread 0 overwrites it
        t = tn;
        tn = tnn;
        g = gn;
    }
    // drain: the last tile's outputs
    if constexpr (DET != 0) return;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if constexpr (OP == OP_TW_INV) {
            const uint32_t tt = tto + (uint32_t)i * rowstep;
            buf_store(rdo, (tt >= c0 && tt < c1) ? voff : (int)0x80000000, i * stepb, out[i]);
        } else {
            buf_store(rdo, voff, i * stepb, out[i]);
        }
    }
}

// ---- fused row pass --------------------------------------------------------------------------------
struct RowParams {
    cf* data;          // planar rows, in place: row r at r * M
    const cf* chirp;   // plan order: chirp row (chan*N1 + k1) at that index * M, pre-scaled by 1/N
    const cf* tw16k;
    int64_t nrows;     // S * N1
    int N1, npol;
    int perm_w;        // chirp row order (see ChirpParams::perm_w); 8 selects k_row2
    unsigned* counter; // SP = 2: dynamic tile hand-out (zeroed before the launch); null = static stride
    int cdiv = 1;      // rows of one tile that share a chirp row: npol when the rows of a tile are consecutive SERIES (N1 = 1,
                       // one-tile blocks with many series), 1 when they are consecutive k1 of one series
    int cP = 1;        // the chirp rows are stored in the order of a column transform split cP x (N1/cP) while the data rows
                       // are in natural k1 order (one row per tile only): row k1 of the chirp is at (k1 % cP)*(N1/cP) + k1/cP
#ifdef PBH_DIAGNOSTIC
    unsigned long long* dbg;  // ABL = 4: [block][iteration < 64][8] s_memtime stamps
#endif
};

// Persistent, software-pipelined: a workgroup walks over tiles; while tile i is transformed the
// memory system works for it -- the chirp row is requested BEFORE the forward FFT (consumed after
// it) and tile i+1's samples are requested BEFORE the inverse FFT (consumed next iteration).  Both
// sets live in VGPRs (64 each) beside the 64 data registers; buffer loads survive the barriers.
// Tile order: the pols of one (channel, k1) run back to back on the same workgroup so the second
// one finds the chirp row in L2/MALL instead of HBM.
// ABL (diagnostic builds only): 0 = product kernel, 1 = no FFTs (memory traffic only),
// 2 = no chirp loads (constant multiplier), 3 = neither loads nor stores after the first tile,
// 4 = product arithmetic with forced waits and s_memtime stamps per phase (timeline of one wave),
// 5 = ablation 3 with the stamps (shader clock without memory traffic)
// SP = 1: the chirp loads, the next tile's loads and the stores are spread over the stages of the
// two transforms through fft_tile's hook instead of being issued in three bursts.
template <int M, int R, bool PF, int ABL = 0, int SP = 0>
__global__ __launch_bounds__(kTilePoints / R) void k_row(RowParams p) {
    constexpr int NST = stage_count(M, R);
    constexpr int FR = kTilePoints / M;  // rows per tile
    constexpr int MR = M / R;
    constexpr int STEP = MR * (int)sizeof(cf);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int tau = tid % MR, f = tid / MR;
    const int voff = (f * M + tau) * (int)sizeof(cf);
    const int cvoff = ((f / p.cdiv) * M + tau) * (int)sizeof(cf);   // chirp row of tile row f
    // tiles of several rows never straddle two series: a series of N1 rows is ceil(N1 / FR) tiles, the last one short when
    // N1 is not a multiple of FR (odd N1 of the 7-smooth lengths); N1 = 1 (rows = consecutive series) packs FR series per tile
    const bool by_series = FR > 1 && p.N1 > 1;
    const int64_t tps = by_series ? (p.N1 + FR - 1) / FR : 1;
    const int64_t ntile = by_series ? (p.nrows / p.N1) * tps : (p.nrows + FR - 1) / FR;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    // tile id -> first row.  FR == 1: enumerate (chan, k1) pairs outermost and pol innermost.
    auto first_row = [&](int64_t t) -> int64_t {
        if (FR == 1 && p.npol > 1) {
            const int64_t pair = t / p.npol;
            const int pol = (int)(t - pair * p.npol);
            const int64_t chan = pair / p.N1, k1 = pair - chan * p.N1;
            return (chan * p.npol + pol) * (int64_t)p.N1 + k1;
        }
        if (by_series) return (t / tps) * p.N1 + (t % tps) * FR;
        return t * FR;
    };
    auto rows_of = [&](int64_t r0) -> int64_t {   // rows of the tile that starts at row r0
        int64_t left = p.nrows - r0;
        if (by_series) left = p.N1 - r0 % p.N1;
        return left < FR ? left : FR;
    };
    auto data_rsrc = [&](int64_t r0) {
        return make_rsrc(p.data + r0 * M, (uint32_t)(rows_of(r0) * (int64_t)M * sizeof(cf)));
    };

    int64_t t = blockIdx.x;
    if (t >= ntile) return;
    int64_t r0 = first_row(t);
    rsrc_t rd = data_rsrc(r0);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff, i * STEP);
    // SP = 2: tiles after the first two of a workgroup come from an atomic counter, fetched one
    // iteration ahead and passed through an LDS slot (see k_colq)
    unsigned* slot = reinterpret_cast<unsigned*>(smem + lds_tile_bytes<true>());
    int64_t tnx = t + gridDim.x;

#ifdef PBH_DIAGNOSTIC
    int iter = 0;
    auto stamp = [&](int slot) {
        if constexpr (ABL >= 4) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (tid == 0 && iter < 64) p.dbg[((int64_t)blockIdx.x * 64 + iter) * 8 + slot] = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto wait_vm = [&](int left) {  // s_waitcnt vmcnt(left), gfx9 encoding
        if constexpr (ABL == 4) {
            __builtin_amdgcn_sched_barrier(0);
            if (left == 32) __builtin_amdgcn_s_waitcnt((32 & 0xF) | ((32 >> 4) << 14) | (0x7 << 4) | (0xF << 8));
            else __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#else
    auto stamp = [&](int) {};
    auto wait_vm = [&](int) {};
#endif
    while (true) {
        stamp(0);
        wait_vm(32);  // x of this tile (the previous tile's stores may still be in flight)
        stamp(1);
        // keep the twiddle-power trees inside the iteration: hoisted out of the loop they would
        // pin ~140 VGPRs (LICM), which is what the prefetch registers need
        launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
        unsigned fetched = (unsigned)(tnx + gridDim.x);
        if constexpr (SP == 2) {
            if (tid == 0 && p.counter) fetched = 2 * gridDim.x + atomicAdd(p.counter, 1u);
        }
        const int64_t srs = r0 / p.N1;
        const int k1d = (int)(r0 - srs * p.N1);
        const int k1 = p.cP > 1 ? (k1d % p.cP) * (p.N1 / p.cP) + k1d / p.cP : k1d;
        const rsrc_t rc = make_rsrc(p.chirp + ((srs / p.npol) * p.N1 + k1) * (int64_t)M,
                                    (uint32_t)((by_series ? rows_of(r0) : FR) * (int64_t)M * sizeof(cf)));
        if constexpr (PF) {
            cf c[R];
            if constexpr (SP == 2) {
                // chirp row requested two loads per tick (pinned), i.e. within the first 16 of the ~40 ticks
                int cnt = 0;
                auto hk = [&](auto st, auto q) {
                    if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                        __builtin_amdgcn_sched_barrier(0x38E);
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            if (cnt < R) { c[cnt] = buf_load(rc, cvoff, cnt * STEP); ++cnt; }
                        __builtin_amdgcn_sched_barrier(0x38E);
                    }
                };
                fft_tile<M, 1, R, -1, 1, true, false, false>(v, lds, tau, f * M, w, hk);
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (cnt < R) { c[cnt] = buf_load(rc, cvoff, cnt * STEP); ++cnt; }
                if (tid == 0) slot[0] = fetched;
            } else if constexpr (SP && NST >= 2) {
                // chirp row requested during the first NST-1 stages (consumed right after the last one)
                auto hk = [&](auto st, auto q) {
                    constexpr int ST = decltype(st)::value;
                    if constexpr (std::is_same<decltype(q), ic<-1>>::value && ST < NST - 1) {
#pragma unroll
                        for (int i = ST * R / (NST - 1); i < (ST + 1) * R / (NST - 1); ++i) c[i] = buf_load(rc, cvoff, i * STEP);
                    }
                };
                fft_tile<M, 1, R, -1, 1, true, false, false>(v, lds, tau, f * M, w, hk);
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i)
                    c[i] = (ABL == 2 || ABL == 3 || ABL == 5) ? make_cf(RC(0.999), RC(0.001) * i) : buf_load(rc, cvoff, i * STEP);
                if constexpr (ABL != 1) fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, f * M, w);
            }
            stamp(2);
            wait_vm(0);  // chirp row (and the previous stores)
            stamp(3);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], c[i]);
        } else {
            fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], buf_load(rc, cvoff, i * STEP));
        }

        const int64_t tn = SP == 2 ? tnx : t + gridDim.x;
        const bool more = tn < ntile;
        const int64_t rn = more ? first_row(tn) : r0;
        const rsrc_t rdn = more ? data_rsrc(rn) : make_rsrc(p.data, 0);
        if constexpr (PF) {
            cf nx[R];
            if constexpr (SP == 2) {
                constexpr int NBL = R / stage_radix(M, last_stage_ns(M, R), R);
                int cnt = 0;
                auto hk = [&](auto st, auto q) {
                    if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                        __builtin_amdgcn_sched_barrier(0x38E);
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            if (cnt < R) { nx[cnt] = buf_load(rdn, voff, cnt * STEP); ++cnt; }
                        __builtin_amdgcn_sched_barrier(0x38E);
                    } else if constexpr (!std::is_same<decltype(q), ic<-1>>::value) {
                        __builtin_amdgcn_sched_barrier(0x38E);
#pragma unroll
                        for (int u = 0; u < R / NBL; ++u) buf_store(rd, voff, (q + u * NBL) * STEP, v[q + u * NBL]);
                            __builtin_amdgcn_sched_barrier(0x38E);
                    }
                };
                fft_tile<M, 1, R, +1, 1, true, false, false>(v, lds, tau, f * M, w, hk);
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (cnt < R) { nx[cnt] = buf_load(rdn, voff, cnt * STEP); ++cnt; }
                if constexpr (NST < 2) __syncthreads();
                tnx = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)slot[0]);
                if constexpr (NST < 2) __syncthreads();   // (and nobody is still reading the slot when thread 0 writes it again)
            } else if constexpr (SP && NST >= 2) {
                // next tile requested stage by stage; outputs stored as the last stage produces them
                constexpr int NBL = R / stage_radix(M, last_stage_ns(M, R), R);  // butterflies per thread in the last stage
                auto hk = [&](auto st, auto q) {
                    constexpr int ST = decltype(st)::value;
                    if constexpr (std::is_same<decltype(q), ic<-1>>::value) {
#pragma unroll
                        for (int i = ST * R / NST; i < (ST + 1) * R / NST; ++i) nx[i] = buf_load(rdn, voff, i * STEP);
                    } else if constexpr (!std::is_same<decltype(q), tick_tag>::value) {
#pragma unroll
                        for (int u = 0; u < R / NBL; ++u) buf_store(rd, voff, (q + u * NBL) * STEP, v[q + u * NBL]);
                    }
                };
                fft_tile<M, 1, R, +1, 1, true, false, false>(v, lds, tau, f * M, w, hk);
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) nx[i] = (ABL == 3 || ABL == 5) ? v[i] : buf_load(rdn, voff, i * STEP);
                stamp(4);
                if constexpr (ABL != 1) fft_tile<M, 1, R, +1, 1, true>(v, lds, tau, f * M, w);
                stamp(5);
                if ((ABL != 3 && ABL != 5) || !more) {
#pragma unroll
                    for (int i = 0; i < R; ++i) buf_store(rd, voff, i * STEP, v[i]);
                }
            }
            stamp(6);
#ifdef PBH_DIAGNOSTIC
            ++iter;
#endif
            if (!more) break;
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = (ABL == 3 || ABL == 5) ? make_cf(v[i].x + nx[i].y, v[i].y) : nx[i];
        } else {
            fft_tile<M, 1, R, +1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
            for (int i = 0; i < R; ++i) buf_store(rd, voff, i * STEP, v[i]);
            if (!more) break;
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = buf_load(rdn, voff, i * STEP);
        }
        t = tn;
        r0 = rn;
        rd = rdn;
    }
}

// ---- fused row pass fed by the chirp's PHASE (float32 only, one row per tile) ---------------------------
// k_row reads the complex64 chirp row once per polarisation: 8 B per sample of traffic for 4 B of
// algorithmic need (the second read does not hit in L2: a CU streams ~12 MiB between the two uses).
// Here the unit of work is a (channel, k1) PAIR: the row's phase (4 B per bin, float32 revolutions,
// written by k_chirp next to the complex64 chirp) is loaded once, stays in 32 VGPRs while the
// polarisations are transformed one after the other, and cos / sin come from the hardware
// (v_cos_f32 / v_sin_f32 take revolutions; max error 1.4e-7, rms 4.9e-8: tools/micro/sincos.hip --
// the same size as the complex64 rounding of the reference's chirp).  Chirp traffic: 8 -> 2 B/sample.
// Pairs are handed out by an atomic counter (one pair ahead, through an LDS slot); the next tile's
// samples are requested two per tick of the inverse transform and the stores are pinned to the last
// stage's butterflies.
#ifndef PBH_F64
struct RowpParams {
    cf* data;            // planar rows, in place: row r at r * M
    const float* phase;  // plan order [chan][k1][k2], revolutions
    const cf* tw16k;
    int nchan, N1, npol;
    real scale;          // 1/N
    unsigned* counter;   // pair hand-out (zeroed before the launch); null = static stride
    int cP = 1;          // phase rows stored in split order cP x (N1/cP), data rows in natural order (RowParams::cP)
    int phase16 = 0;     // the phase rows are stored in k_rowp16's order (ChirpParams::phase16): k_rowp16 runs the pass
    cf* out = nullptr;   // k_rowp16: rows are stored here (same geometry) instead of in place -- the ping-pong schedule
    // k_rowp16<.., OTF>: the phase is not read but computed where it is used, in float64 as k_chirp does (aux_kernels.hpp):
    // f = chan_freq[chan] + bin * inv_ndt, phi = coeff f (1/f_ref - 1/f)^2 -- 4.8 -> 4.3 GB per launch for ~15 float64
    // instructions per bin and pol pair
    const double* chan_freq = nullptr;
    double coeff = 0, inv_ndt = 0, inv_ref = 0;
    int64_t N = 0;
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_rowp(RowpParams p) {
    // A tile is FR = 2^tile / M consecutive rows (k1 values) of one series: one contiguous block of 2^tile elements, like the
    // one-row tile of M = 2^tile; the unit of work is the block of a (channel, k1 group) with its polarisations one after
    // the other, the phase block (2^tile floats, same order) loaded once.  N1 must be a multiple of FR.
    constexpr int FR = kTilePoints / M;
    constexpr int MR = M / R;
    constexpr int STEP = MR * (int)sizeof(cf), PSTEP = MR * (int)sizeof(float);
    constexpr int NBL = R / stage_radix(M, last_stage_ns(M, R), R);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    unsigned* slot = reinterpret_cast<unsigned*>(smem + lds_tile_bytes<true>());

    const int tau = threadIdx.x % MR, frow = threadIdx.x / MR;
    const int fofs = frow * M;
    const int voff = (fofs + tau) * (int)sizeof(cf), pvoff = (fofs + tau) * (int)sizeof(float);
    const uint32_t NB = ((uint32_t)p.N1 + FR - 1) / FR;       // k1 groups per series (the last one short when FR does not divide N1)
    const uint32_t npair = (uint32_t)p.nchan * NB;
    const uint32_t G = gridDim.x;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    auto row_rsrc = [&](uint32_t u, int pol) {
        if (u >= npair) return make_rsrc(p.data, 0);
        const uint32_t chan = u / NB, kb = u - chan * NB;
        const int64_t row = ((int64_t)chan * p.npol + pol) * p.N1 + (int64_t)kb * FR;
        const uint32_t rows = (uint32_t)p.N1 - kb * FR < (uint32_t)FR ? (uint32_t)p.N1 - kb * FR : (uint32_t)FR;
        return make_rsrc(p.data + row * M, (uint32_t)(rows * M * sizeof(cf)));
    };

    uint32_t u = blockIdx.x;
    if (u >= npair) return;
    uint32_t unx = u + G;   // pair after this one
    int pol = 0;
    rsrc_t rd = row_rsrc(u, 0);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff, i * STEP);
    float ph[R];
    unsigned fetched = 0;

    while (true) {
        launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
        if (pol == 0) {   // wave-uniform: a new pair -- its phase row, and the index of the pair after next
            uint32_t up = u;
            if (FR == 1 && p.cP > 1) {
                const uint32_t ch = u / (uint32_t)p.N1, k1d = u - ch * (uint32_t)p.N1;
                up = ch * (uint32_t)p.N1 + (k1d % (uint32_t)p.cP) * (uint32_t)(p.N1 / p.cP) + k1d / (uint32_t)p.cP;
            }
            const uint32_t pch = up / NB, pkb = up - pch * NB;   // phase rows of the group: (chan N1 + kb FR) .. + rows
            const uint32_t prows = (uint32_t)p.N1 - pkb * FR < (uint32_t)FR ? (uint32_t)p.N1 - pkb * FR : (uint32_t)FR;
            const rsrc_t rp = make_rsrc(p.phase + ((int64_t)pch * p.N1 + (int64_t)pkb * FR) * M, (uint32_t)(prows * M * sizeof(float)));
#pragma unroll
            for (int i = 0; i < R; ++i)
                ph[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, pvoff, i * PSTEP, 0));
            fetched = unx + G;
            if (threadIdx.x == 0 && p.counter) fetched = 2 * G + atomicAdd(p.counter, 1u);
        }
        fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, fofs, w);
        if (pol == 0 && threadIdx.x == 0) slot[0] = fetched;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const cf c = make_cf(__builtin_amdgcn_cosf(ph[i]) * p.scale, __builtin_amdgcn_sinf(ph[i]) * p.scale);
            v[i] = cmul(v[i], c);
        }

        const bool last_pol = pol == p.npol - 1;
        const uint32_t un = last_pol ? unx : u;
        const int poln = last_pol ? 0 : pol + 1;
        const bool more = un < npair;
        const rsrc_t rdn = row_rsrc(un, poln);
        cf nx[R];
        int cnt = 0;
        auto hk = [&](auto st, auto q) {
            if constexpr (std::is_same<decltype(q), tick_tag>::value) {
                __builtin_amdgcn_sched_barrier(0x38E);
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (cnt < R) { nx[cnt] = buf_load(rdn, voff, cnt * STEP); ++cnt; }
                __builtin_amdgcn_sched_barrier(0x38E);
            } else if constexpr (!std::is_same<decltype(q), ic<-1>>::value) {
                __builtin_amdgcn_sched_barrier(0x38E);
#pragma unroll
                for (int k = 0; k < R / NBL; ++k) buf_store(rd, voff, (q + k * NBL) * STEP, v[q + k * NBL]);
                __builtin_amdgcn_sched_barrier(0x38E);
            }
        };
        fft_tile<M, 1, R, +1, 1, true, false, false>(v, lds, tau, fofs, w, hk);
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (cnt < R) { nx[cnt] = buf_load(rdn, voff, cnt * STEP); ++cnt; }
        if (!more) break;
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = nx[i];
        if (last_pol) {
            // The slot is written by thread 0 after the forward transform and read here by every wave.  With two or more
            // stages the inverse transform's exchange barriers lie between the two; a 32-point row (7-smooth plans with five
            // factors of two) is ONE stage without any exchange, and the waves raced: pairs handed out by the counter were
            // transformed twice or not at all (found by tests/tools/fuzz_parity.py, seed 7: 1 372 000 samples x 9 channels).
            if constexpr (stage_count(M, R) < 2) __syncthreads();
            u = unx;
            unx = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[0]);
            if constexpr (stage_count(M, R) < 2) __syncthreads();   // ... and everyone has read it before it is written again
        }
        pol = poln;
        rd = rdn;
    }
}
// ---- the same pass with 16-byte accesses ------------------------------------------------------------------------------
// k_rowp moves its rows with 8 bytes per lane (the tile FFT leaves thread tau with points tau + 512 i).  One persistent
// 512-thread workgroup per CU streaming 128-KiB rows that way needs 1.05 ms for the pass's 4.3 GB WITHOUT any transform
// (tools/micro/rowcopy.hip), which is what k_rowp takes; with 16 bytes per lane the same bytes move in 0.82 ms.  Here the
// forward transform runs its stages as 16, 32, 32 with the two radix-16 butterflies of a thread starting from ADJACENT
// points (2 tau, 2 tau + 1) + 1024 j, and the inverse transform (32, 32, 16) ends with the same assignment: a row is
// loaded and stored as 16 x 16 bytes per thread instead of 32 x 8.  Between the two transforms the points sit in the
// natural distribution tau + 512 i, which is the order of the phase row.  (M = 2^14, float32.)
// ABL (experiments): 1 = no transforms (memory traffic and the loop only)
// DS: deferred stores -- a tile's results are not stored in a burst behind its inverse transform (16 back-to-back 16-byte
// stores per thread during which the wave issues nothing else) but stay in registers and leave one per tick of the NEXT tile's
// forward transform, in the registers the next tile's samples vacated (k_colq's scheme; the loads already ride in the inverse
// transform's ticks): memory traffic in both halves of an iteration, no burst.
template <int R, int ABL = 0, bool DS = false, bool OTF = false>
__global__ __launch_bounds__(kTilePoints / R) void k_rowp16(RowpParams p) {
    constexpr int M = kTilePoints;
    static_assert(M == 16384 && R == 32, "k_rowp16: 2^14-point rows, 32 points per thread");
    constexpr int MR = M / R;
    constexpr int PSTEP = MR * (int)sizeof(float);
    constexpr int STEP16 = (M / 16) * (int)sizeof(cf);          // 1024 elements between a thread's 16-byte pairs
    constexpr int NSF = tw_seeds_or1(M, R, 1, false), NSI = tw_seeds_or1(M, R, 0, true);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    unsigned* slot = reinterpret_cast<unsigned*>(smem + lds_tile_bytes<true>());

    const int tau = threadIdx.x;
    const int voff16 = tau * 2 * (int)sizeof(cf), pvoff = tau * (int)sizeof(float);
    const uint32_t npair = (uint32_t)p.nchan * (uint32_t)p.N1;
    const uint32_t G = gridDim.x;

    cf wf[NSF], wi[NSI];
    load_tw_seeds<M, 1, R, 1, false>(wf, tau, p.tw16k);
    load_tw_seeds<M, 1, R, 0, true>(wi, tau, p.tw16k);

    auto row_rsrc = [&](uint32_t u, int pol) {
        if (u >= npair) return make_rsrc(p.data, 0);
        const uint32_t chan = u / (uint32_t)p.N1, k1 = u - chan * (uint32_t)p.N1;
        const int64_t row = ((int64_t)chan * p.npol + pol) * p.N1 + k1;
        return make_rsrc(p.data + row * M, (uint32_t)(M * sizeof(cf)));
    };
    cf* const obase = p.out ? p.out : p.data;
    auto out_rsrc = [&](uint32_t u, int pol) {   // where the tile's rows are stored (the row itself unless p.out is set)
        if (u >= npair) return make_rsrc(obase, 0);
        const uint32_t chan = u / (uint32_t)p.N1, k1 = u - chan * (uint32_t)p.N1;
        const int64_t row = ((int64_t)chan * p.npol + pol) * p.N1 + k1;
        return make_rsrc(obase + row * M, (uint32_t)(M * sizeof(cf)));
    };
    auto load_pair = [&](rsrc_t r, int j, cf& a, cf& b) {   // points (2 tau, 2 tau + 1) + 1024 j
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff16, j * STEP16, 0);
        a = make_cf(__uint_as_float(x.x), __uint_as_float(x.y));
        b = make_cf(__uint_as_float(x.z), __uint_as_float(x.w));
    };
    auto store_pair = [&](rsrc_t r, int j, cf a, cf b) { buf_store_pair(r, voff16, j * STEP16, a, b); };   // (hazard wait states: fft_core.hpp)

    uint32_t u = blockIdx.x;
    if (u >= npair) return;
    uint32_t unx = u + G;
    int pol = 0;
    rsrc_t rd = row_rsrc(u, 0);
    rsrc_t ws = out_rsrc(u, 0);
    cf v[R];
#pragma unroll
    for (int j = 0; j < R / 2; ++j) load_pair(rd, j, v[2 * j], v[2 * j + 1]);
    // the first tile's samples are waited for HERE: a wait at the loop header would be the merge of this path (16 loads pending)
    // and the back edge (only the previous tile's 16 stores pending) -- vmcnt(0), every iteration waiting for its stores' acknowledgements
    __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));   // vmcnt(0): the first tile's samples and the twiddle seeds
    launder_all(v, std::make_integer_sequence<int, R>{});
    launder_all(wf, std::make_integer_sequence<int, NSF>{});
    launder_all(wi, std::make_integer_sequence<int, NSI>{});
    float ph[R];
    unsigned fetched = 0;
    cf out[R];                              // DS: the previous tile's results, waiting to be stored
    rsrc_t wso = make_rsrc(obase, 0);       // ... and where they go (nothing yet: a zero-size descriptor drops the stores)
    if constexpr (DS) {
#pragma unroll
        for (int i = 0; i < R; ++i) out[i] = make_cf(0, 0);
    }

    while (true) {
        launder_all(wf, std::make_integer_sequence<int, NSF>{});
        launder_all(wi, std::make_integer_sequence<int, NSI>{});
        // the wait for this tile's samples (requested during the previous inverse transform) belongs HERE, in front of the
        // phase loads of a new pair: left to the first butterfly it lands behind them and waits for them as well
        launder_all(v, std::make_integer_sequence<int, R>{});
        if (pol == 0) {
            uint32_t up = u;
            if (p.cP > 1) {
                const uint32_t ch = u / (uint32_t)p.N1, k1d = u - ch * (uint32_t)p.N1;
                up = ch * (uint32_t)p.N1 + (k1d % (uint32_t)p.cP) * (uint32_t)(p.N1 / p.cP) + k1d / (uint32_t)p.cP;
            }
            if constexpr (OTF) {
                // this thread's bins of the row: k2 = tau + 512 i, k = k1 + N1 k2, in numpy.fft.fftfreq order
                const uint32_t ch = up / (uint32_t)p.N1, k1 = up - ch * (uint32_t)p.N1;
                const double fc = p.chan_freq[ch];
                const int half = (int)((p.N - 1) / 2), n = (int)p.N;   // (N < 2^31: host-checked)
                int t2 = tau;
                asm volatile("" : "+v"(t2));   // not loop-invariant for the compiler: else 32 bin indices are hoisted out of the persistent loop
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int k = (int)k1 + p.N1 * (t2 + MR * i);
                    const double f = fma((double)(k <= half ? k : k - n), p.inv_ndt, fc);
                    double r = __builtin_amdgcn_rcp(f);          // + two Newton steps: full float64 accuracy
                    r = fma(r, fma(-f, r, 1.0), r);
                    r = fma(r, fma(-f, r, 1.0), r);
                    const double dd = p.inv_ref - r;
                    const double phi = (p.coeff * f) * (dd * dd);
                    ph[i] = (float)(__builtin_rint(phi) - phi);   // chirp = exp(2 pi i ph), ph = -(phi mod 1)
                    // four independent chains at a time: left alone the scheduler interleaves all 32 and spills 220 B/lane
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            const rsrc_t rp = make_rsrc(p.phase + (int64_t)up * M, (uint32_t)(M * sizeof(float)));
#pragma unroll
            for (int j = 0; j < R / 4; ++j) {   // the row is stored in this order (ChirpParams::phase16): 8 x 16 bytes per thread
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (4 * MR * (int)sizeof(float)), 0);
                ph[4 * j] = __uint_as_float(x.x);
                ph[4 * j + 1] = __uint_as_float(x.y);
                ph[4 * j + 2] = __uint_as_float(x.z);
                ph[4 * j + 3] = __uint_as_float(x.w);
            }
            }
        }
        // forward: 16 (pair-adjacent bases), 32, 32 -> natural distribution tau + 512 i
        if constexpr (DS) {
            int scnt = 0;
            auto hkf = [&](auto st, auto q) {
                if constexpr (std::is_same<decltype(q), tick_tag>::value) {   // the previous tile: one 16-byte store per tick
                    __builtin_amdgcn_sched_barrier(0x38E);
                    if (scnt < R / 2) { store_pair(wso, scnt, out[2 * scnt], out[2 * scnt + 1]); ++scnt; }
                    __builtin_amdgcn_sched_barrier(0x38E);
                }
            };
            if constexpr (ABL != 1) fft_tile<M, 1, R, -1, 1, true, false, false, decltype(hkf), 1, true, false>(v, lds, tau, 0, wf, hkf);
#pragma unroll
            for (int k = 0; k < R / 2; ++k)
                if (scnt < R / 2) { store_pair(wso, scnt, out[2 * scnt], out[2 * scnt + 1]); ++scnt; }
        } else {
            if constexpr (ABL != 1) fft_tile<M, 1, R, -1, 1, true, false, false, NoHook, 1, true, false>(v, lds, tau, 0, wf);
        }
        if (pol == 0) {   // the index of the pair after next: the atomic's round trip is waited for where the phase values are
                          // needed anyway (nothing else is in flight here), not at the top of the loop
            fetched = unx + G;
            if (tau == 0 && p.counter) fetched = 2 * G + atomicAdd(p.counter, 1u);
            if (tau == 0) slot[0] = fetched;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const cf c = make_cf(__builtin_amdgcn_cosf(ph[i]) * p.scale, __builtin_amdgcn_sinf(ph[i]) * p.scale);
            v[i] = cmul(v[i], c);
        }

        const bool last_pol = pol == p.npol - 1;
        const uint32_t un = last_pol ? unx : u;
        const int poln = last_pol ? 0 : pol + 1;
        const bool more = un < npair;
        const rsrc_t rdn = row_rsrc(un, poln);
        cf nx[R];
        int cnt = 0;
        auto hk = [&](auto st, auto q) {
            if constexpr (std::is_same<decltype(q), tick_tag>::value) {   // the next tile: one 16-byte load per tick
                __builtin_amdgcn_sched_barrier(0x38E);
                if (cnt < R / 2) { load_pair(rdn, cnt, nx[2 * cnt], nx[2 * cnt + 1]); ++cnt; }
                __builtin_amdgcn_sched_barrier(0x38E);
            }
        };
        // inverse: 32, 32, 16 ending on pair-adjacent bases: v[q + 2 u] = X[2 tau + q + 1024 u]
        if constexpr (ABL != 1) fft_tile<M, 1, R, +1, 1, true, false, false, decltype(hk), 0, false, true>(v, lds, tau, 0, wi, hk);
        else __syncthreads();
        if (!DS || !more) {   // (the last tile of a workgroup has no successor to hide its stores behind)
#pragma unroll
            for (int j = 0; j < R / 2; ++j) store_pair(ws, j, v[2 * j], v[2 * j + 1]);
        }
#pragma unroll
        for (int k = 0; k < R / 2; ++k)
            if (cnt < R / 2) { load_pair(rdn, cnt, nx[2 * cnt], nx[2 * cnt + 1]); ++cnt; }
        if (!more) break;
        if constexpr (DS) {
#pragma unroll
            for (int i = 0; i < R; ++i) out[i] = v[i];
            wso = ws;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = nx[i];
        if (last_pol) {
            u = unx;
            unx = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot[0]);
        }
        pol = poln;
        rd = rdn;
        ws = out_rsrc(un, poln);
    }
}
#endif  // !PBH_F64

// ---- stand-alone row FFT, in place (the Bluestein kernel's spectrum in plan order; DIR = +1: the unscaled inverse) ------
template <int M, int R, int DIR = -1>
__global__ __launch_bounds__(kTilePoints / R) void k_rowfft(cf* data, const cf* tw16k, int64_t nrows) {
    constexpr int FR = kTilePoints / M, MR = M / R, STEP = MR * (int)sizeof(cf);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int tau = tid % MR, f = tid / MR;
    const int64_t r0 = (int64_t)blockIdx.x * FR;
    const int64_t left = nrows - r0;
    const rsrc_t rd = make_rsrc(data + r0 * M, (uint32_t)((left < FR ? left : FR) * (int64_t)M * sizeof(cf)));
    const int voff = (f * M + tau) * (int)sizeof(cf);
    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, tw16k);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff, i * STEP);
    fft_tile<M, 1, R, DIR, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
    for (int i = 0; i < R; ++i) buf_store(rd, voff, i * STEP, v[i]);
}

#ifndef PBH_F64
// ---- fused row pass, wave-decoupled form (M = 16384 = 8 x 2048) ------------------------------------------
// k = ka + 8 kb, n = 2048 na + nb.  Forward: radix-8 over na inside each thread (its 32 points
// tau + 512 i contain na = 0..7 for four values of nb), twiddle W_M^{nb ka}, ONE cross-wave exchange
// that hands wave `ka` the 2048 points A[ka][.], then every wavefront runs its own 2048-point FFT
// through a private LDS region with wave-level synchronisation only.  The inverse mirrors it.  Per
// tile: 4 workgroup barriers instead of 8, and between them the eight waves run decoupled, so one
// wave's LDS exchange overlaps another's butterflies.  The chirp row is stored in the matching
// order (position ka*2048 + kb holds bin ka + 8 kb; ChirpParams::perm_w).
template <bool PF>
__global__ __launch_bounds__(512) void k_row2(RowParams p) {
    constexpr int M = 16384, R = 32, MR = 512, MW = 2048;
    constexpr int STEP = 64 * (int)sizeof(cf);   // a wave's points: lane + 64 i
    constexpr int STEPT = MR * (int)sizeof(cf);  // a thread's natural points: tau + 512 i
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tau = threadIdx.x, lane = tau & 63, wave = tau >> 6;
    const int voff_t = tau * (int)sizeof(cf);
    const int voff_w = (wave * MW + lane) * (int)sizeof(cf);
    const int64_t ntile = p.nrows;

    cf wq[4];  // W_M^{nb}, nb = tau + 512 c
#pragma unroll
    for (int c = 0; c < 4; ++c) wq[c] = p.tw16k[tau + MR * c];
    cf w[tw_seeds_or1(MW, R)];
    load_tw_seeds<MW, 1, R>(w, lane, p.tw16k);

    auto first_row = [&](int64_t t) -> int64_t {
        if (p.npol > 1) {
            const int64_t pair = t / p.npol;
            const int pol = (int)(t - pair * p.npol);
            const int64_t chan = pair / p.N1, k1 = pair - chan * p.N1;
            return (chan * p.npol + pol) * (int64_t)p.N1 + k1;
        }
        return t;
    };
    constexpr uint32_t ROWB = (uint32_t)(M * sizeof(cf));

    int64_t t = blockIdx.x;
    if (t >= ntile) return;
    int64_t r0 = first_row(t);
    rsrc_t rd = make_rsrc(p.data + r0 * M, ROWB);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff_t, i * STEPT);

    while (true) {
        launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(MW, R)>{});
        launder_all(wq, std::make_integer_sequence<int, 4>{});
        const int64_t srs = r0 / p.N1;
        const int k1 = (int)(r0 - srs * p.N1);
        const rsrc_t rc = make_rsrc(p.chirp + ((srs / p.npol) * p.N1 + k1) * (int64_t)M, ROWB);
        cf c[PF ? R : 1];
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < R; ++i) c[i] = buf_load(rc, voff_w, i * STEP);
        }

        // ---- forward: radix-8 over na, twiddle, cross-wave exchange ----
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            cf t8[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) t8[a] = v[cc + 4 * a];
            Dft<8, -1>::run(t8);
            apply_powers<8>(t8, wq[cc]);
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) lds[ka * MW + tau + MR * cc] = t8[ka];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = lds[wave * MW + lane + 64 * i];
        __syncthreads();
        fft_tile<MW, 1, R, -1, 1, true, false, true>(v, lds, lane, wave * MW, w);

        // ---- chirp ----
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], c[i]);
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], buf_load(rc, voff_w, i * STEP));
        }

        const int64_t tn = t + gridDim.x;
        const bool more = tn < ntile;
        const int64_t rn = more ? first_row(tn) : r0;
        const rsrc_t rdn = more ? make_rsrc(p.data + rn * M, ROWB) : make_rsrc(p.data, 0);
        cf nx[PF ? R : 1];
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < R; ++i) nx[i] = buf_load(rdn, voff_t, i * STEPT);
        }

        // ---- inverse: per-wave 2048-point IFFT, cross-wave exchange back, twiddle, radix-8 ----
        fft_tile<MW, 1, R, +1, 1, true, false, true>(v, lds, lane, wave * MW, w);
        __syncthreads();  // every wave is done with its private (padded) region
#pragma unroll
        for (int i = 0; i < R; ++i) lds[wave * MW + lane + 64 * i] = v[i];
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            cf t8[8];
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) t8[ka] = lds[ka * MW + tau + MR * cc];
            apply_powers<8>(t8, cconj(wq[cc]));
            Dft<8, +1>::run(t8);
#pragma unroll
            for (int a = 0; a < 8; ++a) v[cc + 4 * a] = t8[a];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) buf_store(rd, voff_t, i * STEPT, v[i]);
        if (!more) break;
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = nx[i];
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = buf_load(rdn, voff_t, i * STEPT);
        }
        __syncthreads();  // the cross buffer is rewritten by the next tile's forward pass
        t = tn;
        r0 = rn;
        rd = rdn;
    }
}

#endif  // !PBH_F64

// ---- single-tile transform (nsample = M <= 2^14) -------------------------------------------------------
struct SmallParams {
    const cf* in;     // (M, S) interleaved
    cf* out;          // (stop-start, S) interleaved
    const cf* chirp;  // (nchan, M), pre-scaled by 1/M; nullptr: plain FFT (no product)
    const cf* tw16k;
    int S, npol;
    int64_t crop_start, crop_stop;
    int dir;      // plain-FFT mode only: -1 forward, +1 inverse
    real scale;   // plain-FFT mode only
    // contrib.stft / istft (pulsarbat/contrib/misc.py:17-93): blockIdx.y = segment, a segment is an
    // (M, S) block of the time-ordered side; the channelised side is (nchan*M, E) per segment with
    // bin k of channel c at row c*M + ((k + M/2) % M)  (fftshift; for M = 2^m that is k ^ M/2).
    int seg_mode = 0;  // 0: plain block;  1: stft (shifted store, * scale);  2: istft (shifted load)
    int E = 1;         // inner elements per channel (npol * ...): series q = c*E + e
    int segs = 1;      // segments per tile (when the tile's F columns exceed S): column f = (f / S, f % S)
    int64_t nseg = 1;  // total segments (tail tiles are masked)
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_small(SmallParams p) {
    constexpr int F = kTilePoints / M;
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    // column -> (segment within the tile, series).  Plain blocks and large M: one segment per tile
    // and blockIdx.x walks the series; small M with segments: the tile spans `segs` whole segments.
    int sl = 0;
    int64_t q = (int64_t)blockIdx.x * F + f;
    bool valid = q < p.S;
    if (p.segs > 1) {
        sl = f / p.S;
        q = f - sl * p.S;
        valid = sl < p.segs && (int64_t)blockIdx.y * p.segs + sl < p.nseg;
    }
    const int64_t qq = valid ? q : 0;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    // buffer addressing: wave-uniform descriptors over the tile's segments, per-lane offset
    // (sl*M*S + tau*S + q)*8, scalar step MR*S*8; invalid lanes get an out-of-range offset
    // (reads return 0, writes are dropped)
    const uint32_t in_bytes = (uint32_t)((int64_t)p.segs * M * p.S * sizeof(cf));
    const int64_t seg_off = (int64_t)blockIdx.y * p.segs * M * p.S;  // segments are consecutive (M, S) blocks
    const rsrc_t ri = make_rsrc(p.in + seg_off, in_bytes);
    const int oob = 0x7ffffff0;
    const int64_t lane_seg = (int64_t)sl * M * p.S;
    const int voff = valid ? (int)((lane_seg + tau * (int64_t)p.S + qq) * sizeof(cf)) : oob;
    const int step = MR * p.S * (int)sizeof(cf);
    // channelised-side addressing: row r of series q = (c, e) at ((c*M + r)*E + e)
    const int cq = (int)(qq / p.E), eq = (int)(qq - (int64_t)cq * p.E);
    const int voff_ch = valid ? (int)((lane_seg + ((int64_t)cq * M + tau) * p.E + eq) * sizeof(cf)) : oob;
    const int step_ch = MR * p.E * (int)sizeof(cf);
    cf v[R];
    if (p.seg_mode == 2) {
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = buf_load(ri, voff_ch, (i ^ (R / 2)) * step_ch);  // ifftshift
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = buf_load(ri, voff, i * step);
    }

    if (p.chirp) {
        const rsrc_t rc = make_rsrc(p.chirp + (qq / p.npol) * (int64_t)M, (uint32_t)(M * sizeof(cf)));
        const int coff = tau * (int)sizeof(cf);
        fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = cmul(v[i], buf_load(rc, coff, i * MR * (int)sizeof(cf)));
        fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
    } else if (p.dir < 0) {
        fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
    } else {
        fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = make_cf(v[i].x * p.scale, v[i].y * p.scale);
    }
    if (p.seg_mode == 1) {  // stft: fftshift + 1/M on the way out
        const rsrc_t rs = make_rsrc(p.out + seg_off, in_bytes);
#pragma unroll
        for (int i = 0; i < R; ++i)
            buf_store(rs, voff_ch, (i ^ (R / 2)) * step_ch, make_cf(v[i].x * p.scale, v[i].y * p.scale));
        return;
    }
    // output rows [crop_start, crop_stop) -> out row (row - crop_start)
    const int64_t nout_rows = p.crop_stop - p.crop_start;
    const rsrc_t ro = make_rsrc(p.out + seg_off, (uint32_t)(((int64_t)(p.segs - 1) * M + nout_rows) * p.S * sizeof(cf)));
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t row = tau + i * MR;
        const bool keep = valid && row >= p.crop_start && row < p.crop_stop;
        const int off = keep ? (int)((lane_seg + (row - p.crop_start) * p.S + qq) * sizeof(cf)) : oob;
        buf_store(ro, off, 0, v[i]);
    }
}

// ---- contrib.stft, forward, with a linearised store ---------------------------------------------------------------------
// k_small's stft mode stores every bin from the lane that computed it: 8- or 16-byte pieces 1 KiB apart, a 128-byte line
// of the channelised block filled by eight different waves.  The lines do not always survive in L2 until they are
// complete: rocprofv3 counts 6.2 GB of HBM traffic for 4.3 GB of algorithmic bytes (nperseg 64, 16 series).  But a
// tile's output is ONE contiguous block of the channelised array -- G whole segments x all series (tile/M >= S), or all
// bins of tile/M series of one segment -- so the outputs take one more trip through LDS, laid out in output order, and
// leave as 16-byte stores of consecutive addresses.
struct StftFwdParams {
    const cf* in;      // (nseg*M, S) sample-major
    cf* out;           // (nseg, nchan*M, E) sample-major
    const cf* tw16k;
    int S, E;          // series (nchan*E), inner elements per channel
    int SB, G;         // series per tile (S when a tile spans whole segments), segments per tile; SB*G <= tile/M
    int64_t nseg;
    real scale;        // 1/M
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_stft_fwd(StftFwdParams p) {
    constexpr int F = kTilePoints / M;
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;
    constexpr int NT = kTilePoints / R;
    static_assert(F >= 16 && F <= NT, "k_stft_fwd: segment length out of range");
    typedef real vec4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const int j = f % p.SB, sl = f / p.SB;                 // series within the tile's subset, segment within the tile
    const bool col_ok = sl < p.G;                          // (SB*G may be less than F when S is not a power of two)
    const int q0 = blockIdx.x * p.SB;
    const int64_t g0 = (int64_t)blockIdx.y * p.G;
    const bool valid = col_ok && g0 + sl < p.nseg;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    const int64_t nleft = p.nseg - g0 < p.G ? p.nseg - g0 : p.G;
    const rsrc_t ri = make_rsrc(p.in + (g0 * M) * (int64_t)p.S + q0, (uint32_t)((nleft * M * p.S - q0) * (int64_t)sizeof(cf)));
    const int voff = valid ? (int)(((int64_t)sl * M * p.S + (int64_t)tau * p.S + j) * (int64_t)sizeof(cf)) : 0x7ffffff0;
    const int step = MR * p.S * (int)sizeof(cf);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(ri, voff, i * step);
    fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
    __syncthreads();
    // staging in output order: element (sl, channel cl of the subset, shifted bin ks, e) at ((sl*CB + cl)*M + ks)*E + e, every
    // channel block (M*E elements) shifted by 2 slots so that the 32 lanes of a half-wave fall into different banks
    const int CB = p.SB / p.E;                             // channels in the tile's subset
    const int cl = j / p.E, e = j - cl * p.E;
    const int blk = sl * CB + cl;
    cf* mine = lds + (int64_t)blk * (M * p.E + 2) + e;
    if (col_ok) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int k = tau + i * MR;
            mine[(k ^ (M / 2)) * p.E] = make_cf(v[i].x * p.scale, v[i].y * p.scale);
        }
    }
    __syncthreads();
    // the tile's output: for every segment of the tile, CB channels x M bins x E elements, contiguous from
    // ((g0 + sl)*nchan + c0)*M*E;  linear element o of the tile = (segment ss, offset within its CB*M*E run)
    const int run = CB * M * p.E;                          // contiguous elements per segment
    const int total = (int)nleft * run;
    const int64_t seg_pitch = (int64_t)p.S * M;            // elements per segment of the channelised block
    cf* obase = p.out + g0 * seg_pitch + (int64_t)(q0 / p.E) * M * p.E;
    for (int o = 2 * tid; o < total; o += 2 * NT) {        // two elements (16 bytes) per thread and round
        const int ss = o / run, r = o - ss * run;
        const int b = ss * CB + r / (M * p.E);
        const cf* src = lds + (int64_t)b * (M * p.E + 2) + (r % (M * p.E));
        const cf a0 = src[0], a1 = src[1];
        vec4 x;
        x[0] = a0.x; x[1] = a0.y; x[2] = a1.x; x[3] = a1.y;
        *reinterpret_cast<vec4*>(obase + (int64_t)ss * seg_pitch + r) = x;
    }
}

// (The mirrored form for istft -- linear 16-byte loads of the channelised block into LDS -- was measured and is slower than
//  k_small's direct loads, 0.75 vs 0.60 ms at nperseg 32: partial-line READS are cheap, it is the partial-line writes that
//  cost; istft keeps k_small.)

// ---- contrib.stft written series-major: the channeliser in front of coherent_dedispersion ----------------------------
// stft -> coherent_dedispersion is the typical pipeline: the channelised block is "hundreds of
// narrow series" that the dedispersion's first pass would de-interleave again.  This kernel writes the channeliser's
// output straight into the dedispersion plan's planar work buffer (series (c*M + shifted k)*E + e at q'*plane, time =
// segment index), so the dedispersion starts at its column pass: one full read + write of the block less.
// A tile is M (bins) x G (consecutive segments) x SB (a subset of the input's series), G*SB*M = 2^14 points: with
// G >= 16 every (series', bin) row receives >= 128 contiguous bytes per tile.  The G*SB column FFTs are interleaved in
// LDS as in k_small; after the last stage the outputs take one more trip through LDS to put the segment index on the
// lanes.  Sibling tiles (the other series subsets of the same segments, which share the input's 128-byte lines) are
// adjacent in blockIdx.x.
struct StftPlanarParams {
    const cf* in;      // (nseg*M, S) sample-major
    cf* out;           // planar: series' q' = (c*M + ((k + M/2) % M))*E + e at q'*plane + segment
    const cf* tw16k;
    int64_t plane;
    int S, E;          // series of the input (nchan*E), inner elements per channel
    int SB, G;         // series per tile, segments per tile (SB*G = tile/M)
    real scale;        // 1/M
    int nsub = 1;      // sibling tiles (the other series subsets of the same segments) a workgroup does one after the other
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_stft_planar(StftPlanarParams p) {
    constexpr int F = kTilePoints / M;
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;
    constexpr int NT = kTilePoints / R;      // threads
    static_assert(F <= NT && F >= 16, "k_stft_planar: segment length out of range for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const int j = f % p.SB, sl = f / p.SB;
    const int64_t g0 = (int64_t)blockIdx.y * p.G;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    // Sibling tiles read the SAME 128-byte lines of the input (SB of its S series each).  As separate workgroups they land on
    // different XCDs (consecutive workgroups are dealt round the eight of them) and every L2 fetches the lines again: with
    // SB < 8 the input crossed the fabric S/SB times and the fused call lost to two steps.  One workgroup doing the siblings
    // one after the other re-reads the lines out of its own L2 a few microseconds later.
    for (int sub = 0; sub < p.nsub; ++sub) {
    launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
    const int q0 = (blockIdx.x * p.nsub + sub) * p.SB;
    if (sub) __syncthreads();   // the previous sibling's staging reads are done
    const rsrc_t ri = make_rsrc(p.in + (g0 * M) * (int64_t)p.S + q0, (uint32_t)(((int64_t)p.G * M * p.S - q0) * (int64_t)sizeof(cf)));
    const int voff = (int)(((int64_t)sl * M * p.S + (int64_t)tau * p.S + j) * (int64_t)sizeof(cf));
    const int step = MR * p.S * (int)sizeof(cf);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(ri, voff, i * step);
    fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
    __syncthreads();   // every wave is done reading the last exchange
    // bin k = tau + i*MR of column (sl, j)  ->  staging slot (k*SB + j)*G + sl
#pragma unroll
    for (int i = 0; i < R; ++i) lds[((tau + i * MR) * p.SB + j) * p.G + sl] = v[i];
    __syncthreads();
    // thread tid, round i: slot tid + i*NT = (k*SB + jj)*G + ss with ss fastest across lanes.  NT is a multiple of
    // F = SB*G, so ss and jj are the thread's own and only the bin advances, by NT/F per round: no division in the loop
    {
        const int ss = tid % p.G, rest = tid / p.G;
        const int jj = rest % p.SB, k0 = rest / p.SB;
        const int q = q0 + jj;
        const int c = q / p.E, e = q - c * p.E;
        cf* base = p.out + ((int64_t)c * M * p.E + e) * p.plane + g0 + ss;
        const int64_t kstep = (int64_t)p.E * p.plane;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int k = k0 + i * (NT / F);
            const cf x = lds[tid + i * NT];
            base[(int64_t)(k ^ (M / 2)) * kstep] = make_cf(x.x * p.scale, x.y * p.scale);
        }
    }
    }
}

// ---- contrib.istft fed series-major: the synthesis filterbank behind coherent_dedispersion -------------------------
// coherent_dedispersion -> istft is the way back from a channelised block (pulsarbat/contrib/misc.py:58-93 after
// transforms/dedispersion.py:125).  The dedispersion's last column pass can leave its cropped result series-major
// (series' q' = (c*M + shifted k)*E + e at q'*plane, time = segment index); this kernel reads THAT -- the mirror of
// k_stft_planar -- so the channelised result is neither re-interleaved into the reference layout nor read back from it.
// A tile is M (bins) x G (consecutive segments) x SB (series of the output): lanes run along the segment index on the
// load side (G >= 16: runs of >= 128 bytes), the values take one trip through LDS (rows padded to G + 1 slots: the column
// reads are then conflict-free) into the interleaved-column order of the tile FFT, and the inverse transform's outputs
// are stored as rows of SB series.  x M (the reference's x *= nperseg) and ifft's 1/M cancel: the transform is unscaled.
struct IstftPlanarParams {
    const cf* in;      // planar: series' q' = (c*M + (k ^ M/2))*E + e at q'*plane + segment
    cf* out;           // (nseg*M, S) sample-major
    const cf* tw16k;
    int64_t plane;
    int64_t nseg;      // segments that exist from this launch's first one on (the last tile may be short)
    int S, E;          // series of the output (nchan_out*E), inner elements per channel
    int SB, G;         // series per tile, segments per tile (SB*G = tile/M)
    int nsub = 1;      // sibling tiles a workgroup does one after the other (k_stft_planar): their partial lines of the output
                       // meet in one L2 within microseconds and leave it as whole lines
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_istft_planar(IstftPlanarParams p) {
    constexpr int F = kTilePoints / M;
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;
    constexpr int NT = kTilePoints / R;      // threads
    static_assert(F <= NT && F >= 16, "k_istft_planar: segment length out of range for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const int j = f % p.SB, sl = f / p.SB;
    const int64_t g0 = (int64_t)blockIdx.y * p.G;
    const int GP = p.G + 1;                  // padded row of the staging image

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);
    for (int sub = 0; sub < p.nsub; ++sub) {
    launder_all(w, std::make_integer_sequence<int, tw_seeds_or1(M, R)>{});
    const int q0 = (blockIdx.x * p.nsub + sub) * p.SB;
    if (sub) __syncthreads();   // the previous sibling's transform is done with the tile
    {
        // thread tid, round i: bin k = k0 + i*NT/F of series jj, segment ss (ss fastest across lanes)
        const int ss = tid % p.G, rest = tid / p.G;
        const int jj = rest % p.SB, k0 = rest / p.SB;
        const int q = q0 + jj;
        const int c = q / p.E, e = q - c * p.E;
        const bool have = g0 + ss < p.nseg;
        const cf* base = p.in + ((int64_t)c * M * p.E + e) * p.plane + g0 + ss;
        const int64_t kstep = (int64_t)p.E * p.plane;
        cf x[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int k = k0 + i * (NT / F);
            x[i] = have ? base[(int64_t)(k ^ (M / 2)) * kstep] : make_cf(0, 0);   // ifftshift
        }
#pragma unroll
        for (int i = 0; i < R; ++i) lds[((k0 + i * (NT / F)) * p.SB + jj) * GP + ss] = x[i];
    }
    __syncthreads();
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = lds[((tau + i * MR) * p.SB + j) * GP + sl];
    __syncthreads();
    fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
    // rows m = tau + i*MR of segment g0 + sl, series q0 + j: SB series of a time sample are adjacent.  Segments beyond the
    // end are dropped by an out-of-range offset (a branch around the stores invites the compiler to sink the transform into it)
    const rsrc_t ro = make_rsrc(p.out + g0 * M * (int64_t)p.S, (uint32_t)(((int64_t)p.G * M * p.S) * (int64_t)sizeof(cf)));
    const int voff = (g0 + sl < p.nseg) ? (int)((((int64_t)sl * M + tau) * p.S + q0 + j) * (int64_t)sizeof(cf)) : (int)0x80000000;
    const int step = MR * p.S * (int)sizeof(cf);
#pragma unroll
    for (int i = 0; i < R; ++i) buf_store(ro, voff, i * step, v[i]);
    }
}

// ---- contrib.stft / istft with one segment per tile (nperseg = 2^tile) and an even number of inner elements ----
// k_small would give each of the two polarisations of a channel to a different workgroup: 8-byte pieces at a
// 16-byte stride on both sides (2.0 TB/s).  Here a workgroup transforms BOTH series of a pair: it loads 16 or 32
// bytes per row (the pair is adjacent on the time-ordered side, (t*S + q), and on the channelised side,
// ((c*M + r)*E + e)), runs the two tile FFTs one after the other through the same LDS, and stores the pair.
struct SegPairParams {
    const cf* in;
    cf* out;
    const cf* tw16k;
    int S, E;        // series per time sample, inner elements per channel (even)
    int inverse;     // 0 stft (x 1/M, fftshift on the way out), 1 istft (ifftshift on the way in)
    real scale;
};

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_seg_pair(SegPairParams p) {
    static_assert(M == kTilePoints, "k_seg_pair: one segment of one series per tile");
    constexpr int MR = M / R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int tau = threadIdx.x;
    const int q = 2 * blockIdx.x;            // first series of the pair
    const int c = q / p.E, e = q - c * p.E;
    // segments are consecutive (M, S) blocks on both sides: one buffer descriptor per side and segment (the host checks that a
    // segment stays below 2 GiB), a per-lane byte offset and scalar row steps -- per-row 64-bit addresses cost this kernel
    // 252 B/lane of scratch (348 in the float64 build)
    const int64_t seg = (int64_t)blockIdx.y * M * p.S;
    const uint32_t span = (uint32_t)((int64_t)M * p.S * (int64_t)sizeof(cf));
    const rsrc_t ri = make_rsrc(p.in + seg, span), ro = make_rsrc(p.out + seg, span);
    // byte offsets of row (tau + i*MR): time-ordered (t*S + q); channelised ((c*M + r)*E + e)
    const int t_v = (tau * p.S + q) * (int)sizeof(cf), t_s = MR * p.S * (int)sizeof(cf);
    const int c_v = ((c * M + tau) * p.E + e) * (int)sizeof(cf), c_s = MR * p.E * (int)sizeof(cf);
    auto load2 = [&](int voff, int soff, cf& x, cf& y) {
        if constexpr (sizeof(cf) == 8) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(ri, voff, soff, 0);
            union { u32x4 u; real r[4]; } t;
            t.u = u;
            x = make_cf(t.r[0], t.r[1]);
            y = make_cf(t.r[2], t.r[3]);
        } else {
            x = buf_load(ri, voff, soff);
            y = buf_load(ri, voff + (int)sizeof(cf), soff);
        }
    };
    auto store2 = [&](int voff, int soff, cf x, cf y) {
        if constexpr (sizeof(cf) == 8) {
            union { u32x4 u; real r[4]; } t;
            t.r[0] = x.x; t.r[1] = x.y; t.r[2] = y.x; t.r[3] = y.y;
            __builtin_amdgcn_raw_buffer_store_b128(t.u, ro, voff, soff, 0);
        } else {
            buf_store(ro, voff, soff, x);
            buf_store(ro, voff + (int)sizeof(cf), soff, y);
        }
    };

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);
    cf a[R], b[R];
    if (p.inverse) {
#pragma unroll
        for (int i = 0; i < R; ++i) load2(c_v, (i ^ (R / 2)) * c_s, a[i], b[i]);   // ifftshift
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i) load2(t_v, i * t_s, a[i], b[i]);
    }
    // (the seeds are laundered before each transform: otherwise the twiddle-power trees of the first are kept
    //  alive for the second, ~140 VGPRs on top of the 128 the pair occupies)
    constexpr auto seeds = std::make_integer_sequence<int, tw_seeds_or1(M, R)>{};
    if (p.inverse) {
        launder_all(w, seeds);
        fft_tile<M, 1, R, +1, 1, true>(a, lds, tau, 0, w);
        __syncthreads();
        launder_all(w, seeds);
        fft_tile<M, 1, R, +1, 1, true>(b, lds, tau, 0, w);
    } else {
        launder_all(w, seeds);
        fft_tile<M, 1, R, -1, 1, true>(a, lds, tau, 0, w);
        __syncthreads();
        launder_all(w, seeds);
        fft_tile<M, 1, R, -1, 1, true>(b, lds, tau, 0, w);
    }
    if (p.inverse) {
#pragma unroll
        for (int i = 0; i < R; ++i)
            store2(t_v, i * t_s, make_cf(a[i].x * p.scale, a[i].y * p.scale), make_cf(b[i].x * p.scale, b[i].y * p.scale));
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i)   // fftshift
            store2(c_v, (i ^ (R / 2)) * c_s, make_cf(a[i].x * p.scale, a[i].y * p.scale), make_cf(b[i].x * p.scale, b[i].y * p.scale));
    }
}

}  // namespace PBH_NS
