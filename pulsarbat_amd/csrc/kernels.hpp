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

namespace pbh {

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
__global__ __launch_bounds__(kTilePoints / R, XS ? 4 : 1) void k_col(ColParams p) {
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
        for (int i = 0; i < R; ++i) v[i] = valid ? src[i * step] : make_float2(0.f, 0.f);
    }
    cf* dst = p.out + col_addr(p.os, tau, n2, s, p.S, p.N2) - p.out_shift;
    const int64_t ostep = (int64_t)MR * p.os.row_stride;

    if constexpr (OP == OP_TW_INV) {
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = cmul(v[i], make_float2((float)z.x, (float)-z.y));
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
            cf r = cmul(v[i], make_float2((float)z.x, (float)z.y));
            z = zmul(z, zs);
            if (valid) dst[i * ostep] = r;
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
};

// Persistent, software-pipelined: a workgroup walks over tiles; while tile i is transformed the
// memory system works for it -- the chirp row is requested BEFORE the forward FFT (consumed after
// it) and tile i+1's samples are requested BEFORE the inverse FFT (consumed next iteration).  Both
// sets live in VGPRs (64 each) beside the 64 data registers; buffer loads survive the barriers.
// Tile order: the pols of one (channel, k1) run back to back on the same workgroup so the second
// one finds the chirp row in L2/MALL instead of HBM.
template <int M, int R, bool PF>
__global__ __launch_bounds__(kTilePoints / R) void k_row(RowParams p) {
    constexpr int FR = kTilePoints / M;  // rows per tile
    constexpr int MR = M / R;
    constexpr int STEP = MR * (int)sizeof(cf);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int tau = tid % MR, f = tid / MR;
    const int voff = (f * M + tau) * (int)sizeof(cf);
    const int64_t ntile = (p.nrows + FR - 1) / FR;

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
        return t * FR;
    };
    auto data_rsrc = [&](int64_t r0) {
        const int64_t left = p.nrows - r0;
        const uint32_t bytes = (uint32_t)((left < FR ? left : FR) * (int64_t)M * sizeof(cf));
        return make_rsrc(p.data + r0 * M, bytes);
    };

    int64_t t = blockIdx.x;
    if (t >= ntile) return;
    int64_t r0 = first_row(t);
    rsrc_t rd = data_rsrc(r0);
    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff, i * STEP);

    while (true) {
        // keep the twiddle-power trees inside the iteration: hoisted out of the loop they would
        // pin ~140 VGPRs (LICM), which is what the prefetch registers need
#pragma unroll
        for (int i = 0; i < tw_seeds_or1(M, R); ++i) asm volatile("" : "+v"(w[i].x), "+v"(w[i].y));
        const int64_t srs = r0 / p.N1;
        const int k1 = (int)(r0 - srs * p.N1);
        const rsrc_t rc = make_rsrc(p.chirp + ((srs / p.npol) * p.N1 + k1) * (int64_t)M,
                                    (uint32_t)(FR * (int64_t)M * sizeof(cf)));
        if constexpr (PF) {
            cf c[R];
#pragma unroll
            for (int i = 0; i < R; ++i) c[i] = buf_load(rc, voff, i * STEP);
            fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], c[i]);
        } else {
            fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], buf_load(rc, voff, i * STEP));
        }

        const int64_t tn = t + gridDim.x;
        const bool more = tn < ntile;
        const int64_t rn = more ? first_row(tn) : r0;
        const rsrc_t rdn = more ? data_rsrc(rn) : make_rsrc(p.data, 0);
        if constexpr (PF) {
            cf nx[R];
#pragma unroll
            for (int i = 0; i < R; ++i) nx[i] = buf_load(rdn, voff, i * STEP);
            fft_tile<M, 1, R, +1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
            for (int i = 0; i < R; ++i) buf_store(rd, voff, i * STEP, v[i]);
            if (!more) break;
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = nx[i];
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

// ---- single-tile transform (nsample = M <= 2^14) -------------------------------------------------------
struct SmallParams {
    const cf* in;     // (M, S) interleaved
    cf* out;          // (stop-start, S) interleaved
    const cf* chirp;  // (nchan, M), pre-scaled by 1/M; nullptr: plain FFT (no product)
    const cf* tw16k;
    int S, npol;
    int64_t crop_start, crop_stop;
    int dir;      // plain-FFT mode only: -1 forward, +1 inverse
    float scale;  // plain-FFT mode only
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
    const int64_t q = (int64_t)blockIdx.x * F + f;
    const bool valid = q < p.S;
    const int64_t qq = valid ? q : 0;

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int64_t row = tau + i * MR;
        v[i] = valid ? p.in[row * p.S + qq] : make_float2(0.f, 0.f);
    }
    if (p.chirp) {
        const cf* crow = p.chirp + (qq / p.npol) * (int64_t)M;
        fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = cmul(v[i], crow[tau + i * MR]);
        fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
    } else if (p.dir < 0) {
        fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
    } else {
        fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = make_float2(v[i].x * p.scale, v[i].y * p.scale);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int64_t row = tau + i * MR;
        if (valid && row >= p.crop_start && row < p.crop_stop)
            p.out[(row - p.crop_start) * p.S + qq] = v[i];
    }
}

}  // namespace pbh
