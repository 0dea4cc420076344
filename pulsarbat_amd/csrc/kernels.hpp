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
enum Layout { LAYOUT_INTERLEAVED = 0, LAYOUT_PLANAR = 1 };

// Column addressing.  A "column" is one (n2, series) pair; element (row, n2, s):
//   interleaved: row * (N2*S) + n2 * S + s          (the reference's (nsample, nchan, npol) block)
//   planar     : s * plane + row * N2 + n2          (one contiguous series per plane)
struct ColSide {
    int layout;
    int64_t plane;  // planar: elements between series
};

struct ColParams {
    const cf* in;
    cf* out;
    ColSide is, os;
    int enum_layout;  // which side's column order the tiles enumerate (that side is contiguous)
    int S;            // series = nchan * npol
    int N2;           // columns per series
    int64_t ncols;    // S * N2
    BigTwiddle tw;    // W_N, N = M * N2
    const cf* tw16k;  // stage twiddles
    int64_t crop_start, crop_stop;  // OP_TW_INV: keep time index t in [start, stop), t = row*N2 + n2
    int64_t out_shift;              // subtracted from the output offset (crop_start * S when interleaved)
};

__device__ __forceinline__ int64_t col_addr(const ColSide& sd, int64_t row, int n2, int s, int S, int N2) {
    return sd.layout == LAYOUT_INTERLEAVED ? (row * N2 + n2) * (int64_t)S + s
                                           : (int64_t)s * sd.plane + row * N2 + n2;
}

template <int M, int OP, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_col(ColParams p) {
    constexpr int F = kTilePoints / M;  // columns per tile
    constexpr bool PAD = F < 16;
    constexpr int MR = M / R;           // row stride between a thread's points
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int f = tid % F, tau = tid / F;
    const int64_t q = (int64_t)blockIdx.x * F + f;
    const bool valid = q < p.ncols;
    int n2 = 0, s = 0;
    if (valid) {
        if (p.enum_layout == LAYOUT_INTERLEAVED) {
            n2 = (int)(q / p.S);
            s = (int)(q % p.S);
        } else {
            s = (int)(q / p.N2);
            n2 = (int)(q % p.N2);
        }
    }

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int row = tau + i * MR;
        v[i] = valid ? p.in[col_addr(p.is, row, n2, s, p.S, p.N2)] : make_float2(0.f, 0.f);
    }

    // inter-pass twiddle W_N^{n2 * k1}, k1 = tau + i*M/R: base * step^i, float64 recurrence
    double2 zb = big_tw(p.tw, (int64_t)n2 * tau);
    double2 zs = big_tw(p.tw, (int64_t)n2 * MR);

    if constexpr (OP == OP_TW_INV) {
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            v[i] = cmul(v[i], make_float2((float)z.x, (float)-z.y));
            z = zmul(z, zs);
        }
        fft_tile<M, 1, R, +1, F, PAD>(v, lds, tau, f, w);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int row = tau + i * MR;
            int64_t t = (int64_t)row * p.N2 + n2;
            if (valid && t >= p.crop_start && t < p.crop_stop)
                p.out[col_addr(p.os, row, n2, s, p.S, p.N2) - p.out_shift] = v[i];
        }
    } else {
        fft_tile<M, 1, R, -1, F, PAD>(v, lds, tau, f, w);
        double2 z = zb;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int row = tau + i * MR;
            cf r = cmul(v[i], make_float2((float)z.x, (float)z.y));
            z = zmul(z, zs);
            if (valid) p.out[col_addr(p.os, row, n2, s, p.S, p.N2)] = r;
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

template <int M, int R>
__global__ __launch_bounds__(kTilePoints / R) void k_row(RowParams p) {
    constexpr int FR = kTilePoints / M;  // rows per tile
    constexpr int MR = M / R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);

    const int tid = threadIdx.x;
    const int tau = tid % MR, f = tid / MR;
    // tile = FR consecutive rows (same series: FR divides N1); rows past nrows are masked by
    // the descriptor's byte count (loads return 0, stores are dropped)
    const int64_t r0 = (int64_t)blockIdx.x * FR;
    const int64_t left = p.nrows - r0;
    const uint32_t bytes = (uint32_t)((left < FR ? left : FR) * (int64_t)M * sizeof(cf));
    const int64_t srs = r0 / p.N1;
    const int k1 = (int)(r0 % p.N1);
    const rsrc_t rd = make_rsrc(p.data + r0 * M, bytes);
    const rsrc_t rc = make_rsrc(p.chirp + ((srs / p.npol) * p.N1 + k1) * (int64_t)M, bytes);
    const int voff = (f * M + tau) * (int)sizeof(cf);
    constexpr int STEP = MR * (int)sizeof(cf);

    cf w[tw_seeds_or1(M, R)];
    load_tw_seeds<M, 1, R>(w, tau, p.tw16k);

    cf v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = buf_load(rd, voff, i * STEP);

    fft_tile<M, 1, R, -1, 1, true>(v, lds, tau, f * M, w);
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = cmul(v[i], buf_load(rc, voff, i * STEP));
    fft_tile<M, 1, R, +1, 1, true>(v, lds, tau, f * M, w);

#pragma unroll
    for (int i = 0; i < R; ++i) buf_store(rd, voff, i * STEP, v[i]);
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
