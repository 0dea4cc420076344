// fft_core.hpp -- register-resident tile FFT for gfx950 (wave64, 160 KiB LDS / CU).
//
// A workgroup of 1024 threads owns a TILE of 2^14 complex64 points = F independent FFTs of
// length M (F * M = 2^14).  Every thread keeps R = 16 points in VGPRs; a Stockham radix-16
// stage is one in-register 16-point DFT per thread, and LDS is only the exchange medium
// between stages (write scattered, read unit-stride).  The Stockham index algebra makes the
// thread <-> point distribution identical before the first and after the last stage
// (thread tau holds positions tau + i*M/16, i = 0..15, in natural order), so a forward
// transform, a pointwise multiply and an inverse transform chain with no extra shuffles.
//
// No reference counterpart: the reference delegates to scipy.fft (pocketfft) via
// pulsarbat/fft.py:36-38; this is the from-scratch replacement of that call for c64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbh {

typedef float2 cf;

constexpr int kTileLog2 = 14;
constexpr int kTilePoints = 1 << kTileLog2;  // complex points per workgroup tile
constexpr int kR = 16;                       // points per thread
constexpr int kThreads = kTilePoints / kR;   // 1024
constexpr int kTwTable = 1 << 14;            // stage twiddle table: W_16384^p (forward sign)

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cf csqr(cf a) { return make_float2(a.x * a.x - a.y * a.y, 2.0f * a.x * a.y); }
__device__ __forceinline__ cf cconj(cf a) { return make_float2(a.x, -a.y); }
// multiply by -i (DIR = -1, forward) or +i (DIR = +1, inverse)
template <int DIR>
__device__ __forceinline__ cf mul_i(cf a) {
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// Multiply by W_16^p (forward: exp(-2 pi i p/16); inverse: conjugate).  p is a compile-time
// constant after unrolling, so the branches fold away.
template <int DIR>
__device__ __forceinline__ cf mul_w16(cf a, int p) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    p &= 15;
    if (p == 0) return a;
    if (p == 4) return mul_i<DIR>(a);
    if (p == 8) return make_float2(-a.x, -a.y);
    if (p == 12) return mul_i<-DIR>(a);
    float c, s;  // cos / sin of 2 pi p / 16
    switch (p) {
        case 1: c = c1; s = s1; break;
        case 2: c = h; s = h; break;
        case 3: c = s1; s = c1; break;
        case 5: c = -s1; s = c1; break;
        case 6: c = -h; s = h; break;
        case 7: c = -c1; s = s1; break;
        case 9: c = -c1; s = -s1; break;
        case 10: c = -h; s = -h; break;
        case 11: c = -s1; s = -c1; break;
        case 13: c = s1; s = -c1; break;
        case 14: c = h; s = -h; break;
        default: c = c1; s = -s1; break;  // 15
    }
    return cmul(a, make_float2(c, DIR < 0 ? -s : s));
}

// ---- in-register DFTs of size 1, 2, 4, 8, 16 (natural-order output) -------------------------
template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<1, DIR> {
    static __device__ __forceinline__ void run(cf (&)[1]) {}
};
template <int DIR>
struct Dft<2, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[2]) {
        cf a = v[0];
        v[0] = cadd(a, v[1]);
        v[1] = csub(a, v[1]);
    }
};
template <int DIR>
struct Dft<4, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[4]) {
        cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        cf t2 = cadd(v[1], v[3]), t3 = mul_i<DIR>(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3);
        v[3] = csub(t1, t3);
    }
};
// R = R1 * R2 with n = R2*n1 + n2, k = k1 + R1*k2:
//   A[k1][n2] = W_R^{n2 k1} * sum_n1 x[R2 n1 + n2] W_R1^{n1 k1};  X[k1 + R1 k2] = sum_n2 A[k1][n2] W_R2^{n2 k2}
template <int R1, int R2, int DIR>
__device__ __forceinline__ void dft_composite(cf (&v)[R1 * R2]) {
    constexpr int R = R1 * R2;
    cf a[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        cf t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
        Dft<R1, DIR>::run(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) a[n2][k1] = mul_w16<DIR>(t[k1], n2 * k1 * (16 / R));
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        cf t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[n2][k1];
        Dft<R2, DIR>::run(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
    }
}
template <int DIR>
struct Dft<8, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[8]) { dft_composite<2, 4, DIR>(v); }
};
template <int DIR>
struct Dft<16, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[16]) { dft_composite<4, 4, DIR>(v); }
};

// ---- stage plan ------------------------------------------------------------------------------
// Stages for length M: radix 16 while at least 16 remain, then one stage of the remainder.
constexpr int stage_radix(int M, int NS) { return (M / NS >= 16) ? 16 : (M / NS); }
// number of stage-twiddle seeds a thread needs for the stages starting at sub-length NS
constexpr int tw_seeds(int M, int NS) {
    return NS >= M ? 0
                   : ((NS > 1 ? 16 / stage_radix(M, NS) : 0) + tw_seeds(M, NS * stage_radix(M, NS)));
}
constexpr int tw_seeds_or1(int M) { return tw_seeds(M, 1) > 0 ? tw_seeds(M, 1) : 1; }

// LDS addressing of a tile: logical slot L = pos * PS + fofs, optionally padded by one slot
// every 16 to break power-of-two strides (ds_write_b64 lane groups are 16 lanes wide).
template <bool PAD>
__device__ __forceinline__ int lds_slot(int L) {
    return PAD ? L + (L >> 4) : L;
}
template <bool PAD>
constexpr int lds_bytes(int points) {
    return (PAD ? points + (points >> 4) : points) * (int)sizeof(cf);
}

// Load the per-stage twiddle seeds W_{NS*RAD}^{k} for this thread (global table, L2-resident).
template <int M, int NS>
__device__ __forceinline__ void load_tw_seeds(cf* w, int tau, const cf* __restrict__ tw) {
    if constexpr (NS < M) {
        constexpr int RAD = stage_radix(M, NS);
        constexpr int NB = 16 / RAD;
        if constexpr (NS > 1) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                int jb = tau + q * (M / 16);
                int k = jb & (NS - 1);
                w[q] = tw[k * (kTwTable / (NS * RAD))];
            }
            load_tw_seeds<M, NS * RAD>(w + NB, tau, tw);
        } else {
            load_tw_seeds<M, NS * RAD>(w, tau, tw);
        }
    }
}

// t[j] *= w^j, j = 1..RAD-1, powers by a depth<=4 product tree (w2=w^2, w3=w2*w, w4=w2^2, ...)
template <int RAD>
__device__ __forceinline__ void apply_powers(cf (&t)[RAD], cf w1) {
    if constexpr (RAD >= 2) t[1] = cmul(t[1], w1);
    if constexpr (RAD >= 4) {
        cf w2 = csqr(w1);
        cf w3 = cmul(w2, w1);
        t[2] = cmul(t[2], w2);
        t[3] = cmul(t[3], w3);
        if constexpr (RAD >= 8) {
            cf w4 = csqr(w2);
            cf w5 = cmul(w4, w1), w6 = csqr(w3), w7 = cmul(w4, w3);
            t[4] = cmul(t[4], w4);
            t[5] = cmul(t[5], w5);
            t[6] = cmul(t[6], w6);
            t[7] = cmul(t[7], w7);
            if constexpr (RAD >= 16) {
                cf w8 = csqr(w4);
                t[8] = cmul(t[8], w8);
                t[9] = cmul(t[9], cmul(w8, w1));
                t[10] = cmul(t[10], csqr(w5));
                t[11] = cmul(t[11], cmul(w8, w3));
                t[12] = cmul(t[12], csqr(w6));
                t[13] = cmul(t[13], cmul(w8, w5));
                t[14] = cmul(t[14], csqr(w7));
                t[15] = cmul(t[15], cmul(w8, w7));
            }
        }
    }
}

// One tile FFT of length M on the 16 points of each thread.
//   v[i]  : position tau + i*M/16 of this thread's FFT, natural order, in and out
//   lds   : tile exchange buffer; slot(pos) = lds_slot<PAD>(pos * PS + fofs)
//   w     : seeds from load_tw_seeds<M, 1> (forward sign; conjugated here for DIR = +1)
// Must be called by all threads of the workgroup (contains barriers).
template <int M, int NS, int DIR, int PS, bool PAD>
__device__ __forceinline__ void fft_tile(cf (&v)[16], cf* lds, int tau, int fofs, const cf* w) {
    if constexpr (NS < M) {
        constexpr int RAD = stage_radix(M, NS);
        constexpr int NB = 16 / RAD;
        constexpr bool LAST = (NS * RAD == M);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            cf t[RAD];
#pragma unroll
            for (int j = 0; j < RAD; ++j) t[j] = v[q + j * NB];
            int jb = tau + q * (M / 16);
            int k = jb & (NS - 1);
            if constexpr (NS > 1) {
                cf w1 = w[q];
                if (DIR > 0) w1 = cconj(w1);
                apply_powers<RAD>(t, w1);
            }
            Dft<RAD, DIR>::run(t);
            if constexpr (LAST) {
#pragma unroll
                for (int u = 0; u < RAD; ++u) v[q + u * NB] = t[u];
            } else {
                int base = (jb - k) * RAD + k;
#pragma unroll
                for (int u = 0; u < RAD; ++u) lds[lds_slot<PAD>((base + u * NS) * PS + fofs)] = t[u];
            }
        }
        if constexpr (!LAST) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = lds[lds_slot<PAD>((tau + i * (M / 16)) * PS + fofs)];
            __syncthreads();
            fft_tile<M, NS * RAD, DIR, PS, PAD>(v, lds, tau, fofs, w + (NS > 1 ? NB : 0));
        }
    }
}

// ---- float64 inter-pass twiddles W_N^p via a two-level table ------------------------------------
struct BigTwiddle {
    const double2* hi;  // W_N^{m << shift}
    const double2* lo;  // W_N^{l}, l < 2^shift
    int shift;
    int64_t mask;  // N - 1
};
__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 big_tw(const BigTwiddle& t, int64_t p) {
    p &= t.mask;
    double2 a = t.hi[p >> t.shift];
    double2 b = t.lo[p & ((1LL << t.shift) - 1)];
    return zmul(a, b);
}

}  // namespace pbh
