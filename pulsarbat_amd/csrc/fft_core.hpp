// fft_core.hpp -- register-resident tile FFT for gfx950 (wave64, 160 KiB LDS / CU).
//
// A workgroup owns a TILE of 2^14 complex64 points = F independent FFTs of length M
// (F * M = 2^14).  Every thread keeps R points (R = 16 or 32) in VGPRs; a Stockham stage is
// one in-register radix-R DFT per thread, and LDS is only the exchange medium between stages
// (write scattered, read unit-stride).  The Stockham index algebra makes the thread <-> point
// distribution identical before the first and after the last stage (thread tau holds
// positions tau + i*M/R, i = 0..R-1, in natural order), so a forward transform, a pointwise
// multiply and an inverse transform chain with no extra shuffles.
//
// R = 32 with 512 threads gives each thread a 256-VGPR budget (no spills) and turns a 2^14
// transform into 32 x 32 x 16: two LDS exchanges instead of three.
//
// No reference counterpart: the reference delegates to scipy.fft (pocketfft) via
// pulsarbat/fft.py:36-38; this is the from-scratch replacement of that call for c64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "pbh_config.hpp"

namespace PBH_NS {

typedef PBH_REAL real;
typedef PBH_REAL2 cf;
__host__ __device__ __forceinline__ cf make_cf(real x, real y) { return PBH_MAKE2(x, y); }

constexpr int kTileLog2 = PBH_TILE_LOG2;
constexpr int kTilePoints = 1 << kTileLog2;  // complex points per workgroup tile
constexpr int kTwTable = 1 << 14;            // stage twiddle table: W_16384^p (forward sign)
constexpr int kMixMaxStages = 14;            // stages of a mixed-radix transform (mixed_kernels.hpp; a member of the plan structure)

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_cf(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_cf(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    return make_cf(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cf csqr(cf a) { return make_cf(a.x * a.x - a.y * a.y, RC(2.0) * a.x * a.y); }
__device__ __forceinline__ cf cconj(cf a) { return make_cf(a.x, -a.y); }
// multiply by -i (DIR = -1, forward) or +i (DIR = +1, inverse)
template <int DIR>
__device__ __forceinline__ cf mul_i(cf a) {
    return DIR < 0 ? make_cf(a.y, -a.x) : make_cf(-a.y, a.x);
}

// Multiply by W_32^p (forward: exp(-2 pi i p/32); inverse: conjugate).  p is a compile-time
// constant after unrolling, so the branches fold away.
template <int DIR>
__device__ __forceinline__ cf mul_w32(cf a, int p) {
    p &= 31;
    const int quad = p >> 3, r = p & 7;
    if (r != 0) {
        real c, s;
        switch (r) {
            case 1: c = RC(0.98078528040323043); s = RC(0.19509032201612825); break;
            case 2: c = RC(0.92387953251128674); s = RC(0.38268343236508977); break;
            case 3: c = RC(0.83146961230254524); s = RC(0.55557023301960218); break;
            case 4: c = RC(0.70710678118654752); s = RC(0.70710678118654752); break;
            case 5: c = RC(0.55557023301960218); s = RC(0.83146961230254524); break;
            case 6: c = RC(0.38268343236508977); s = RC(0.92387953251128674); break;
            default: c = RC(0.19509032201612825); s = RC(0.98078528040323043); break;
        }
        a = cmul(a, make_cf(c, DIR < 0 ? -s : s));
    }
    if (quad == 1) return mul_i<DIR>(a);
    if (quad == 2) return make_cf(-a.x, -a.y);
    if (quad == 3) return mul_i<-DIR>(a);
    return a;
}

// ---- in-register DFTs of size 1..32 (natural-order output) --------------------------------------
template <int R, int DIR>
struct Dft;

template <int DIR>
struct Dft<1, DIR> {
    static __device__ __forceinline__ void run(cf (&)[1]) {}
    template <class TK>
    static __device__ __forceinline__ void run(cf (&)[1], TK tick) { tick(); }
};
template <int DIR>
struct Dft<2, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[2]) {
        cf a = v[0];
        v[0] = cadd(a, v[1]);
        v[1] = csub(a, v[1]);
    }
    template <class TK>
    static __device__ __forceinline__ void run(cf (&v)[2], TK tick) { run(v); tick(); }
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
    template <class TK>
    static __device__ __forceinline__ void run(cf (&v)[4], TK tick) { run(v); tick(); }
};
// R = R1 * R2 with n = R2*n1 + n2, k = k1 + R1*k2:
//   A[k1][n2] = W_R^{n2 k1} * sum_n1 x[R2 n1 + n2] W_R1^{n1 k1};  X[k1 + R1 k2] = sum_n2 A[k1][n2] W_R2^{n2 k2}
struct NoTick {
    __device__ __forceinline__ void operator()() const {}
};
// tick(): called after every sub-transform of the OUTER composite level (R2 + R1 times for a radix-32
// butterfly); kernels hang their global loads / stores on it to spread them through the arithmetic.
template <int R1, int R2, int DIR, class TK = NoTick>
__device__ __forceinline__ void dft_composite(cf (&v)[R1 * R2], TK tick = TK{}) {
    constexpr int R = R1 * R2;
    cf a[R2][R1];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
        cf t[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[R2 * n1 + n2];
        Dft<R1, DIR>::run(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) a[n2][k1] = mul_w32<DIR>(t[k1], n2 * k1 * (32 / R));
        tick();
    }
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        cf t[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[n2][k1];
        Dft<R2, DIR>::run(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
        tick();
    }
}
template <int DIR>
struct Dft<8, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[8]) { dft_composite<2, 4, DIR>(v); }
    template <class TK>
    static __device__ __forceinline__ void run(cf (&v)[8], TK tick) { dft_composite<2, 4, DIR, TK>(v, tick); }
};
template <int DIR>
struct Dft<16, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[16]) { dft_composite<4, 4, DIR>(v); }
    template <class TK>
    static __device__ __forceinline__ void run(cf (&v)[16], TK tick) { dft_composite<4, 4, DIR, TK>(v, tick); }
};
template <int DIR>
struct Dft<32, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[32]) { dft_composite<4, 8, DIR>(v); }
    template <class TK>
    static __device__ __forceinline__ void run(cf (&v)[32], TK tick) { dft_composite<4, 8, DIR, TK>(v, tick); }
};

// ---- small odd DFTs (3, 5, 7 points): the odd factor of lengths m * 2^k, done across the chunks of the
// radix-P stage (aux_kernels.hpp: k_radix_p, k_deint_radix).  Direct O(P^2) form with literal roots of unity:
// P complex multiplies per element in passes that are bound by memory anyway.
template <int P>
struct OddRoots;
template <>
struct OddRoots<3> {
    static __device__ __forceinline__ real c(int j) { constexpr real t[3] = {RC(1.0), RC(-0.49999999999999978), RC(-0.50000000000000044)}; return t[j]; }
    static __device__ __forceinline__ real s(int j) { constexpr real t[3] = {RC(0.0), RC(0.86602540378443871), RC(-0.86602540378443837)}; return t[j]; }
};
template <>
struct OddRoots<5> {
    static __device__ __forceinline__ real c(int j) { constexpr real t[5] = {RC(1.0), RC(0.30901699437494745), RC(-0.80901699437494734), RC(-0.80901699437494756), RC(0.30901699437494723)}; return t[j]; }
    static __device__ __forceinline__ real s(int j) { constexpr real t[5] = {RC(0.0), RC(0.95105651629515353), RC(0.58778525229247325), RC(-0.58778525229247303), RC(-0.95105651629515364)}; return t[j]; }
};
template <>
struct OddRoots<7> {
    static __device__ __forceinline__ real c(int j) { constexpr real t[7] = {RC(1.0), RC(0.62348980185873359), RC(-0.22252093395631434), RC(-0.90096886790241903), RC(-0.90096886790241915), RC(-0.22252093395631459), RC(0.62348980185873337)}; return t[j]; }
    static __device__ __forceinline__ real s(int j) { constexpr real t[7] = {RC(0.0), RC(0.7818314824680298), RC(0.97492791218182362), RC(0.43388373911755823), RC(-0.43388373911755801), RC(-0.97492791218182362), RC(-0.78183148246802991)}; return t[j]; }
};
template <int P, int DIR>
__device__ __forceinline__ void dft_odd(cf (&v)[P]) {
    cf o[P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        cf acc = v[0];
#pragma unroll
        for (int a = 1; a < P; ++a) {
            const int j = (a * c) % P;   // compile-time after unrolling
            const real wr = OddRoots<P>::c(j), wi = DIR < 0 ? -OddRoots<P>::s(j) : OddRoots<P>::s(j);
            acc.x += v[a].x * wr - v[a].y * wi;
            acc.y += v[a].x * wi + v[a].y * wr;
        }
        o[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < P; ++c) v[c] = o[c];
}
template <int DIR>
struct Dft<3, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[3]) { dft_odd<3, DIR>(v); }
};
template <int DIR>
struct Dft<5, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[5]) { dft_odd<5, DIR>(v); }
};
template <int DIR>
struct Dft<7, DIR> {
    static __device__ __forceinline__ void run(cf (&v)[7]) { dft_odd<7, DIR>(v); }
};

// ---- stage plan ------------------------------------------------------------------------------
// Stages for length M with R points per thread: radix R while at least R remain, then one
// stage of the remainder.
constexpr int kMaxRadix = 32;  // largest in-register DFT; R = 64 points per thread means two radix-32 butterflies
// ORD = 0: radix R while at least R remain, then one stage of the remainder (32, 32, 16 for 2^14 points at R = 32).
// ORD = 1: the remainder FIRST (16, 32, 32): with PIN (below) the first stage's two radix-16 butterflies of a thread start from
//          ADJACENT points, so a row tile can be loaded with 16-byte accesses.
constexpr int small_factor(int M, int Rm) {
    int m = M;
    while (m >= Rm && m % Rm == 0) m /= Rm;
    return m;
}
constexpr int stage_radix(int M, int NS, int R, int ORD = 0) {
    const int Rm = R < kMaxRadix ? R : kMaxRadix;
    if (ORD == 1) return (NS == 1 && small_factor(M, Rm) > 1) ? small_factor(M, Rm) : Rm;
    return (M / NS >= Rm) ? Rm : (M / NS);
}
// number of stage-twiddle seeds a thread needs for the stages starting at sub-length NS
// (the last stage's butterflies q = 0..NB-1 share one seed: W_M^{(tau + q M/R) j} = (W_M^tau)^j W_R^{q j},
//  and W_R^{q j} is a compile-time constant; with POUT their bases are NB tau + q and each has its own seed)
constexpr int stage_seeds(int M, int NS, int R, int ORD = 0, bool POUT = false) {
    return NS <= 1 ? 0
                   : ((NS * stage_radix(M, NS, R, ORD) == M && R <= kMaxRadix && !(POUT && R / stage_radix(M, NS, R, ORD) > 1))
                          ? 1 : R / stage_radix(M, NS, R, ORD));
}
constexpr int tw_seeds(int M, int NS, int R, int ORD = 0, bool POUT = false) {
    return NS >= M ? 0 : (stage_seeds(M, NS, R, ORD, POUT) + tw_seeds(M, NS * stage_radix(M, NS, R, ORD), R, ORD, POUT));
}
constexpr int tw_seeds_or1(int M, int R, int ORD = 0, bool POUT = false) {
    return tw_seeds(M, 1, R, ORD, POUT) > 0 ? tw_seeds(M, 1, R, ORD, POUT) : 1;
}
// index of the stage that starts at sub-length NS, and the number of stages
constexpr int stage_index(int M, int NS, int R, int ORD = 0) {
    int s = 0;
    for (int ns = 1; ns < NS; ns *= stage_radix(M, ns, R, ORD)) ++s;
    return s;
}
constexpr int stage_count(int M, int R, int ORD = 0) { return stage_index(M, M, R, ORD); }
// sub-length at which the last stage starts
constexpr int last_stage_ns(int M, int R, int ORD = 0) {
    int ns = 1;
    while (ns * stage_radix(M, ns, R, ORD) < M) ns *= stage_radix(M, ns, R, ORD);
    return ns;
}

// LDS addressing of a tile.  Logical slot L = pos * PS + fofs (8-byte slots).
//   PAD = false: identity (column tiles with >= 16 interleaved FFTs are conflict-free as is)
//   PAD = true : one pad slot per 32, phys = L + (L >> 5).  A radix-R stage writes with lane
//                stride R slots; the pad turns that into stride R+1 (conflict-free across a
//                16-lane ds_write_b64 group), and unit-stride reads of 32 lanes stay 32
//                consecutive slots.  The map is linear (phys(L0 + d) = phys(L0) + d + (d >> 5)
//                when the low 5 bits do not carry), so every access is base VGPR + immediate.
template <bool PAD>
__device__ __forceinline__ int lds_phys(int L) {
    return PAD ? L + (L >> 5) : L;
}
template <bool PAD>
constexpr int lds_lin(int d) {
    return PAD ? d + (d >> 5) : d;
}
template <bool PAD>
constexpr int lds_tile_bytes() {
    return (PAD ? kTilePoints + (kTilePoints >> 5) : kTilePoints) * (int)sizeof(cf);
}

// Load the per-stage twiddle seeds W_{NS*RAD}^{k} for this thread (global table, L2-resident).
template <int M, int NS, int R, int ORD = 0, bool POUT = false>
__device__ __forceinline__ void load_tw_seeds(cf* w, int tau, const cf* __restrict__ tw) {
    if constexpr (NS < M) {
        constexpr int RAD = stage_radix(M, NS, R, ORD);
        constexpr int NB = R / RAD;
        if constexpr (NS > 1) {
            constexpr int NSEED = stage_seeds(M, NS, R, ORD, POUT);
            constexpr bool PAIRS = POUT && NS * RAD == M && NB > 1;   // last stage, bases NB tau + q
#pragma unroll
            for (int q = 0; q < NSEED; ++q) {
                int jb = PAIRS ? NB * tau + q : tau + q * (M / R);
                int k = jb & (NS - 1);
                w[q] = tw[k * (kTwTable / (NS * RAD))];
            }
            load_tw_seeds<M, NS * RAD, R, ORD, POUT>(w + NSEED, tau, tw);
        } else {
            load_tw_seeds<M, NS * RAD, R, ORD, POUT>(w, tau, tw);
        }
    }
}

// t[j] *= w^j, j = 1..RAD-1.  Powers come from a product tree of depth <= 3 for w^1..w^7 and of depth
// <= 5 for the block bases w^8, w^16, w^24; t[8b + j] gets (w^8b * w^j).  Only w^1..w^7 and one base are
// live at a time (18 registers instead of the 34 a full table of powers needs).
template <int RAD>
__device__ __forceinline__ void apply_powers(cf (&t)[RAD], cf w1) {
    if constexpr (RAD >= 2) t[1] = cmul(t[1], w1);
    if constexpr (RAD >= 4) {
        cf w2 = csqr(w1);
        cf w3 = cmul(w2, w1);
        t[2] = cmul(t[2], w2);
        t[3] = cmul(t[3], w3);
        if constexpr (RAD >= 8) {
            cf p[8];
            p[1] = w1; p[2] = w2; p[3] = w3;
            p[4] = csqr(w2);
            p[5] = cmul(p[4], w1);
            p[6] = csqr(w3);
            p[7] = cmul(p[4], w3);
#pragma unroll
            for (int j = 4; j < 8; ++j) t[j] = cmul(t[j], p[j]);
            if constexpr (RAD >= 16) {
                const cf w8 = csqr(p[4]);
                t[8] = cmul(t[8], w8);
#pragma unroll
                for (int j = 1; j < 8; ++j) t[8 + j] = cmul(t[8 + j], cmul(w8, p[j]));
                if constexpr (RAD >= 32) {
                    const cf w16 = csqr(w8);
                    t[16] = cmul(t[16], w16);
#pragma unroll
                    for (int j = 1; j < 8; ++j) t[16 + j] = cmul(t[16 + j], cmul(w16, p[j]));
                    const cf w24 = cmul(w16, w8);
                    t[24] = cmul(t[24], w24);
#pragma unroll
                    for (int j = 1; j < 8; ++j) t[24 + j] = cmul(t[24 + j], cmul(w24, p[j]));
                }
            }
        }
    }
}

// One tile FFT of length M on the R points of each thread.
//   v[i]  : position tau + i*M/R of this thread's FFT, natural order, in and out
//   lds   : tile exchange buffer; slot(pos) = lds_phys<PAD>(pos * PS + fofs)
//   w     : seeds from load_tw_seeds<M, 1, R> (forward sign; conjugated here for DIR = +1)
// fofs must be a multiple of 32 when PS == 1 (row tiles: fofs = f * M).
// Must be called by all threads of the workgroup (contains barriers).
// WSYNC = true: the "tile" is private to ONE wavefront (tau = lane); LDS operations of a wave execute
// in order, so the exchange needs only a compiler-level fence, no workgroup barrier.
template <bool WSYNC>
__device__ __forceinline__ void tile_sync() {
    if constexpr (WSYNC) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// HK: optional hook the caller uses to spread its global-memory instructions over the transform
// instead of issuing them in one burst (a burst blocks the in-order issue of every wave for
// thousands of cycles).  hk(stage, -1) runs before each stage's butterflies; hk(stage, q) runs in
// the LAST stage after butterfly q has put its outputs v[q + u*NB] in place (q is the unrolled loop
// variable, a plain int).  The stage is a std::integral_constant so the hook can index register arrays.
struct NoHook {
    template <class A, class B>
    __device__ __forceinline__ void operator()(A, B) const {}
};
struct tick_tag {};  // hk(stage, tick_tag{}) runs after every sub-transform of a stage's butterflies
template <int V>
using ic = std::integral_constant<int, V>;

// ORD: stage order (stage_radix).  PIN: the FIRST stage's NB butterflies of a thread have the adjacent bases NB tau + q
// (the caller loads v[q + j NB] = x[NB tau + q + j M/RAD]: NB consecutive points per access) instead of tau + q M/R.
// POUT: the same for the LAST stage: the thread ends up with v[q + u NB] = X[NB tau + q + u M/RAD] -- NB consecutive
// outputs per store; the exchange in front of that stage hands every thread the inputs of those butterflies.
template <int M, int NS, int R, int DIR, int PS, bool PAD, bool XS = false, bool WSYNC = false, class HK = NoHook,
          int ORD = 0, bool PIN = false, bool POUT = false>
__device__ __forceinline__ void fft_tile(cf (&v)[R], cf* lds, int tau, int fofs, const cf* w, HK hk = HK{}) {
    if constexpr (NS < M) {
        constexpr int RAD = stage_radix(M, NS, R, ORD);
        constexpr int NB = R / RAD;
        constexpr int MR = M / R;
        constexpr bool LAST = (NS * RAD == M);
        constexpr int SI = stage_index(M, NS, R, ORD);
        constexpr bool PAIRS = NB > 1 && ((PIN && NS == 1) || (POUT && LAST));   // bases NB tau + q in this stage
        // the exchange in front of a pair-adjacent LAST stage is not padded: its writes are lane-consecutive slots as they
        // stand (conflict-free), and without the pad slots a thread's two adjacent points are one ALIGNED 16-byte read
        // (ds_read_b128, 16 lanes = 256 contiguous bytes).  Padded, that read became a ds_read2_b64 with its 16-lane groups two
        // slots apart: 2-way bank conflicts on every access (1.68e7 SQ_LDS_BANK_CONFLICT cycles per launch of k_rowp16).
        constexpr int NRAD_N = LAST ? 1 : stage_radix(M, NS * RAD, R, ORD);
        constexpr bool NEXT_IS_PAIRS = !LAST && POUT && NS * RAD * NRAD_N == M && R / NRAD_N == 2 && sizeof(cf) == 8;
        constexpr bool WPAD = PAD && !NEXT_IS_PAIRS;
        static_assert(!(PIN || POUT) || (!XS && !WSYNC), "pair-adjacent stages: plain exchange only");
        hk(ic<SI>{}, ic<-1>{});
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            cf t[RAD];
#pragma unroll
            for (int j = 0; j < RAD; ++j) t[j] = v[q + j * NB];
            int jb = PAIRS ? NB * tau + q : tau + q * MR;
            int k = jb & (NS - 1);
            if constexpr (NS > 1) {
                constexpr bool ONE_SEED = LAST && NB > 1 && R <= kMaxRadix && !PAIRS;
                cf w1 = w[ONE_SEED ? 0 : q];
                if (DIR > 0) w1 = cconj(w1);
                apply_powers<RAD>(t, w1);
                if constexpr (ONE_SEED) {
#pragma unroll
                    for (int j = 1; j < RAD; ++j) t[j] = mul_w32<DIR>(t[j], q * j * (32 / R));
                }
            }
            if constexpr (std::is_same<HK, NoHook>::value) Dft<RAD, DIR>::run(t);
            else Dft<RAD, DIR>::run(t, [&]() { hk(ic<SI>{}, tick_tag{}); });
            if constexpr (LAST) {
#pragma unroll
                for (int u = 0; u < RAD; ++u) v[q + u * NB] = t[u];
                hk(ic<SI>{}, q);
            } else if constexpr (XS) {
                // split exchange (half the LDS): park the butterfly outputs back in v; the real and
                // imaginary parts go through a float buffer one after the other below
#pragma unroll
                for (int u = 0; u < RAD; ++u) v[q + u * NB] = t[u];
            } else {
                const int base = (jb - k) * RAD + k;
                // linear form is exact when the step is a multiple of 32 slots, or when it is the
                // first stage of a row tile (base*PS+fofs is a multiple of RAD, u < RAD <= 32)
                constexpr bool LIN = !WPAD || ((NS * PS) % 32 == 0) || (NS == 1 && PS == 1 && 32 % RAD == 0);
                if constexpr (LIN) {
                    cf* wp = lds + lds_phys<WPAD>(base * PS + fofs);
#pragma unroll
                    for (int u = 0; u < RAD; ++u) wp[lds_lin<WPAD>(u * NS * PS)] = t[u];
                } else {
#pragma unroll
                    for (int u = 0; u < RAD; ++u) lds[lds_phys<WPAD>((base + u * NS) * PS + fofs)] = t[u];
                }
            }
        }
        if constexpr (!LAST && XS) {
            static_assert(!PAD, "split exchange is implemented for unpadded (column) tiles");
            real* fl = reinterpret_cast<real*>(lds);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int jb = tau + q * MR;
                    const int k = jb & (NS - 1);
                    real* wp = fl + ((jb - k) * RAD + k) * PS + fofs;
#pragma unroll
                    for (int u = 0; u < RAD; ++u) wp[u * NS * PS] = part ? v[q + u * NB].y : v[q + u * NB].x;
                }
                __syncthreads();
                const real* rp = fl + tau * PS + fofs;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (part) v[i].y = rp[i * MR * PS];
                    else v[i].x = rp[i * MR * PS];
                }
                __syncthreads();
            }
            fft_tile<M, NS * RAD, R, DIR, PS, PAD, XS, false, HK, ORD, PIN, POUT>(v, lds, tau, fofs, w + stage_seeds(M, NS, R, ORD, POUT), hk);
        } else if constexpr (!LAST) {
            tile_sync<WSYNC>();
            constexpr int NRAD = stage_radix(M, NS * RAD, R, ORD), NNB = R / NRAD;
            constexpr bool NEXT_PAIRS = POUT && NS * RAD * NRAD == M && NNB > 1;   // the next stage is the last, pair-adjacent
            if constexpr (NEXT_PAIRS) {
                // v[q + j NNB] = position (NNB tau + q) + j M/NRAD: NNB consecutive slots per read (M/NRAD is a multiple of 32
                // slots for row tiles, so the padded map stays linear in j; the pair never straddles a pad slot)
                static_assert(PS == 1 && (M / NRAD) % 32 == 0 && 32 % NNB == 0, "pair-adjacent last stage: row tiles");
                if constexpr (NEXT_IS_PAIRS) {
                    static_assert(NS % 16 == 0 || NS == 1, "unpadded exchange: lane-consecutive writes");
                    const float4* rp4 = reinterpret_cast<const float4*>(lds + NNB * tau + fofs);   // fofs even, lds 16-byte aligned
#pragma unroll
                    for (int j = 0; j < NRAD; ++j) {
                        const float4 x = rp4[j * (M / NRAD) / 2];
                        v[2 * j] = make_cf(x.x, x.y);
                        v[2 * j + 1] = make_cf(x.z, x.w);
                    }
                } else {
                    const cf* rp = lds + lds_phys<PAD>(NNB * tau + fofs);
#pragma unroll
                    for (int j = 0; j < NRAD; ++j)
#pragma unroll
                        for (int q = 0; q < NNB; ++q) v[q + j * NNB] = rp[lds_lin<PAD>(j * (M / NRAD)) + q];
                }
            } else {
            constexpr bool RLIN = !PAD || ((MR * PS) % 32 == 0);
            if constexpr (RLIN) {
                const cf* rp = lds + lds_phys<PAD>(tau * PS + fofs);
#pragma unroll
                for (int i = 0; i < R; ++i) v[i] = rp[lds_lin<PAD>(i * MR * PS)];
            } else {
#pragma unroll
                for (int i = 0; i < R; ++i) v[i] = lds[lds_phys<PAD>((tau + i * MR) * PS + fofs)];
            }
            }
            tile_sync<WSYNC>();
            fft_tile<M, NS * RAD, R, DIR, PS, PAD, XS, WSYNC, HK, ORD, PIN, POUT>(v, lds, tau, fofs, w + stage_seeds(M, NS, R, ORD, POUT), hk);
        }
    }
}

// Make the compiler forget where a register value came from (stops LICM from hoisting the twiddle
// power trees out of a persistent loop).  A fold expression, NOT a loop: a loop over asm volatile
// is not unrolled, which would put the array in scratch memory and a vmcnt(0) behind every reload.
__device__ __forceinline__ void launder1(cf& a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }
template <int... I>
__device__ __forceinline__ void launder_all(cf* a, std::integer_sequence<int, I...>) {
    (launder1(a[I]), ...);
}

// cache-policy bits of the streaming buffer accesses (gfx940+: 1 = sc0, 2 = nt, 16 = sc1); experiments only
#ifndef PBH_LOAD_AUX
#define PBH_LOAD_AUX 0
#endif
#ifndef PBH_STORE_AUX
#define PBH_STORE_AUX 0
#endif

// ---- buffer (SRD) addressing: wave-uniform base in SGPRs, 32-bit per-lane offset, scalar step ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
#ifdef PBH_F64
__device__ __forceinline__ cf buf_load(rsrc_t r, int voff, int soff) {
    union { u32x4 u; cf c; } x;
    x.u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return x.c;
}
// gfx950 hazard LLVM does not cover (found with tools/stress_c128.py): a buffer store of more than
// 64 bits per lane WITH an SGPR offset, directly followed by a VALU write of its data registers
// (`buffer_store_dwordx4 v[92:95], ..., s63 offen` then `v_fma_f64 v[94:95], ...`), stores stale data
// in 4 of every 16 lanes.  The hazard recognizer only guards the no-soffset form.  Every complex128
// store therefore carries its own wait states, and nothing is scheduled across them.
template <int AUX = 0>
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, cf a) {
    union { u32x4 u; cf c; } x;
    x.c = a;
    __builtin_amdgcn_raw_buffer_store_b128(x.u, r, voff, soff, AUX);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 3");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
#else
__device__ __forceinline__ cf buf_load(rsrc_t r, int voff, int soff) {
    u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, PBH_LOAD_AUX);
    return make_cf(__uint_as_float(x.x), __uint_as_float(x.y));
}
template <int AUX = PBH_STORE_AUX>   // cache policy bits: 1 sc0, 2 nt, 16 sc1
__device__ __forceinline__ void buf_store(rsrc_t r, int voff, int soff, cf a) {
    u32x2 x;
    x.x = __float_as_uint(a.x);
    x.y = __float_as_uint(a.y);
    __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, AUX);
}
// Two complex64 values as ONE 16-byte store.  The same gfx950 hazard as the complex128 store above applies: met again in
// round 4 in k_rowq16, where hipcc computed the last butterfly's sums into the registers of the previous store
// (`buffer_store_dwordx4 v[70:73], v84, s[16:19], s44 offen` / `v_add_f32 v70, ...`): lanes 12-15 of every 16 stored the NEXT
// pair's first value.  The wait states ride on every 16-byte store; tests/test_abi.py scans the built code object for the
// pattern (tools/isa_hazards.py) so that a store added without them fails on the CPU box.
template <int AUX = 0>
__device__ __forceinline__ void buf_store_pair(rsrc_t r, int voff, int soff, cf a, cf b) {
    u32x4 x;
    x.x = __float_as_uint(a.x);
    x.y = __float_as_uint(a.y);
    x.z = __float_as_uint(b.x);
    x.w = __float_as_uint(b.y);
    __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, AUX);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}
#endif

// ---- float64 inter-pass twiddles W_N^p via a two-level table ------------------------------------
struct BigTwiddle {
    const double2* hi;  // W_N^{m << shift}
    const double2* lo;  // W_N^{l}, l < 2^shift
    int shift;
    int64_t mask;  // N - 1 (N a power of two)
    int64_t nmod = 0;  // N when it is NOT a power of two (m * 2^k plans): exponents are reduced with % instead of &
};
__device__ __forceinline__ int64_t tw_reduce(const BigTwiddle& t, int64_t p) { return t.nmod ? p % t.nmod : (p & t.mask); }
__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 big_tw(const BigTwiddle& t, int64_t p) {
    p = tw_reduce(t, p);
    double2 a = t.hi[p >> t.shift];
    double2 b = t.lo[p & ((1LL << t.shift) - 1)];
    return zmul(a, b);
}

}  // namespace PBH_NS
