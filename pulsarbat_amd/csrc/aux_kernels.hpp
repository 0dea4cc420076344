// aux_kernels.hpp -- chirp generation, layout transposes, detection, streaming copy (gfx950).
#pragma once
#include "fft_core.hpp"
#include "bench_kernels.hpp"

namespace PBH_NS {

// ---- K1: chirp (transfer function) ----------------------------------------------------------------
// Reproduces _transfer_function (pulsarbat/transforms/dedispersion.py:19-23) in float64:
//   f   = chan_freq + fftfreq(N, dt)[k]          (numpy: integer bin * (1.0 / (N * dt)))
//   phi = (coeff * f) * (1/ref - 1/f)^2          [cycles]
//   out = complex64(exp(-2 pi i phi)) * scale
// The phase reaches ~5e6 cycles at config 2 (1e8 at DM 1000), so phi is reduced to
// phi - rint(phi) in float64 before sincospi.  Destination order:
//   plan order  (n1 > 1): index chan*N + k1*N2 + k2 for bin k = k1 + N1*k2   (rows of the fused pass)
//   natural order (n1 = 1, N2 = N): index chan*N + k
struct ChirpParams {
    cf* out;
    const double* chan_freq;  // [nchan] device
    double coeff, inv_ndt, inv_ref;
    int64_t N;
    int N1, N2, nchan;
    real scale;
    int perm_w;  // 0: row position e2 holds bin k2 = e2;  W: position e2 holds k2 = e2/(N2/W) + W*(e2 % (N2/W))
    float* phase = nullptr;  // optional second output, same order: the chirp's phase in revolutions, [-0.5, 0.5]
    int P = 1;               // column transform split P x (N1/P): row r of a series holds k1 = r/Q + P*(r%Q)
    const unsigned short* row_perm = nullptr;   // see row_bin
    int phase16 = 0;         // phase rows (2^14 bins) in k_rowp16's order: thread tau's 32 bins tau + 512 i as eight 16-byte groups,
                             // position ((i >> 2) << 11) + 4 tau + (i & 3) -- the row is then read with 16-byte loads
};
__device__ __forceinline__ int64_t phase_index(int64_t d, int N2, int phase16) {
    if (!phase16) return d;
    const int64_t e2 = d % N2;
    const int64_t tau = e2 & 511, i = e2 >> 9;
    return d - e2 + (((i >> 2) << 11) + (tau << 2) + (i & 3));
}

// Row order of the planar work buffer when the column transform of length N1 is split into a radix-P stage
// (k_radix_p) and P blocks of Q = N1/P rows (k_colq): block c, row d within it, holds k1 = c + P*d.
__device__ __forceinline__ int64_t row_k1(int64_t r, int P, int N1) {
    if (P <= 1) return r;
    const int Q = N1 / P;
    return r / Q + (int64_t)P * (r % Q);
}

// row_perm (7-smooth plans whose rows are transformed by k_rowmix): position e2 of a row holds bin row_perm[e2], the
// digit-reversed order its forward decimation-in-frequency stages leave (and its inverse stages start from)
__device__ __forceinline__ int64_t row_bin(int64_t e2, int N2, int perm_w, const unsigned short* row_perm = nullptr) {
    if (row_perm) return row_perm[e2];
    if (perm_w == 0) return e2;
    const int mw = N2 / perm_w;
    return e2 / mw + (int64_t)perm_w * (e2 % mw);
}

__global__ __launch_bounds__(256) void k_chirp(ChirpParams p) {
    // A unit of work is 1024 consecutive positions of one row (chan, r): its indices cost two 64-bit divisions per unit, not
    // three per element (the grid-stride form spent more time on index arithmetic than on the float64 phase).
    const int64_t nrow = p.N / p.N2;
    const int chunks = (p.N2 + 1023) / 1024;
    const int64_t units = (int64_t)p.nchan * nrow * chunks;
    for (int64_t uidx = blockIdx.x; uidx < units; uidx += gridDim.x) {
        const int c = (int)(uidx % chunks);
        const int64_t cr = uidx / chunks;
        const int chan = (int)(cr / nrow);
        const int64_t r = cr - (int64_t)chan * nrow;
        const double fcen = p.chan_freq[chan];
        const int64_t k1 = row_k1(r, p.P, p.N1);
        const int64_t rowbase = (int64_t)chan * p.N + r * p.N2;
        const int e_end = (c + 1) * 1024 < p.N2 ? (c + 1) * 1024 : p.N2;
        for (int e2 = c * 1024 + (int)threadIdx.x; e2 < e_end; e2 += 256) {
            const int64_t k2 = row_bin(e2, p.N2, p.perm_w, p.row_perm);
            const int64_t k = k1 + (int64_t)p.N1 * k2;
            const int64_t bin = (k <= (p.N - 1) / 2) ? k : k - p.N;  // numpy.fft.fftfreq ordering
            const double f = fcen + (double)bin * p.inv_ndt;
            const double dd = p.inv_ref - 1.0 / f;
            const double phi = (p.coeff * f) * (dd * dd);
            const double fr = phi - rint(phi);
            if (p.out) {   // (null: the plan's row pass reads the phase only; the complex form is made when somebody asks for it)
                double sn, cs;
                sincospi(2.0 * fr, &sn, &cs);
                // the reference rounds the transfer function to complex64 (dedispersion.py:23) for both data dtypes
                p.out[rowbase + e2] = make_cf((real)(float)cs * p.scale, (real)(float)(-sn) * p.scale);
            }
            if (p.phase) {   // chirp = exp(2 pi i phase): what k_rowp feeds to v_cos / v_sin
                const int tau = e2 & 511, i = e2 >> 9;
                p.phase[rowbase + (p.phase16 ? (((i >> 2) << 11) + (tau << 2) + (i & 3)) : e2)] = (float)(-fr);
            }
        }
    }
}

// natural (N, nchan) complex64 (the reference's chirp dtype) <-> plan order [chan][k1][k2] in the
// plan's precision; `to_plan` selects the direction.
// (TNAT: element type of the natural-order side -- float2 for the reference's complex64 chirps, double2 for a
//  complex128 chirp handed to the float64 build, which the reference would also keep at that precision)
template <typename TNAT>
__global__ __launch_bounds__(256) void k_chirp_reorder(const TNAT* __restrict__ nat_in, TNAT* __restrict__ nat_out,
                                                       const cf* __restrict__ plan_in, cf* __restrict__ plan_out,
                                                       int64_t N, int N1, int N2, int nchan, real scale,
                                                       int to_plan, int perm_w, int P,
                                                       const unsigned short* __restrict__ row_perm = nullptr) {
    const int64_t total = N * nchan;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < total;
         d += (int64_t)gridDim.x * blockDim.x) {
        const int chan = (int)(d / N);
        const int64_t e = d - (int64_t)chan * N;
        const int64_t r = e / N2, k2 = row_bin(e - r * N2, N2, perm_w, row_perm);
        const int64_t k = row_k1(r, P, N1) + (int64_t)N1 * k2;
        const int64_t nat = k * nchan + chan;
        if (to_plan) {
            const TNAT v = nat_in[nat];
            plan_out[d] = make_cf((real)v.x * scale, (real)v.y * scale);
        } else {
            const cf v = plan_in[d];
            TNAT o;
            o.x = (decltype(o.x))(v.x * scale);
            o.y = (decltype(o.y))(v.y * scale);
            nat_out[nat] = o;
        }
    }
}

// ---- layout transposes (PLANAR5 variant) -------------------------------------------------------------
// (N, S) interleaved -> planar [s][n]; one workgroup moves a contiguous chunk of TN*S elements
// through an LDS tile padded to S+1 per row so the transposed read is conflict-free.
constexpr int kTrElems = 4096;

// nvalid <= N: time samples at and beyond nvalid are not read but taken as zero (the zero padding of the
// convolution plan of an arbitrary-length transform happens here instead of in a copy + memset pass)
__global__ __launch_bounds__(256) void k_deinterleave(const cf* __restrict__ in, cf* __restrict__ out,
                                                      int64_t N, int S, int TN, int64_t plane, int64_t nvalid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const int rows = (int)min((int64_t)TN, N - n0);
    const int cnt = rows * S;
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        int n = e / S, s = e - n * S;
        lds[n * (S + 1) + s] = (n0 + n < nvalid) ? in[n0 * S + e] : make_cf(0, 0);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        int s = e / rows, n = e - s * rows;
        out[(int64_t)s * plane + n0 + n] = lds[n * (S + 1) + s];
    }
}

// planar [s][t] -> (stop-start, S) interleaved, keeping t in [start, stop)
// (opitch: elements between consecutive output rows; S for a compact output, more when the rows are a
//  channel slice of a wider array -- the multi-GPU gather writes a rank's channels into the full-band block)
// (dly: optional per-series time offsets, series s is read at t + dly[s] -- the shifted gather of incoherent dedispersion)
__global__ __launch_bounds__(256) void k_reinterleave(const cf* __restrict__ in, cf* __restrict__ out,
                                                      int64_t start, int64_t stop, int S, int TN,
                                                      int64_t plane, int64_t opitch, const int64_t* __restrict__ dly = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* lds = reinterpret_cast<cf*>(smem);
    const int64_t t0 = start + (int64_t)blockIdx.x * TN;
    const int rows = (int)min((int64_t)TN, stop - t0);
    const int cnt = rows * S;
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        int s = e / rows, n = e - s * rows;
        lds[n * (S + 1) + s] = in[(int64_t)s * plane + t0 + n + (dly ? dly[s] : 0)];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
        int n = e / S, s = e - n * S;
        out[(t0 - start + n) * opitch + s] = lds[n * (S + 1) + s];
    }
}

// (nrow, ncol) block with row pitch ipitch -> rows of another array with row pitch opitch (elements of T): the placing
// pass of the sliced output, and the push of a rank's channel slice into a peer GPU's full-band block
template <typename T>
__global__ __launch_bounds__(256) void k_place(const T* __restrict__ in, int64_t ipitch, T* __restrict__ out, int64_t opitch,
                                               int64_t nrow, int ncol) {
    const int64_t total = nrow * ncol;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = e / ncol;
        const int c = (int)(e - t * ncol);
        out[t * opitch + c] = in[t * ipitch + c];
    }
}

// ---- layout passes for many series (channelised blocks: S = hundreds to thousands) ---------------------------
// With S series a time sample is S*8 bytes of input; tiles of "4096/S time samples x all series" would write
// only a few elements per series.  These kernels tile BOTH axes instead: a workgroup transposes SB series x TB
// time samples (32 KiB) through LDS, reading and writing 512-byte runs on both sides.  Any S (f32: even), any N.
template <int SB, int TB>
__global__ __launch_bounds__(256) void k_deint_blk(const cf* __restrict__ in, cf* __restrict__ out, int64_t N, int S,
                                                   int64_t plane, int64_t nvalid) {
    constexpr int VE = 16 / (int)sizeof(cf);          // elements per 16-byte vector (2 for complex64, 1 for complex128)
    constexpr int LD = TB + 1;
    constexpr int NV = SB * TB / VE / 256;
    typedef float vec16 __attribute__((ext_vector_type(4)));
    __shared__ cf lds[SB * LD];
    const int64_t n0 = (int64_t)blockIdx.x * TB;
    const int s0 = blockIdx.y * SB;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int t = q / (SB / VE), sv = q % (SB / VE);
        const int s = s0 + sv * VE;
        if (n0 + t < N && s < S) {
            union { vec16 v; cf c[VE]; } x;
            if (n0 + t < nvalid) x.v = *reinterpret_cast<const vec16*>(in + (n0 + t) * S + s);
            else x.v = vec16{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < VE; ++e) lds[(sv * VE + e) * LD + t] = x.c[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int sl = q / (TB / VE), t = (q % (TB / VE)) * VE;
        if (s0 + sl < S && n0 + t < N) {
            union { vec16 v; cf c[VE]; } x;
#pragma unroll
            for (int e = 0; e < VE; ++e) x.c[e] = lds[sl * LD + t + e];
            *reinterpret_cast<vec16*>(out + (int64_t)(s0 + sl) * plane + n0 + t) = x.v;
        }
    }
}

// planar [s][t] -> (stop-start, S) interleaved, keeping t in [start, stop)
// DET >= 0 (float32 build): the pass detects instead (see k_reinterleave_p2): a lane's two series are a polarisation pair
template <int SB, int TB, int DET = -1>
__global__ __launch_bounds__(256) void k_reint_blk(const cf* __restrict__ in, cf* __restrict__ out, int64_t start,
                                                   int64_t stop, int S, int64_t plane) {
    constexpr int VE = 16 / (int)sizeof(cf);
    constexpr int LD = TB + 1;
    constexpr int NE = SB * TB / 256, NV = SB * TB / VE / 256;
    typedef float vec16 __attribute__((ext_vector_type(4)));
    __shared__ cf lds[SB * LD];
    const int64_t t0 = start + (int64_t)blockIdx.x * TB;
    const int s0 = blockIdx.y * SB;
#pragma unroll
    for (int j = 0; j < NE; ++j) {   // element loads: t0 need not be 16-byte aligned
        const int q = threadIdx.x + 256 * j;
        const int sl = q / TB, t = q % TB;
        if (s0 + sl < S && t0 + t < stop) lds[sl * LD + t] = in[(int64_t)(s0 + sl) * plane + t0 + t];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int t = q / (SB / VE), sv = q % (SB / VE);
        const int s = s0 + sv * VE;
        if (t0 + t < stop && s < S) {
            union { vec16 v; cf c[VE]; } x;
#pragma unroll
            for (int e = 0; e < VE; ++e) x.c[e] = lds[(sv * VE + e) * LD + t];
            if constexpr (DET >= 0) {
                real* ro = reinterpret_cast<real*>(out);
                const int64_t r = t0 - start + t;
                const cf a = x.c[0], b = x.c[VE - 1];
                const real aa = a.x * a.x + a.y * a.y, bb = b.x * b.x + b.y * b.y;
                if constexpr (DET == 0) {
                    *reinterpret_cast<float2*>(ro + r * S + s) = make_float2(aa, bb);
                } else if constexpr (DET == 1) {
                    ro[r * (S / 2) + s / 2] = aa + bb;
                } else {
                    const real re2 = 2 * (a.x * b.x + a.y * b.y), im2 = 2 * (a.x * b.y - a.y * b.x);   // 2 conj(a) b
                    *reinterpret_cast<float4*>(ro + (r * (S / 2) + s / 2) * 4) =
                        DET == 2 ? make_float4(aa + bb, aa - bb, re2, im2) : make_float4(aa + bb, re2, im2, aa - bb);
                }
            } else
            *reinterpret_cast<vec16*>(out + (t0 - start + t) * S + s) = x.v;
        }
    }
}

// ---- last pass of a stand-alone multi-pass FFT (pbh_fft_c2c beyond one tile) --------------------------------
// After the column pass and the row transforms the spectrum sits in plan order: series s, row r (holding
// k1 = row_k1(r)), position k2 -> bin k = k1 + N1*k2.  This pass writes it in natural order, (n, B)
// interleaved: out[k*B + s].  Seen as a transpose it is k_reint_blk with "series" j = k1*B + s (N1*B of
// them) and "time" k2, so both sides move 512-byte runs.  inverse: out[((N - k) % N)*B + s] = value * scale,
// which turns the forward transform into the unnormalised-inverse-times-scale (ifft with scale = 1/N).
// 16-byte stores when B is a multiple of the elements per 16 bytes, 8-byte stores otherwise.
template <int SB, int TB>
// (row_perm: rows transformed by k_rowmix hold bin row_perm[position] at each position -- see row_bin)
__global__ __launch_bounds__(256) void k_fft_out(const cf* __restrict__ in, cf* __restrict__ out, int N1, int N2, int B,
                                                 int P, int inverse, real scale,
                                                 const unsigned short* __restrict__ row_perm = nullptr) {
    constexpr int VE = 16 / (int)sizeof(cf);
    constexpr int LD = TB + 1;
    constexpr int NE = SB * TB / 256, NV = SB * TB / VE / 256;
    typedef float vec16 __attribute__((ext_vector_type(4)));
    __shared__ cf lds[SB * LD];
    const int64_t k20 = (int64_t)blockIdx.y * TB;
    const int64_t j0 = (int64_t)blockIdx.x * SB, ncol = (int64_t)N1 * B;
    const int Q = N1 / P;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int sl = q / TB, t = q % TB;
        const int64_t col = j0 + sl;
        if (col < ncol && k20 + t < N2) {
            const int64_t k1 = col / B;
            const int s = (int)(col - k1 * B);
            const int64_t r = P > 1 ? (k1 % P) * Q + k1 / P : k1;
            lds[sl * LD + t] = in[((int64_t)s * N1 + r) * N2 + k20 + t];
        }
    }
    __syncthreads();
    const int64_t N = (int64_t)N1 * N2;
    if (VE > 1 && B % VE != 0) {   // odd number of complex64 series: 8-byte stores, still contiguous across the lanes
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int q = threadIdx.x + 256 * j;
            const int t = q / SB, sl = q % SB;
            const int64_t col = j0 + sl;
            if (k20 + t < N2 && col < ncol) {
                const cf a = lds[sl * LD + t];
                const int64_t k1 = col / B;
                const int s = (int)(col - k1 * B);
                int64_t k = k1 + (int64_t)N1 * (row_perm ? (int64_t)row_perm[k20 + t] : k20 + t);
                if (inverse) k = k ? N - k : 0;
                out[k * B + s] = make_cf(a.x * scale, a.y * scale);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int t = q / (SB / VE), sv = q % (SB / VE);
        const int64_t col = j0 + (int64_t)sv * VE;
        if (k20 + t < N2 && col < ncol) {
            union { vec16 v; cf c[VE]; } x;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const cf a = lds[(sv * VE + e) * LD + t];
                x.c[e] = make_cf(a.x * scale, a.y * scale);
            }
            const int64_t k1 = col / B;
            const int s = (int)(col - k1 * B);
            int64_t k = k1 + (int64_t)N1 * (row_perm ? (int64_t)row_perm[k20 + t] : k20 + t);
            if (inverse) k = k ? N - k : 0;
            *reinterpret_cast<vec16*>(out + k * B + s) = x.v;   // B % VE == 0: the VE elements share one k
        }
    }
}

// ---- reader-side decode: raw integer payload -> float32 / complex64 (nsample, nchan, npol) -------------------------
// What the reference's readers do on the host after `baseband` has unpacked a payload
// (pulsarbat/readers/_baseband_readers.py:136-153: sideband conjugation + astype; :223-226, :268-275: axis
// transposes / channel flip) done in the pass that unpacks it, so the raw bytes (1/4 .. 1/16 of the complex64
// size) are what crosses PCIe.  The payload is addressed in place: the byte range of a file is a sequence of
// blocks (header + payload, `blk_stride` apart) each holding `blk_t` time samples, and inside a payload element
// (t, chan, pol) sits at element index e0 + t*st_t + chan*st_c + pol*st_p (strides may be negative: channel flip),
// an element being 1 (real) or 2 (complex, re then im) components of `nbits` bits, low bits first in each byte.
struct DecodeParams {
    const unsigned char* raw;   // device copy of bytes [skip, ...) of the payload stream
    int64_t skip;        // stream offset of raw[0]: only the blocks that are needed are on the device
    int64_t first;       // first wanted time sample, in valid samples from the start of the stream
    int64_t blk_t;       // time samples per block
    int64_t blk_stride;  // bytes from one block to the next
    int64_t hdr;         // bytes from the start of a block to its payload
    int64_t e0, st_t, st_c, st_p;
    int nbits;           // 8, 4 or 2
    int code;            // nbits 8: 0 two's complement, 1 offset binary (v - 128); nbits 4: offset binary (v - 8);
                         // nbits 2: 4-level VDIF table
    int lanes_t;         // consecutive lanes read consecutive time samples (else consecutive series)
    int ls;              // log2 of the series per tile
    int npol_shift;      // log2(npol), or -1
    int pair16;          // 8-bit complex elements all at even addresses
    float scale;
    const unsigned char* conj;   // per series: negate the imaginary part (lower sideband); may be null
    int64_t n;
    int nchan, npol;
    float* out;
    int series_major;    // out[s*pitch + t] (time fastest) instead of out[t*S + s]
    int64_t pitch;
};

// One workgroup unpacks a tile of 4096 elements, TS = 2^ls series by TT = 4096/TS time samples (TS = the series
// count rounded up to a power of two, at most 64, so that no lanes idle when there are few series), through LDS:
// consecutive lanes read along the payload's fastest axis and write along the output's.
constexpr int kDecodeTile = 4096;
constexpr int kDecodeLds = 5120;   // TS >= 4 rows are padded by one element

// FAST: blocks at least as long as a tile's time extent (one wrap at most), npol a power of two and, for 8-bit
// complex data, elements at even addresses (one 16-bit load) -- the loop body is then free of branches and all
// 16 loads of a thread are in flight together; the general form keeps the divisions and byte loads.
template <int NC, int NBITS, bool FAST>
__global__ __launch_bounds__(256) void k_decode(DecodeParams q) {
    __shared__ float lds[NC][kDecodeLds];
    constexpr int NJ = kDecodeTile / 256;
    constexpr bool PAIR = FAST && NC == 2 && NBITS == 8;
    const int ls = q.ls, TS = 1 << ls, lt = 12 - ls, TT = 1 << lt;
    const int pitch = TS < 4 ? TS : TS + 1;
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int s0 = blockIdx.y * TS, S = q.nchan * q.npol;
    const int64_t g0 = q.first + t0, blk0 = g0 / q.blk_t, w0 = g0 - blk0 * q.blk_t;   // uniform: once per workgroup
    // indices of lanes beyond the data are clamped to the last sample / series (addresses the host has
    // bounds-checked) instead of branching around the load
    const int64_t tlim = q.n - 1 - t0;   // >= 0: the grid covers [0, n)
    unsigned raw[NC][NJ];
    int shift[NC][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = threadIdx.x + 256 * j;
        int tt = q.lanes_t ? (idx & (TT - 1)) : (idx >> ls), ss = q.lanes_t ? (idx >> lt) : (idx & (TS - 1));
        tt = tt > tlim ? (int)tlim : tt;
        const int s = s0 + ss < S ? s0 + ss : S - 1;
        int64_t w = w0 + tt, blk = blk0;
        int c;
        if (FAST) {
            const bool wrap = w >= q.blk_t;
            w -= wrap ? q.blk_t : 0;
            blk += wrap ? 1 : 0;
            c = s >> q.npol_shift;
        } else {
            const int64_t d = w / q.blk_t;
            blk += d;
            w -= d * q.blk_t;
            c = s / q.npol;
        }
        const int p = s - c * q.npol;
        const int64_t e = q.e0 + w * q.st_t + c * q.st_c + p * q.st_p;
        const unsigned char* pay = q.raw + (blk * q.blk_stride + q.hdr - q.skip);
        if (PAIR) {
            raw[0][j] = *reinterpret_cast<const unsigned short*>(pay + 2 * e);
        } else {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int64_t ci = e * NC + k;
                raw[k][j] = NBITS == 8 ? pay[ci] : (NBITS == 4 ? pay[ci >> 1] : pay[ci >> 2]);
                shift[k][j] = NBITS == 4 ? 4 * (int)(ci & 1) : 2 * (int)(ci & 3);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = threadIdx.x + 256 * j;
        const int tt = q.lanes_t ? (idx & (TT - 1)) : (idx >> ls), ss = q.lanes_t ? (idx >> lt) : (idx & (TS - 1));
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float val;
            if (NBITS == 8) {
                const int v = PAIR ? (int)((raw[0][j] >> (8 * k)) & 0xFF) : (int)raw[k][j];
                val = q.code ? (float)(v - 128) : (float)(signed char)v;
            } else if (NBITS == 4) {
                val = (float)((int)((raw[k][j] >> shift[k][j]) & 15) - 8);   // offset binary (VDIF)
            } else {
                const int c2 = (raw[k][j] >> shift[k][j]) & 3;
                // high level of the 4-level code: baseband's OPTIMAL_2BIT_HIGH = 3.316505 (baseband/base/encoding.py: the mean of
                // the samples beyond 1 sigma over the mean of those within, for a normal distribution) -- the reference
                // reads VDIF through baseband; mark5access would use 3.3359
                const float mag = (c2 == 0 || c2 == 3) ? 3.316505f : 1.0f;
                val = (c2 & 2) ? mag : -mag;
            }
            lds[k][tt * pitch + ss] = val;   // (clamped lanes fill their own, unused, slot)
        }
    }
    __syncthreads();
    const bool has_conj = q.conj != nullptr;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = threadIdx.x + 256 * j;
        const int tt = q.series_major ? (idx & (TT - 1)) : (idx >> ls), ss = q.series_major ? (idx >> lt) : (idx & (TS - 1));
        const int64_t t = t0 + tt;
        const int s = s0 + ss;
        if (t < q.n && s < S) {
            const int64_t o = q.series_major ? (int64_t)s * q.pitch + t : t * S + s;
            const float re = lds[0][tt * pitch + ss] * q.scale;
            if (NC == 2) {
                float im = lds[NC - 1][tt * pitch + ss] * q.scale;
                if (has_conj && q.conj[s]) im = -im;
                reinterpret_cast<float2*>(q.out)[o] = make_float2(re, im);
            } else {
                q.out[o] = re;
            }
        }
    }
}

// contrib.stft / istft with segments longer than one tile: the segments are a batch of native-length transforms
// (de-interleave, column pass, row transforms as in fft_c2c_native) and this pass writes the spectra where the
// reference's reshapes put them.  Plan order in: batch row b, row r (bin k1, split order P), position k2
// -> bin k = k1 + N1*k2.  Tile: 64 "columns" J x TB positions k2 through LDS; J runs over everything that is
// contiguous in the output for one k2, so both sides move 512-byte runs.
//   stft  (inverse = 0): b = (c*E + e)*nseg + g;  J = ((g*nchan + c)*N1 + k1)*E + e;
//                        out[((g*nchan + c)*n + (k + n/2) % n)*E + e] = value * scale        (fftshift, 1/n)
//   istft (inverse = 1): b = e*nseg*nchan + g*nchan + c;  J = ((g*N1 + k1)*nchan + c)*E + e;
//                        out[((g*n + t)*nchan + c)*E + e] = (-1)^k value * scale,  t = (n - k) % n
//                        (the input rows were stored fftshift-ed: a factor (-1)^t = (-1)^k; the inverse transform is
//                         the forward one read at index n - t)
template <int TB>
__global__ __launch_bounds__(256) void k_stft_out(const cf* __restrict__ in, cf* __restrict__ out, int N1, int N2, int P,
                                                 int64_t nseg, int nchan, int E, int inverse, real scale,
                                                  const unsigned short* __restrict__ row_perm = nullptr) {
    constexpr int SB = 64, LD = TB + 1, NE = SB * TB / 256;
    __shared__ cf lds[SB * LD];
    __shared__ int64_t src_row[SB];
    const int64_t k20 = (int64_t)blockIdx.y * TB, J0 = (int64_t)blockIdx.x * SB;
    const int S = nchan * E, Q = N1 / P;
    const int64_t n = (int64_t)N1 * N2, NJ = nseg * S * N1;
    // this thread's column (the same for every element it stores; the first 64 threads also publish its source row)
    const int sl_own = threadIdx.x % SB;
    const int64_t J = J0 + sl_own;
    int64_t g = 0, dst_base = 0;
    int c = 0, e = 0, k1 = 0;
    if (J < NJ) {
        if (!inverse) {
            const int64_t gc = J / ((int64_t)N1 * E);
            const int rem = (int)(J - gc * N1 * E);
            k1 = rem / E;
            e = rem - k1 * E;
            g = gc / nchan;
            c = (int)(gc - g * nchan);
            dst_base = gc * n * E + e;                       // + kk * E
        } else {
            g = J / ((int64_t)N1 * S);
            const int rem = (int)(J - g * N1 * S);
            k1 = rem / S;
            const int sq = rem - k1 * S;
            c = sq / E;
            e = sq - c * E;
            dst_base = g * n * S + sq;                       // + t * S
        }
        if (threadIdx.x < SB) {
            const int64_t b = inverse ? ((int64_t)e * nseg + g) * nchan + c : ((int64_t)c * E + e) * nseg + g;
            const int r = P > 1 ? (k1 % P) * Q + k1 / P : k1;
            src_row[sl_own] = (b * N1 + r) * N2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int q = threadIdx.x + 256 * j;
        const int sl = q / TB, t = q % TB;
        if (J0 + sl < NJ && k20 + t < N2) lds[sl * LD + t] = in[src_row[sl] + k20 + t];
    }
    __syncthreads();
    if (J >= NJ) return;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int t = (threadIdx.x + 256 * j) / SB;
        const int64_t k2 = k20 + t;
        if (k2 < N2) {
            const cf a = lds[sl_own * LD + t];
            const int64_t k = k1 + (int64_t)N1 * (row_perm ? (int64_t)row_perm[k2] : k2);   // (rows of k_rowmix: see row_bin)
            if (!inverse) {
                const int64_t kk = k + n / 2 < n ? k + n / 2 : k + n / 2 - n;   // fftshift, odd n too
                out[dst_base + kk * E] = make_cf(a.x * scale, a.y * scale);
            } else {
                const int64_t tt = k ? n - k : 0;
                const real sg = (k & 1) ? -scale : scale;
                out[dst_base + tt * S] = make_cf(a.x * sg, a.y * sg);
            }
        }
    }
}

#ifndef PBH_F64
// Power-of-two S fast paths: a tile is TN = 4096/S time samples (32 KiB), 256 threads, each thread
// moves 8 float4 (two complex) per side with all loads issued before the first use -- the shape the
// streaming-copy calibration (tools/micro/membench.hip) found fastest on MI355X.
// x * exp(2 pi i ft n): float64 phase reduced to one revolution, then the hardware's cos / sin (max error 1.4e-7, the
// size of the complex64 rounding the reference applies to its phasor, transforms.py:346)
__device__ __forceinline__ cf mix_sample(cf x, double ft, int64_t n) {
    const double phi = ft * (double)n;
    const float fr = (float)(phi - rint(phi));
    return cmul(x, make_cf(__builtin_amdgcn_cosf(fr), __builtin_amdgcn_sinf(fr)));
}

// (MIX: freq_shift's mixer folded into the pass -- every sample is multiplied by exp(2 pi i ft[s] n) on its way
//  through, which saves the copy and the mixing pass in front of the pipeline)
template <int S, bool MIX = false>
__global__ __launch_bounds__(256) void k_deinterleave_p2(const cf* __restrict__ in, cf* __restrict__ out,
                                                         int64_t N, int64_t plane, int64_t nvalid,
                                                         const double* __restrict__ ft = nullptr) {
    // LDS image: series s at s * LD.  2 <= S <= 16: no padding, every series' row ROTATED by an even amount instead
    // (slot = s TN + (n + c(s)) mod TN, c = (s >> 1)(32 / S)): the 16 lanes of a ds_write_b64 group -- 32 / S time samples of S / 2
    // even series -- land on 16 different slots mod 16, and a lane's two samples of the planar side are one aligned 16-byte
    // read.  With LD = TN + 1 that read was a ds_read2_b64 whose lanes are two slots apart: 2-way bank conflicts, 1.68e7
    // SQ_LDS_BANK_CONFLICT cycles per launch at config 2 (profiles/r02z3_*).  Other S keep the padded image.
    constexpr bool SWZ = S >= 2 && S <= 16;
    constexpr int TN = kTrElems / S, LD = SWZ ? TN : TN + 1, NV = kTrElems / 2 / 256;  // 8 float4 per thread
    __shared__ __attribute__((aligned(16))) cf lds[S * LD];
    auto slot = [](int s, int n) { return SWZ ? s * TN + ((n + (s >> 1) * (32 / (SWZ ? S : 32))) & (TN - 1)) : s * LD + n; };
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const float4* src = reinterpret_cast<const float4*>(in + n0 * S);
    float4 v[NV];
    if (n0 + TN <= nvalid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = src[threadIdx.x + 256 * j];
    } else {   // tile at or beyond the end of the data: zero padding
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = 2 * (threadIdx.x + 256 * j);
            if (S == 1) {
                // the two halves of a vector are consecutive time samples: with an odd number of samples the last
                // vector holds one of them (read on its own: the element after it is not part of the input)
                const int64_t t = n0 + e;
                const cf a = t < nvalid ? in[t] : make_cf(0, 0), b = t + 1 < nvalid ? in[t + 1] : make_cf(0, 0);
                v[j] = make_float4(a.x, a.y, b.x, b.y);
            } else {
                v[j] = (n0 + e / S < nvalid) ? src[threadIdx.x + 256 * j] : make_float4(0, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = 2 * (threadIdx.x + 256 * j);
        const int n = e / S, s = e % S;
        if (S == 1) {  // two consecutive times of the single series
            lds[e] = make_cf(v[j].x, v[j].y);
            lds[e + 1] = make_cf(v[j].z, v[j].w);
        } else {
            lds[slot(s, n)] = make_cf(v[j].x, v[j].y);
            lds[slot(s + 1, n)] = make_cf(v[j].z, v[j].w);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int pidx = threadIdx.x + 256 * j;
        const int s = pidx / (TN / 2), n = 2 * (pidx % (TN / 2));
        cf a, b;
        if constexpr (SWZ) {
            const float4 q = *reinterpret_cast<const float4*>(&lds[slot(s, n)]);
            a = make_cf(q.x, q.y);
            b = make_cf(q.z, q.w);
        } else {
            a = lds[s * LD + n];
            b = lds[s * LD + n + 1];
        }
        if constexpr (MIX) {
            const double f = ft[s];
            a = mix_sample(a, f, n0 + n);
            b = mix_sample(b, f, n0 + n + 1);
        }
        *reinterpret_cast<float4*>(out + (int64_t)s * plane + n0 + n) = make_float4(a.x, a.y, b.x, b.y);
    }
}

// planar [s][t] -> (stop-start, S); the tail tile is handled by the generic kernel
// DET >= 0 (float32 build): the pass DETECTS instead -- the two series a lane holds for one time sample are the two
// polarisations of a channel (series = chan * npol + pol), so what leaves is float32 at full time resolution,
//   0: |z|^2 per element (rows of S floats);  1: Stokes I (rows of S/2);  2 / 3: I, Q, U, V linear / circular (rows of 2 S)
// -- pulsarbat's to_intensity / to_stokes (core.py:766-774, 930-966) of voltages that are never stored.
template <int S, bool PITCHED = false, bool SHIFTED = false, int DET = -1>
__global__ __launch_bounds__(256) void k_reinterleave_p2(const cf* __restrict__ in, cf* __restrict__ out,
                                                         int64_t start, int64_t plane, int64_t opitch,
                                                         const int64_t* __restrict__ dly = nullptr) {
    // LDS image as in k_deinterleave_p2 (same rotation): the interleaved side's two reads of a lane, series s and s + 1 at the
    // same rotated index, are exactly TN slots apart and become ONE ds_read2st64_b64, whose accesses go in 16-lane groups like
    // the writes there; the planar side writes 16 aligned bytes.
    constexpr bool SWZ = S >= 2 && S <= 16;
    constexpr int TN = kTrElems / S, LD = SWZ ? TN : TN + 1, NV = kTrElems / 2 / 256;
    __shared__ __attribute__((aligned(16))) cf lds[S * LD];
    auto slot = [](int s, int n) { return SWZ ? s * TN + ((n + (s >> 1) * (32 / (SWZ ? S : 32))) & (TN - 1)) : s * LD + n; };
    const int64_t t0 = start + (int64_t)blockIdx.x * TN;
    cf a[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int pidx = threadIdx.x + 256 * j;
        const int s = pidx / (TN / 2), n = 2 * (pidx % (TN / 2));
        const cf* p = in + (int64_t)s * plane + t0 + n;
        if constexpr (SHIFTED) p += dly[s];   // (odd shifts break the 16-byte alignment: the loads stay 8 bytes wide)
        a[j] = p[0];
        b[j] = p[1];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int pidx = threadIdx.x + 256 * j;
        const int s = pidx / (TN / 2), n = 2 * (pidx % (TN / 2));
        if constexpr (SWZ) {
            *reinterpret_cast<float4*>(&lds[slot(s, n)]) = make_float4(a[j].x, a[j].y, b[j].x, b[j].y);
        } else {
            lds[s * LD + n] = a[j];
            lds[s * LD + n + 1] = b[j];
        }
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(out + (t0 - start) * S);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = 2 * (threadIdx.x + 256 * j);
        const int n = e / S, s = e % S;
        cf x, y;
        if (S == 1) {
            x = lds[e];
            y = lds[e + 1];
        } else {
            x = lds[slot(s, n)];
            y = lds[slot(s + 1, n)];
        }
        if constexpr (DET >= 0) {
            real* ro = reinterpret_cast<real*>(out);
            const int64_t r = t0 - start + n;
            const real aa = x.x * x.x + x.y * x.y, bb = y.x * y.x + y.y * y.y;
            if constexpr (DET == 0) {
                *reinterpret_cast<float2*>(ro + r * S + s) = make_float2(aa, bb);
            } else if constexpr (DET == 1) {
                ro[r * (S / 2) + s / 2] = aa + bb;
            } else {
                const real re2 = 2 * (x.x * y.x + x.y * y.y), im2 = 2 * (x.x * y.y - x.y * y.x);   // 2 conj(a) b
                *reinterpret_cast<float4*>(ro + (r * (S / 2) + s / 2) * 4) =
                    DET == 2 ? make_float4(aa + bb, aa - bb, re2, im2) : make_float4(aa + bb, re2, im2, aa - bb);
            }
        } else
        if constexpr (PITCHED)   // rows are a slice of a wider array (S >= 2, opitch and the base even: 16-byte vectors)
            *reinterpret_cast<float4*>(out + (t0 - start + n) * opitch + s) = make_float4(x.x, x.y, y.x, y.y);
        else
            dst[threadIdx.x + 256 * j] = make_float4(x.x, x.y, y.x, y.y);
    }
}

#endif  // !PBH_F64

// ---- layout passes with the radix-P stage of a long column transform folded in ----------------------------
// (see k_radix_p).  A workgroup handles the same TN time samples of all P chunks (N/P apart): forward it
// reads P interleaved tiles, does the P-point DFT + twiddle across them in registers and writes P planar
// tiles; inverse it reads P planar tiles, undoes the stage and writes P interleaved tiles, cropped.  Long
// blocks are back to five passes.  E elements per chunk tile, chosen so a thread holds <= 32 float4.
constexpr int radix_tile_e(int P) { return P <= 4 ? 4096 : (P <= 8 ? 2048 : 1024); }   // (the float32 inverse kernel at P = 7 spills 390 B/lane; 1024-element tiles are slower still: 54 vs 59 Gsamples/s)
template <int P>
struct RadixTile {
    static constexpr int E = radix_tile_e(P);
};
template <int P>
__device__ __forceinline__ void radix_twiddles(cf (&tw)[P], int64_t b, int N1, int dir) {
    double sn, cs;
    sincospi(2.0 * (double)b / (double)N1, &sn, &cs);
    const double2 w1 = make_double2(cs, dir < 0 ? -sn : sn);
    double2 w = w1;
    tw[0] = make_cf(1, 0);
#pragma unroll
    for (int c = 1; c < P; ++c) {
        tw[c] = make_cf((real)w.x, (real)w.y);
        w = make_double2(w.x * w1.x - w.y * w1.y, w.x * w1.y + w.y * w1.x);
    }
}

// (both precisions: a thread moves 16-byte vectors, VE = 2 complex64 or 1 complex128 elements, and a chunk tile
//  is the same 8 * RadixTile<P>::E bytes)
template <int S, int P>
__global__ __launch_bounds__(256) void k_deint_radix(const cf* __restrict__ in, cf* __restrict__ out, int64_t chunk,
                                                     int64_t plane, int N2, int N1, int64_t nvalid) {
    constexpr int VE = 16 / (int)sizeof(cf);
    constexpr int E = RadixTile<P>::E * 8 / (int)sizeof(cf), TN = E / S, LD = TN + 1, NV = E / VE / 256;
    typedef real vecr __attribute__((ext_vector_type(2 * VE)));
    static_assert(S % VE == 0 && TN % VE == 0 && NV >= 1, "k_deint_radix: tile shape");
    __shared__ cf lds[S * LD];
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    vecr v[P][NV];
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const vecr* src = reinterpret_cast<const vecr*>(in + ((int64_t)a * chunk + n0) * S);
        const int64_t t0 = (int64_t)a * chunk + n0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = VE * (threadIdx.x + 256 * j);
            v[a][j] = (t0 + e / S < nvalid) ? src[threadIdx.x + 256 * j] : (vecr)(real)0;   // zero padding
        }
    }
    cf tw[P];
    radix_twiddles<P>(tw, n0 / N2, N1, -1);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int h = 0; h < VE; ++h) {
            cf x[P];
#pragma unroll
            for (int a = 0; a < P; ++a) x[a] = make_cf(v[a][j][2 * h], v[a][j][2 * h + 1]);
            Dft<P, -1>::run(x);
#pragma unroll
            for (int c = 1; c < P; ++c) x[c] = cmul(x[c], tw[c]);
#pragma unroll
            for (int c = 0; c < P; ++c) {
                v[c][j][2 * h] = x[c].x;
                v[c][j][2 * h + 1] = x[c].y;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < P; ++c) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = VE * (threadIdx.x + 256 * j);
            const int n = e / S, s = e % S;
#pragma unroll
            for (int h = 0; h < VE; ++h) lds[(s + h) * LD + n] = make_cf(v[c][j][2 * h], v[c][j][2 * h + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pidx = threadIdx.x + 256 * j;
            const int s = pidx / (TN / VE), n = VE * (pidx % (TN / VE));
            vecr o;
#pragma unroll
            for (int h = 0; h < VE; ++h) {
                const cf t = lds[s * LD + n + h];
                o[2 * h] = t.x;
                o[2 * h + 1] = t.y;
            }
            *reinterpret_cast<vecr*>(out + (int64_t)s * plane + (int64_t)c * chunk + n0 + n) = o;
        }
        __syncthreads();
    }
}

template <int S, int P>
__global__ __launch_bounds__(256) void k_reint_radix(const cf* __restrict__ in, cf* __restrict__ out, int64_t chunk,
                                                     int64_t plane, int N2, int N1, int64_t start, int64_t stop) {
    constexpr int VE = 16 / (int)sizeof(cf);
    constexpr int E = RadixTile<P>::E * 8 / (int)sizeof(cf), TN = E / S, LD = TN + 1, NV = E / VE / 256;
    typedef real vecr __attribute__((ext_vector_type(2 * VE)));
    __shared__ cf lds[S * LD];
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    // nothing to do when none of the P tiles reaches the kept range
    bool any = false;
#pragma unroll
    for (int a = 0; a < P; ++a) any |= ((int64_t)a * chunk + n0 < stop) && ((int64_t)a * chunk + n0 + TN > start);
    if (!any) return;
    // the stage's twiddles first: sincospi in float64 needs ~100 registers of its own, which must not coincide with the P * NV
    // vectors of the loaded tile (computed after the loads, as before, the P = 7 kernel spilled 392 B/lane)
    cf tw[P];
    radix_twiddles<P>(tw, n0 / N2, N1, +1);
    if constexpr (P >= 5) __builtin_amdgcn_sched_barrier(0);
    vecr v[P][NV];
    // chunk c + 1 is requested while chunk c goes through LDS, and no further ahead: with every chunk's loads hoisted to the
    // top (what the compiler does with the unrolled loop) 2 P NV vectors are live at once and the P = 7 kernel spilled 392 B/lane
    vecr t[2][NV];
    auto request = [&](int c, vecr (&dst)[NV]) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pidx = threadIdx.x + 256 * j;
            const int s = pidx / (TN / VE), n = VE * (pidx % (TN / VE));
            dst[j] = *reinterpret_cast<const vecr*>(in + (int64_t)s * plane + (int64_t)c * chunk + n0 + n);
        }
    };
    request(0, t[0]);
#pragma unroll
    for (int c = 0; c < P; ++c) {
        if (c + 1 < P) request(c + 1, t[(c + 1) & 1]);
        if constexpr (P >= 5) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int pidx = threadIdx.x + 256 * j;
            const int s = pidx / (TN / VE), n = VE * (pidx % (TN / VE));
#pragma unroll
            for (int h = 0; h < VE; ++h) lds[s * LD + n + h] = make_cf(t[c & 1][j][2 * h], t[c & 1][j][2 * h + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = VE * (threadIdx.x + 256 * j);
            const int n = e / S, s = e % S;
#pragma unroll
            for (int h = 0; h < VE; ++h) {
                const cf t = lds[(s + h) * LD + n];
                v[c][j][2 * h] = t.x;
                v[c][j][2 * h + 1] = t.y;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int h = 0; h < VE; ++h) {
            cf x[P];
#pragma unroll
            for (int c = 0; c < P; ++c) x[c] = make_cf(v[c][j][2 * h], v[c][j][2 * h + 1]);
#pragma unroll
            for (int c = 1; c < P; ++c) x[c] = cmul(x[c], tw[c]);
            Dft<P, +1>::run(x);
#pragma unroll
            for (int a = 0; a < P; ++a) {
                v[a][j][2 * h] = x[a].x;
                v[a][j][2 * h + 1] = x[a].y;
            }
            // one DFT at a time: left alone, the scheduler interleaves all NV * VE of them (the odd ones are O(P^2) products
            // with literal roots) and the float32 kernel at P = 7 spilled 392 B/lane
            if constexpr (P >= 5) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Cropped samples are dropped by an out-of-range buffer offset, not by a branch: with `if (in range) store` the compiler sank
    // the whole radix stage INTO the 28 branches of the P = 7 kernel (a copy of the DFT per store) and spilled 392 B/lane.
    // (The descriptor's base may lie before `out` when a chunk starts inside the cropped head: only in-range offsets are used.)
#pragma unroll
    for (int a = 0; a < P; ++a) {
        const int64_t t0 = (int64_t)a * chunk + n0;
        const rsrc_t ro = make_rsrc(out + (t0 - start) * S, (uint32_t)(E * sizeof(cf)));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = VE * (threadIdx.x + 256 * j);
            const int64_t t = t0 + e / S;
            union { vecr r; u32x4 u; } x;
            x.r = v[a][j];
            __builtin_amdgcn_raw_buffer_store_b128(x.u, ro, (t >= start && t < stop) ? e * (int)sizeof(cf) : (int)0x80000000, 0, 0);
        }
    }
}

// ---- detection (pulsarbat/core.py:766-774, 930-966), optional time scrunch -----------------------------
// in: (n, nchan, npol) c64.  One thread per (output row, chan); sums nscrunch input rows in float32.
__global__ __launch_bounds__(256) void k_detect(const cf* __restrict__ in, real* __restrict__ out,
                                                int64_t nout, int nchan, int npol, int mode, int nscrunch) {
    const int64_t total = nout * nchan;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < total;
         d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = d / nchan;
        const int chan = (int)(d - o * nchan);
        if (mode == 0) {
            for (int pp = 0; pp < npol; ++pp) {
                real acc = 0;
                for (int j = 0; j < nscrunch; ++j) {
                    cf a = in[((o * nscrunch + j) * nchan + chan) * npol + pp];
                    acc += a.x * a.x + a.y * a.y;
                }
                out[d * npol + pp] = acc;
            }
        } else {
            real si = 0, sq = 0, su = 0, sv = 0;
            for (int j = 0; j < nscrunch; ++j) {
                const cf* base = in + ((o * nscrunch + j) * nchan + chan) * 2;
                cf a = base[0], b = base[1];
                real aa = a.x * a.x + a.y * a.y, bb = b.x * b.x + b.y * b.y;
                real re = a.x * b.x + a.y * b.y;  // Re(conj(a) b)
                real im = a.x * b.y - a.y * b.x;  // Im(conj(a) b)
                si += aa + bb;
                if (mode == 2) {
                    sq += aa - bb; su += 2 * re; sv += 2 * im;
                } else if (mode == 3) {
                    sq += 2 * re; su += 2 * im; sv += aa - bb;
                }
            }
            if (mode == 1) {
                out[d] = si;
            } else {
                real* o4 = out + d * 4;
                o4[0] = si; o4[1] = sq; o4[2] = su; o4[3] = sv;
            }
        }
    }
}

// ---- fused detection + time scrunch from the planar work buffer (tail of the PLANAR5 sequence) ------------
// Replaces k_reinterleave when the caller wants detected, time-scrunched output (BASELINE configs[4]):
// the dedispersed voltages are never written in the reference layout.  One wavefront owns one output
// bin of one channel: lanes stride over the bin's nscrunch consecutive samples (coalesced 512-B reads
// per polarisation), accumulate in float32 and finish with a 64-lane butterfly reduction.
//   mode 0: |z|^2 per pol -> out[o][chan][pol];  1: Stokes I -> out[o][chan];  2/3: IQUV -> out[o][chan][4]
__device__ __forceinline__ real wave_sum(real x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__global__ __launch_bounds__(256) void k_detect_planar(const cf* __restrict__ work, real* __restrict__ out,
                                                       int64_t plane, int64_t start, int64_t nout, int nchan,
                                                       int npol, int mode, int nscrunch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t o = (int64_t)blockIdx.x * 4 + wave;
    const int chan = blockIdx.y;
    if (o >= nout) return;
    const int64_t t0 = start + o * nscrunch;
    if (mode == 0) {
        for (int pp = 0; pp < npol; ++pp) {
            const cf* a = work + (int64_t)(chan * npol + pp) * plane + t0;
            real acc = 0;
            for (int i = lane; i < nscrunch; i += 64) acc += a[i].x * a[i].x + a[i].y * a[i].y;
            acc = wave_sum(acc);
            if (lane == 0) out[(o * nchan + chan) * npol + pp] = acc;
        }
        return;
    }
    const cf* pa = work + (int64_t)(chan * 2) * plane + t0;
    const cf* pb = pa + plane;
    real si = 0, sq = 0, su = 0, sv = 0;
    for (int i = lane; i < nscrunch; i += 64) {
        const cf a = pa[i], b = pb[i];
        const real aa = a.x * a.x + a.y * a.y, bb = b.x * b.x + b.y * b.y;
        si += aa + bb;
        if (mode >= 2) {
            const real re = a.x * b.x + a.y * b.y, im = a.x * b.y - a.y * b.x;  // conj(a) * b
            if (mode == 2) { sq += aa - bb; su += 2 * re; sv += 2 * im; }
            else { sq += 2 * re; su += 2 * im; sv += aa - bb; }
        }
    }
    si = wave_sum(si);
    if (mode == 1) {
        if (lane == 0) out[o * nchan + chan] = si;
        return;
    }
    sq = wave_sum(sq); su = wave_sum(su); sv = wave_sum(sv);
    if (lane == 0) {
        real* o4 = out + (o * nchan + chan) * 4;
        o4[0] = si; o4[1] = sq; o4[2] = su; o4[3] = sv;
    }
}

// ---- second half of the detect tail fused into the inverse column pass (k_colq<.., DET>) ---------------------------
// The column pass left, per (series, 16-column group g, row r), the power summed over the group's columns
// (ColpParams::det_part / det_side).  Output sample o covers times [start + o ns, start + (o + 1) ns), t = r N2 + n2:
// ns / 16 consecutive groups of one row (the run may continue in the next row), plus -- when start is not a multiple of 16
// -- the columns behind the boundary of its first group (det_side) and the columns before the boundary of the group
// after its last one (what det_part holds for a group with a boundary).  Consecutive threads follow the partial sums'
// row order (row = x / R + MR (x % R)): coalesced reads.
//   mode 0: out[o][chan][pol];  mode 1 (Stokes I): out[o][chan], polarisations added.
// Round 4: the group width is a parameter (gw = 16: one float per series; gw = 8, ncomp = 4: the pol-pair form of the column
// pass, planes |a|^2, |b|^2, Re conj(a) b, Im conj(a) b per channel) and the four-parameter modes are assembled here:
//   mode 0: out[o][chan][pol];  mode 1 (Stokes I): out[o][chan];  mode 2 / 3 (linear / circular, core.py:930-966): out[o][chan][4].
__global__ __launch_bounds__(256) void k_detect_reduce(const real* __restrict__ part, const real* __restrict__ side,
                                                       real* __restrict__ out, int N2, int M, int R, int ns, int64_t start,
                                                       int64_t nout, int nchan, int npol, int mode, int parts, int gw, int ncomp) {
    // 256 / parts rows x `parts` (1, 4 or 16) pieces of an output's run of groups per workgroup (the pieces meet in LDS):
    // short runs (small scrunch factors) want many rows per workgroup, long ones many pieces
    __shared__ real sh[256];
    const int xw = 256 / parts;
    const int xl = threadIdx.x % xw, qu = threadIdx.x / xw;
    const int x = blockIdx.x * xw + xl;
    const int kk = blockIdx.y, chan = blockIdx.z;
    const int MR = M / R;
    const int r0 = x / R + MR * (x % R);
    const int bmod = (int)(start % ns), bcol = bmod & (gw - 1);
    const int64_t t0 = (int64_t)r0 * N2 + (int64_t)kk * ns + bmod;
    const int64_t o = t0 >= start ? (t0 - start) / ns : -1;
    const bool live = x < M && o >= 0 && o < nout;
    const int ngrp = N2 / gw, nb = N2 / ns, nj = ns / gw + (bcol != 0);
    // groups g0 .. of row r0, then (when the run crosses the end of the row) groups 0 .. of row r0 + 1; the first one
    // comes from `side` when the boundary lies inside it
    const int g0 = (kk * ns + bmod - bcol) / gw;
    const int n0 = min(nj, ngrp - g0);
    const int xn = ((r0 + 1) % MR) * R + (r0 + 1) / MR;
    const int per = (nj + parts - 1) / parts, ja = max(1, qu * per), jb = min(nj, (qu + 1) * per);
    real tot = 0, comp4[4] = {0, 0, 0, 0};
    for (int pp = 0; pp < ncomp; ++pp) {
        const int s = chan * ncomp + pp;
        const real* ps = part + (int64_t)s * ngrp * M;
        real a[4] = {0, 0, 0, 0};
        if (live) {
            if (qu == 0) a[0] = bcol != 0 ? side[((int64_t)s * nb + kk) * M + x] : ps[(int64_t)g0 * M + x];
            auto at = [&](int j) { return j < n0 ? ps[(int64_t)(g0 + j) * M + x] : ps[(int64_t)(j - n0) * M + xn]; };
            int j = ja;
            for (; j + 3 < jb; j += 4) {   // independent streams: the loads of a thread do not wait for each other
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] += at(j + q);
            }
            for (; j < jb; ++j) a[0] += at(j);
        }
        real acc = (a[0] + a[1]) + (a[2] + a[3]);
        if (parts > 1) {
            __syncthreads();
            sh[qu * xw + xl] = acc;
            __syncthreads();
            acc = 0;
            if (qu == 0)
                for (int q = 0; q < parts; ++q) acc += sh[q * xw + xl];
        }
        if (qu == 0 && live) {
            if (mode == 0) out[(o * nchan + chan) * npol + pp] = acc;
            tot += acc;
            comp4[pp & 3] = acc;
        }
    }
    if (qu == 0 && live && mode == 1) out[o * nchan + chan] = tot;
    if (qu == 0 && live && mode >= 2) {   // planes: |a|^2, |b|^2, Re conj(a) b, Im conj(a) b
        real* o4 = out + (o * nchan + chan) * 4;
        const real i = comp4[0] + comp4[1], d = comp4[0] - comp4[1], re2 = 2 * comp4[2], im2 = 2 * comp4[3];
        o4[0] = i;
        o4[1] = mode == 2 ? d : re2;
        o4[2] = mode == 2 ? re2 : im2;
        o4[3] = mode == 2 ? im2 : d;
    }
}

// ---- Bluestein (arbitrary nsample) ---------------------------------------------------------------------------
// W_N^{nk} = b[n] b[k] conj(b[k-n]) with b[n] = exp(-i pi n^2 / N), so
//   FFT_N(x)[k] = b[k] * ((x b) (*) conj(b))[k]
// and the linear convolution (*) is a circular one of any length L >= 2N-1: the power-of-two
// pipeline IFFT_L(FFT_L(a) * C) with C = FFT_L(wrapped conj(b)) in place of the chirp.  The whole
// dedispersion y = IFFT_N(FFT_N(x) H) becomes
//   a1 = x b (zero-padded)      -> conv1 = pipeline(a1)
//   a2 = conj(conv1 * H/N)      -> conv2 = pipeline(a2)     (|b| = 1 cancels the b[k] factors)
//   y  = conj(b * conv2)
// n^2 mod 2N is formed in 64-bit integers (n < 2^27), the angle in float64.
__device__ __forceinline__ cf bs_chirp(int64_t n, int64_t N, double sign) {
    const int64_t r = (n * n) % (2 * N);
    double s, c;
    sincospi((double)r / (double)N, &s, &c);
    return make_cf((real)c, (real)(sign * s));
}

// b[n] = exp(-i pi n^2/N), n < N
__global__ __launch_bounds__(256) void k_bs_table(cf* __restrict__ b, int64_t N) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        b[n] = bs_chirp(n, N, -1.0);
}

// wrapped convolution kernel conj(b)[m] for |m| < N on a ring of L points, times `scale`
__global__ __launch_bounds__(256) void k_bs_kernel(cf* __restrict__ c, int64_t N, int64_t L, real scale) {
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < L; m += (int64_t)gridDim.x * blockDim.x) {
        cf v = make_cf(0, 0);
        if (m < N) v = bs_chirp(m, N, +1.0);
        else if (m > L - N) v = bs_chirp(L - m, N, +1.0);
        c[m] = make_cf(v.x * scale, v.y * scale);
    }
}

// a[(n, s)] = n < N ? x[(n, s)] * b[n] : 0        (L rows of S series)
__global__ __launch_bounds__(256) void k_bs_pre(const cf* __restrict__ x, const cf* __restrict__ b,
                                                cf* __restrict__ a, int64_t N, int64_t L, int S) {
    const int64_t total = L * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / S;
        cf v = make_cf(0, 0);
        if (n < N) v = cmul(x[e], b[n]);
        a[e] = v;
    }
}

// a[(k, s)] = k < N ? conj(conv[(k, s)] * H[chan(s)][k]) : 0
__global__ __launch_bounds__(256) void k_bs_mid(const cf* __restrict__ conv, const cf* __restrict__ H,
                                                cf* __restrict__ a, int64_t N, int64_t L, int S, int npol) {
    const int64_t total = L * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / S;
        const int s = (int)(e - k * S);
        cf v = make_cf(0, 0);
        if (k < N) v = cconj(cmul(conv[e], H[(int64_t)(s / npol) * N + k]));
        a[e] = v;
    }
}

// out[(n - start, s)] = conj(b[n] * conv[(n, s)]),  start <= n < stop
__global__ __launch_bounds__(256) void k_bs_post(const cf* __restrict__ conv, const cf* __restrict__ b,
                                                 cf* __restrict__ out, int64_t start, int64_t stop, int S) {
    const int64_t total = (stop - start) * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = start + e / S;
        out[e] = cconj(cmul(conv[start * S + e], b[n]));
    }
}

// plain transforms through the same ring (pbh_fft_c2c for lengths beyond one tile / not 2^k):
//   forward: X[k] = b[k] * conv[k]            with a = x * b
//   inverse: x[n] = conj(b[n] * conv[n]) / N  with a = conj(X) * b
__global__ __launch_bounds__(256) void k_bs_pre_conj(const cf* __restrict__ x, const cf* __restrict__ b,
                                                     cf* __restrict__ a, int64_t N, int64_t L, int S) {
    const int64_t total = L * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / S;
        cf v = make_cf(0, 0);
        if (n < N) v = cmul(cconj(x[e]), b[n]);
        a[e] = v;
    }
}
__global__ __launch_bounds__(256) void k_bs_post_fft(const cf* __restrict__ conv, const cf* __restrict__ b,
                                                     cf* __restrict__ out, int64_t N, int S, int inverse, real scale) {
    const int64_t total = N * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const cf v = cmul(conv[e], b[e / S]);
        out[e] = inverse ? make_cf(v.x * scale, -v.y * scale) : v;
    }
}

// ---- radix-P stage of a long column transform (N1 = P * Q, Q rows per column tile) --------------------------
// When N1 exceeds 2^14/16 a column tile would be fewer than 16 columns wide (partial 128-B lines).  The
// transform over n1 = Q*a + b is then split: this elementwise pass does the P-point DFT over a for every
// (series, b, n2) -- P samples N/P apart, fully coalesced -- and the twiddle W_N1^{b c}; k_colq then runs
// Q-point transforms on the P row blocks (block c, row d holds k1 = c + P*d).  The inverse mirrors it:
// conj twiddle, inverse P-point DFT, after the blocks' inverse transforms.  In place.
template <int P, int DIR>
__global__ __launch_bounds__(256) void k_radix_p(const cf* __restrict__ src, int64_t src_plane, cf* __restrict__ dst,
                                                 int64_t dst_plane, int S, int64_t chunk, int N2, int N1, int64_t keep0,
                                                 int64_t keep1) {
    // src / dst may be the same planar array (in place) or different ones (series-major caller arrays);
    // dst is indexed by time - keep0 and only times in [keep0, keep1) are written (crop of the inverse stage)
    const int64_t total = (int64_t)S * chunk;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = e / chunk, np = e - s * chunk;
        const cf* base = src + s * src_plane + np;
        cf v[P];
#pragma unroll
        for (int a = 0; a < P; ++a) v[a] = base[(int64_t)a * chunk];
        // W_N1^{b c}, c = 1..P-1, from one float64 sincospi and a float64 product chain
        const int64_t b = np / N2;
        double sn, cs;
        sincospi(2.0 * (double)b / (double)N1, &sn, &cs);
        const double2 w1 = make_double2(cs, DIR < 0 ? -sn : sn);
        if (DIR < 0) Dft<P, -1>::run(v);
        double2 w = w1;
#pragma unroll
        for (int c = 1; c < P; ++c) {
            v[c] = cmul(v[c], make_cf((real)w.x, (real)w.y));
            w = make_double2(w.x * w1.x - w.y * w1.y, w.x * w1.y + w.y * w1.x);
        }
        if (DIR > 0) Dft<P, +1>::run(v);
        cf* obase = dst + s * dst_plane - keep0;
#pragma unroll
        for (int a = 0; a < P; ++a) {
            const int64_t t = (int64_t)a * chunk + np;
            if (t >= keep0 && t < keep1) obase[t] = v[a];
        }
    }
}

// rows of nplanes planar (N1 x N2) arrays from natural k1 order into the split order: out[r] = in[row_k1(r)]
__global__ __launch_bounds__(256) void k_row_permute(const cf* __restrict__ in, cf* __restrict__ out, int N1, int N2,
                                                     int P, int nplanes) {
    const int64_t total = (int64_t)nplanes * N1 * N2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / N2, j = e - row * N2;
        const int64_t pl = row / N1, r = row - pl * N1;
        out[e] = in[(pl * N1 + row_k1(r, P, N1)) * N2 + j];
    }
}

// ---- circular filter of a non-power-of-two dedispersion plan (pbhip.hip: rebuild_circular_filter) ------
// natural (N, nchan) <- plan-resident natural chirp [chan][k] (no rescaling)
__global__ __launch_bounds__(256) void k_cf_nat(const cf* __restrict__ chirp, cf* __restrict__ nat, int64_t N, int nchan) {
    const int64_t total = N * nchan;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = e / nchan;
        const int c = (int)(e - k * nchan);
        nat[e] = chirp[(int64_t)c * N + k];
    }
}
// ext[chan][j] = scale * h[(j - (N-1)) mod N, chan] for j < 2N-1, else 0: the N-periodic impulse response laid
// out so that a linear convolution with a block of N samples gives the circular one at outputs N-1 .. 2N-2
__global__ __launch_bounds__(256) void k_cf_extend(const cf* __restrict__ h, cf* __restrict__ ext, int64_t N, int64_t L,
                                                   int nchan, real scale) {
    const int64_t total = L * nchan;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / L);
        const int64_t j = e - (int64_t)c * L;
        cf v = make_cf(0, 0);
        if (j < 2 * N - 1) {
            const cf t = h[((j + 1) % N) * nchan + c];
            v = make_cf(t.x * scale, t.y * scale);
        }
        ext[e] = v;
    }
}

// stft / istft through the Bluestein ring (nperseg not a power of two, or beyond one tile).  The ring
// batch is B = nseg*S: column beta = g*S + s.  Time side: element (g, t, s) at g*N*S + t*S + s.
// Channelised side: (g, c, k, e) at g*N*S + (c*N + (k + N/2) % N)*E + e, s = c*E + e.
__device__ __forceinline__ int64_t seg_time_off(int64_t g, int64_t t, int s, int64_t N, int S) {
    return (g * N + t) * S + s;
}
__device__ __forceinline__ int64_t seg_chan_off(int64_t g, int64_t k, int s, int64_t N, int S, int E) {
    const int c = s / E, e = s - c * E;
    return g * N * S + ((int64_t)c * N + (k + N / 2) % N) * E + e;
}
// a[(n, beta)] = n < N ? src(n, beta) * b[n] : 0, with src = x (stft) or conj(Y) read through the shift (istft)
__global__ __launch_bounds__(256) void k_seg_pre(const cf* __restrict__ in, const cf* __restrict__ b,
                                                 cf* __restrict__ a, int64_t N, int64_t L, int S, int E,
                                                 int64_t nseg, int inverse) {
    const int64_t B = nseg * S, total = L * B;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / B, beta = idx - n * B;
        const int64_t g = beta / S;
        const int s = (int)(beta - g * S);
        cf v = make_cf(0, 0);
        if (n < N) {
            if (inverse) v = cmul(cconj(in[seg_chan_off(g, n, s, N, S, E)]), b[n]);
            else v = cmul(in[seg_time_off(g, n, s, N, S)], b[n]);
        }
        a[idx] = v;
    }
}
// stft : out(chan side)[k] = b[k] * conv[k] / N;   istft: out(time side)[t] = conj(b[t] * conv[t])
__global__ __launch_bounds__(256) void k_seg_post(const cf* __restrict__ conv, const cf* __restrict__ b,
                                                  cf* __restrict__ out, int64_t N, int S, int E, int64_t nseg,
                                                  int inverse, real scale) {
    const int64_t B = nseg * S, total = N * B;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / B, beta = idx - n * B;
        const int64_t g = beta / S;
        const int s = (int)(beta - g * S);
        const cf v = cmul(conv[idx], b[n]);
        if (inverse) out[seg_time_off(g, n, s, N, S)] = make_cf(v.x, -v.y);
        else out[seg_chan_off(g, n, s, N, S, E)] = make_cf(v.x * scale, v.y * scale);
    }
}

// ---- time_shift / freq_shift (pulsarbat/transforms/transforms.py:211-361): same FFT * H * IFFT skeleton ----
// H for time_shift: complex64(exp(-2 pi i shift_c f_k)), f_k = fftfreq(N, 1)[k] = bin/N   (transforms.py:270)
// H for freq_shift: 0/1 band mask in fftshifted order (transforms.py:350-359): with a = ft*N,
//   a < 0: zero shifted indices >= N + floor(a);  a >= 0: zero shifted indices < ceil(a);
//   bin k sits at shifted index (k + N/2) % N.
// Written in plan order like k_chirp (mode 0: phase ramp, 1: mask), times `scale`.
__global__ __launch_bounds__(256) void k_chirp_special(ChirpParams p, const double* __restrict__ arg, int mode) {
    const int64_t total = p.N * p.nchan;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < total;
         d += (int64_t)gridDim.x * blockDim.x) {
        const int chan = (int)(d / p.N);
        const int64_t e = d - (int64_t)chan * p.N;
        const int64_t r = e / p.N2, k2 = row_bin(e - r * p.N2, p.N2, p.perm_w, p.row_perm);
        const int64_t k = row_k1(r, p.P, p.N1) + (int64_t)p.N1 * k2;
        const double a = arg[chan];
        if (mode == 0) {
            const int64_t bin = (k <= (p.N - 1) / 2) ? k : k - p.N;
            const double phi = a * ((double)bin / (double)p.N);  // cycles
            const double fr = phi - rint(phi);
            double sn, cs;
            sincospi(2.0 * fr, &sn, &cs);
            p.out[d] = make_cf((real)(float)cs * p.scale, (real)(float)(-sn) * p.scale);
            if (p.phase) p.phase[phase_index(d, p.N2, p.phase16)] = (float)(-fr);
        } else if (mode == 1) {
            const int64_t i = (k + p.N / 2) % p.N;
            bool zero;
            if (a < 0) zero = i >= p.N + (int64_t)floor(a);
            else zero = i < (int64_t)ceil(a);
            p.out[d] = make_cf(zero ? (real)0 : p.scale, (real)0);
        } else {
            // mode 2: analytic-signal weights of real_to_complex (pulsarbat/utils.py:52-57):
            // h[0] = 1, h[1 : N/2] = 2, h[N/2] = 2 if N odd else 1, 0 above
            real hv = 0;
            if (k == 0) hv = 1;
            else if (k < p.N / 2) hv = 2;
            else if (k == p.N / 2) hv = (p.N & 1) ? 2 : 1;
            p.out[d] = make_cf(hv * p.scale, (real)0);
        }
    }
}

// out[n, s] = in[n, s] * exp(2 pi i ft[s] n), the phasor rounded to the data's precision (transforms.py:346)
__global__ __launch_bounds__(256) void k_mix(const cf* __restrict__ in, cf* __restrict__ out,
                                             const double* __restrict__ ft, int64_t N, int S) {
    const int64_t total = N * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / S;
        const int s = (int)(e - n * S);
        const double phi = ft[s] * (double)n;
        const double fr = phi - rint(phi);
        double sn, cs;
        sincospi(2.0 * fr, &sn, &cs);
        out[e] = cmul(in[e], make_cf((real)cs, (real)sn));
    }
}

// real_to_complex tail (pulsarbat/utils.py:59-65): z *= exp(-i pi/2 n), then z[::2]  ==>  out[m] = (-1)^m y[2m]
__global__ __launch_bounds__(256) void k_decimate2(const cf* __restrict__ y, cf* __restrict__ out, int64_t nout, int S) {
    const int64_t total = nout * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / S;
        const cf v = y[(2 * m) * S + (e - m * S)];
        out[e] = (m & 1) ? make_cf(-v.x, -v.y) : v;
    }
}

// zero-fill of the wrapped-around samples of time_shift (transforms.py:274-286): series s with shift a:
// a < 0: rows [N + floor(a), N);  a >= 0: rows [0, ceil(a)).  `data` is (N, S).
__global__ __launch_bounds__(256) void k_zero_edges(cf* __restrict__ data, const double* __restrict__ shift,
                                                    int64_t N, int S, int64_t maxrows) {
    const int64_t total = maxrows * S;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / S;
        const int s = (int)(e - j * S);
        const double a = shift[s];
        if (a < 0) {
            const int64_t cnt = -(int64_t)floor(a);
            if (j < cnt && j < N) data[(N - 1 - j) * S + s] = make_cf(0, 0);
        } else {
            const int64_t cnt = (int64_t)ceil(a);
            if (j < cnt && j < N) data[j * S + s] = make_cf(0, 0);
        }
    }
}

#ifndef PBH_F64
// ---- incoherent dedispersion (pulsarbat/transforms/dedispersion.py:136-177): per-channel integer shift ----
// out[n, c, :] = in[n + delay[c], c, :], n < nout.  A pure gather: rows of `unit` 4-byte words per
// (sample, channel) (float32 Stokes: inner words; complex64: 2*inner; float64: 2*inner; ...).
// T = uint4 when a (sample, channel) cell is a whole number of 16-byte vectors and the arrays are aligned, else
// uint32_t.  A workgroup copies 256*U consecutive elements of the output: the (sample, position in row) pair of a
// thread's first element costs one division, the following ones advance incrementally; all loads before the stores.
template <typename T, int U>
__global__ __launch_bounds__(256) void k_incoherent(const T* __restrict__ in, T* __restrict__ out,
                                                    const int64_t* __restrict__ delay, int64_t nout, int nchan,
                                                    int unit /* elements of T per (sample, channel) */) {
    const int64_t row = (int64_t)nchan * unit;  // elements per time sample
    const int64_t total = nout * row;
    const int64_t e0 = ((int64_t)blockIdx.x * U) * 256 + threadIdx.x;
    int64_t n = e0 / row;
    int64_t r = e0 - n * row;
    const int64_t dn = 256 / row, dr = 256 - dn * row;   // advance of 256 elements
    T v[U];
    int64_t eo[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int64_t e = e0 + 256 * (int64_t)j;
        eo[j] = e < total ? e : -1;
        const int64_t nn = n < nout ? n : nout - 1;   // (tail lanes re-read the last row: no branch around the load)
        v[j] = in[(nn + delay[(int)(r / unit)]) * row + r];
        n += dn;
        r += dr;
        if (r >= row) {
            r -= row;
            ++n;
        }
    }
#pragma unroll
    for (int j = 0; j < U; ++j)
        if (eo[j] >= 0) out[eo[j]] = v[j];
}
#endif  // !PBH_F64

// series-major (time fastest) form of the same gather: out[s][t] = in[s][t + dly[s / per]], every series one contiguous run
template <typename T>
__global__ __launch_bounds__(256) void k_shift_rows(const T* __restrict__ in, int64_t ipitch, T* __restrict__ out, int64_t opitch,
                                                    const int64_t* __restrict__ dly, int per, int64_t nout, int s0) {
    const int s = s0 + blockIdx.y;   // (the launch is cut into slabs of <= 65535 series: grid.y's limit)
    const T* src = in + (int64_t)s * ipitch + dly[s / per];
    T* dst = out + (int64_t)s * opitch;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nout; t += (int64_t)gridDim.x * blockDim.x) dst[t] = src[t];
}

// ---- polarisation basis change (pulsarbat/core.py:882-928) on (n, nchan, 2) data --------------------------------
// to_circular: L = (X - iY)/sqrt2, R = (X + iY)/sqrt2;   to_linear: X = (L + R)/sqrt2, Y = i(L - R)/sqrt2
__global__ __launch_bounds__(256) void k_pol_basis(const cf* __restrict__ in, cf* __restrict__ out, int64_t npairs,
                                                   int to_circular) {
    const real h = RC(0.70710678118654752);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < npairs; e += (int64_t)gridDim.x * blockDim.x) {
        const cf a = in[2 * e], b = in[2 * e + 1];
        cf p, q;
        if (to_circular) {  // a = X, b = Y:  iY = (-b.y, b.x)
            p = make_cf((a.x + b.y) * h, (a.y - b.x) * h);
            q = make_cf((a.x - b.y) * h, (a.y + b.x) * h);
        } else {            // a = L, b = R:  i(L - R) = (-(a.y - b.y), a.x - b.x)
            p = make_cf((a.x + b.x) * h, (a.y + b.y) * h);
            q = make_cf(-(a.y - b.y) * h, (a.x - b.x) * h);
        }
        out[2 * e] = p;
        out[2 * e + 1] = q;
    }
}

#ifndef PBH_F64
// ---- utils.real_to_complex as a HALF-LENGTH complex transform (pulsarbat/utils.py:38-65) -----------------------------
// A real series x of N samples, stored time fastest, IS the complex series p[m] = x[2m] + i x[2m+1] of M = N/2 samples.
// With P = FFT_M(p): X[k] = E[k] + W_N^k O[k], E = (P[k] + conj P[M-k])/2, O = -i (P[k] - conj P[M-k])/2 is the real
// transform's positive half, and the reference's recipe -- Hilbert weights, mixer exp(-i pi n/2), every second sample --
// collapses to z = IFFT_M(C) with C[k'] = X[(k' + M/2) mod M] and C[M/2] = Re P[0]: two transforms of HALF the length
// instead of two of the full length on a complex copy.

// (N, S) float32 sample-major -> planar [s][n] (the planar complex work buffer seen as floats), 64 samples x min(S, 64) series
__global__ __launch_bounds__(256) void k_real_planar(const float* __restrict__ in, float* __restrict__ out, int64_t N, int S,
                                                     int64_t plane) {
    __shared__ float t[64][65];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int s0 = blockIdx.y * 64;
    const int sw = S - s0 < 64 ? S - s0 : 64;             // series in this tile
    const int nt = N - n0 < 64 ? (int)(N - n0) : 64;      // samples in this tile
    if (sw == S) {   // all series: the tile is one contiguous run of nt * S floats
        const float* src = in + n0 * S;
        for (int i = threadIdx.x; i < nt * S; i += 256) t[i / S][i % S] = src[i];
    } else {
        for (int i = threadIdx.x; i < nt * sw; i += 256) t[i / sw][i % sw] = in[(n0 + i / sw) * S + s0 + i % sw];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < sw * 64; i += 256) {
        const int r = i >> 6, c = i & 63;                 // series r, sample c: samples fastest across lanes
        if (c < nt) out[(int64_t)(s0 + r) * plane + n0 + c] = t[c][r];
    }
}

// P, C: plan order (series s at s*M, bin k = k1 + N1 k2 at k1*N2 + k2), out of place.  One thread per bin of the row pair
// (k1, N1 - k1): it reads P[k] and P[M - k] once and writes both bins' outputs.
__global__ __launch_bounds__(256) void k_r2c_mirror(const cf* __restrict__ P, cf* __restrict__ C, int N1, int N2, real scale) {
    const int chunks = N2 / 256;
    const int pr = blockIdx.x / chunks, k2 = (blockIdx.x - pr * chunks) * 256 + threadIdx.x;
    const int A = pr, B = (N1 - pr) % N1;
    const int64_t base = (int64_t)blockIdx.y * N1 * N2;
    const int k2m = A == 0 ? (N2 - k2) % N2 : N2 - 1 - k2;          // M - k sits in row B at this column
    const cf a = P[base + (int64_t)A * N2 + k2], b = P[base + (int64_t)B * N2 + k2m];
    const int64_t k = A + (int64_t)N1 * k2;
    // W_N^k = exp(-2 pi i k / N), N = 2 N1 N2: k / N is exact in float32 (k < 2^24, N a power of two)
    const float rev = (float)k / (2.0f * (float)N1 * (float)N2);
    const cf w = make_cf(__builtin_amdgcn_cosf(rev), -__builtin_amdgcn_sinf(rev));
    const int h = N2 / 2;
    {
        const cf E = make_cf((a.x + b.x) * RC(0.5), (a.y - b.y) * RC(0.5));
        const cf O = make_cf((a.y + b.y) * RC(0.5), -(a.x - b.x) * RC(0.5));          // -i (a - conj b) / 2
        const cf X = k == 0 ? make_cf(a.x, 0) : cadd(E, cmul(w, O));                   // C[M/2] = Re P[0]
        C[base + (int64_t)A * N2 + (k2 ^ h)] = make_cf(X.x * scale, X.y * scale);
    }
    if (A != B) {   // the mirror bin M - k: the roles of a and b swap, W^(M-k) = -conj W^k
        const cf E = make_cf((b.x + a.x) * RC(0.5), (b.y - a.y) * RC(0.5));
        const cf O = make_cf((b.y + a.y) * RC(0.5), -(b.x - a.x) * RC(0.5));
        const cf X = cadd(E, cmul(make_cf(-w.x, w.y), O));
        C[base + (int64_t)B * N2 + (k2m ^ h)] = make_cf(X.x * scale, X.y * scale);
    }
}

// (the streaming yardstick k_copy lives in bench_kernels.hpp: pbhip_measure.hip and the allocation-class probes use it)
#endif  // !PBH_F64

}  // namespace PBH_NS
