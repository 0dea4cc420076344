// pp4bench.hip -- memory patterns of a FOUR-pass sample-major schedule (round 4), without the transforms.
//
// Today (planar5): de-interleave (N,16) -> planar, column pass in place, row pass in place, column pass, re-interleave.
// Candidate: a forward column pass that reads the caller's interleaved block directly with tiles of
// (2 polarisations of one channel) x (8 consecutive columns n2) x 1024 rows -- 16-byte pieces of the input lines, the other
// 112 bytes belong to the 7 sibling tiles (the other channels), which run at the same time on the same XCD -- and writes
// FULL lines of a pol-pair-major workspace  PP[chan][k1][n2/8][pol][n2%8]  (a 128-byte line = 8 columns x 2 pols).
// The row pass then reads its 2^14-point row of one polarisation as 64-byte pieces at a 128-byte stride (pol 0, then pol 1
// by the same workgroup) and writes ordinary planar rows out of place.
//
//   fd  : I(16 B) -> PP full lines, persistent 512-thread workgroups, 32 x 8-byte loads and stores per thread and tile
//         order 0 = one global counter, channel fastest; 1 = one queue per XCD (siblings share an L2); 2 = static stride
//   row : the row pass's traffic (rows + phase), persistent, 16 x 16-byte accesses per thread and tile
//         src 0 = planar rows, 1 = PP halves;  dst 0 = in place, 1 = out of place
//   ref : the two kernels the fd pass replaces, as plain patterns (de-interleave copy + planar column copy) for the same box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

constexpr int N1 = 1024, N2 = 16384, S = 16, NCH = 8;
constexpr uint32_t NTILE = (N2 / 8) * NCH;   // 16384 column tiles of 128 KiB

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// tile -> (channel, column group of 8)
struct Tile { uint32_t chan, g; };
template <int ORDER>
__device__ __forceinline__ Tile tile_of(uint32_t t) {
    Tile r;
    r.chan = t & 7u;
    r.g = t >> 3;
    return r;
}

// thread 0: index of a tile to work on (NTILE = none left).  ORDER 1: own XCD's queue first, then the others'.
template <int ORDER>
__device__ __forceinline__ uint32_t take(unsigned* counters, unsigned xcc, uint32_t& stat, uint32_t G) {
    if (ORDER == 0) return atomicAdd(&counters[0], 1u);
    if (ORDER == 2) { uint32_t t = stat; stat += G; return t; }
    constexpr uint32_t per = NTILE / 8;
    for (unsigned q = 0; q < 8; ++q) {
        const unsigned x = (xcc + q) & 7u;
        const uint32_t idx = atomicAdd(&counters[x * 32], 1u);   // counters 128 B apart
        if (idx < per) {
            // queue x: groups g = 8 j + x, channel fastest
            const uint32_t j = idx >> 3, ch = idx & 7u;
            return ((8 * j + x) << 3) | ch;
        }
    }
    return NTILE;
}

template <int ORDER>
__global__ __launch_bounds__(512) void k_fd(const float2* __restrict__ in, float2* __restrict__ out, unsigned* counters) {
    __shared__ unsigned slot;
    const int tid = threadIdx.x, f = tid & 15, tau = tid >> 4;
    const int c = f >> 1, pol = f & 1;
    const unsigned xcc = xcc_id();
    const uint32_t G = gridDim.x;
    const int voff_in = ((tau * N2 + c) * S + pol) * 8;
    const int voff_out = (tau * 2 * N2 + pol * 8 + c) * 8;
    constexpr int SIN = 32 * N2 * S * 8;      // 64 MiB between a thread's rows (tau + 32 i) in the input
    constexpr int SOUT = 32 * 2 * N2 * 8;     // 8 MiB in PP
    constexpr uint32_t in_span = (uint32_t)(((int64_t)(N1 - 1) * N2 * S + 7 * S + 2) * 8);
    constexpr uint32_t out_span = (uint32_t)(((int64_t)(N1 - 1) * 2 * N2 + 16) * 8);
    auto in_rsrc = [&](uint32_t t) {
        if (t >= NTILE) return make_rsrc(in, 0);
        const Tile T = tile_of<ORDER>(t);
        return make_rsrc(in + ((int64_t)T.g * 8 * S + 2 * T.chan), in_span);
    };
    auto out_rsrc = [&](uint32_t t) {
        if (t >= NTILE) return make_rsrc(out, 0);
        const Tile T = tile_of<ORDER>(t);
        return make_rsrc(out + ((int64_t)T.chan * N1 * 2 * N2 + (int64_t)T.g * 16), out_span);
    };
    uint32_t stat = blockIdx.x;
    // first two tiles
    if (tid == 0) slot = take<ORDER>(counters, xcc, stat, G);
    __syncthreads();
    uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    __syncthreads();
    if (tid == 0) slot = take<ORDER>(counters, xcc, stat, G);
    __syncthreads();
    uint32_t tn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
    __syncthreads();
    if (t >= NTILE) return;
    float2 v[32];
    {
        const rsrc_t rd = in_rsrc(t);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rd, voff_in, i * SIN, 0);
            v[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
        }
    }
    while (true) {
        unsigned fetched = NTILE;
        if (tid == 0) fetched = take<ORDER>(counters, xcc, stat, G);
        const rsrc_t rdn = in_rsrc(tn);
        const rsrc_t wr = out_rsrc(t);
        float2 nx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x2 y;
            y.x = __float_as_uint(v[i].x + 1.0f);
            y.y = __float_as_uint(v[i].y);
            __builtin_amdgcn_raw_buffer_store_b64(y, wr, voff_out, i * SOUT, 0);
            u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rdn, voff_in, i * SIN, 0);
            nx[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
        }
        if (tid == 0) slot = fetched;
        __syncthreads();
        const uint32_t tnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
        if (tn >= NTILE) break;
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
        t = tn;
        tn = tnn;
    }
}


// ---- gang-scheduled form: the 8 sibling tiles (8 channels of one column group) run AT THE SAME TIME on one XCD -----------
// Blocks b with equal b % 8 share an XCD under the observed round-robin dispatch (speed only: every tile is covered exactly
// once whatever the placement).  Gang = 8 such blocks; gang G of 32 walks the column groups G, G + 32, ...; its member m
// takes channel m.  All members do identical work, so they stay within a fraction of a tile of each other and every input
// line is requested by its 8 readers within a short window: one L2 miss + 7 hits instead of 8 trips to the fabric.
template <int MODE, int NCHG, int AUX = 0>   // NCHG: channels per gang (8 = all siblings together; 4, 2: half / quarter gangs); AUX: cache policy of the loads (1 sc0, 2 nt, 16 sc1)
__global__ __launch_bounds__(512) void k_fd_gang(const float2* __restrict__ in, float2* __restrict__ out, float* sink) {
    const int tid = threadIdx.x, f = tid & 15, tau = tid >> 4;
    const int c = f >> 1, pol = f & 1;
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;          // 32 blocks per XCD group
    constexpr uint32_t GPX = 32 / NCHG;                                 // gangs per XCD group
    const uint32_t gang = xg * GPX + li / NCHG, member = li % NCHG;
    constexpr uint32_t NGANG = 8 * GPX;
    const int voff_in = ((tau * N2 + c) * S + pol) * 8;
    const int voff_out = (tau * 2 * N2 + pol * 8 + c) * 8;
    constexpr int SIN = 32 * N2 * S * 8, SOUT = 32 * 2 * N2 * 8;
    constexpr uint32_t in_span = (uint32_t)(((int64_t)(N1 - 1) * N2 * S + 7 * S + 2) * 8);
    constexpr uint32_t out_span = (uint32_t)(((int64_t)(N1 - 1) * 2 * N2 + 16) * 8);
    // work list of this block: (group, channel) pairs; a gang with NCHG < 8 walks its groups once per channel subset
    constexpr uint32_t NG = N2 / 8;                                     // 2048 column groups
    constexpr uint32_t SUB = NCH / NCHG;                                // channel subsets
    const uint32_t nwork = (NG / NGANG) * SUB;
    auto work = [&](uint32_t k, uint32_t& g, uint32_t& ch) {
        g = (k / SUB) * NGANG + gang;
        ch = (k % SUB) * NCHG + member;
    };
    auto in_rsrc = [&](uint32_t k) {
        if (k >= nwork) return make_rsrc(in, 0);
        uint32_t g, ch; work(k, g, ch);
        return make_rsrc(in + ((int64_t)g * 8 * S + 2 * ch), in_span);
    };
    auto out_rsrc = [&](uint32_t k) {
        uint32_t g, ch; work(k, g, ch);
        return make_rsrc(out + ((int64_t)ch * N1 * 2 * N2 + (int64_t)g * 16), out_span);
    };
    float2 v[32];
    float acc = 0.f;
    {
        const rsrc_t rd = in_rsrc(0);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE != 2) { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rd, voff_in, i * SIN, AUX); v[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y)); }
            else v[i] = make_float2((float)tid, (float)i);
        }
    }
    for (uint32_t k = 0; k < nwork; ++k) {
        const rsrc_t rdn = in_rsrc(k + 1);
        const rsrc_t wr = out_rsrc(k);
        float2 nx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE != 1) {
                u32x2 y;
                y.x = __float_as_uint(v[i].x + 1.0f);
                y.y = __float_as_uint(v[i].y);
                __builtin_amdgcn_raw_buffer_store_b64(y, wr, voff_out, i * SOUT, 0);
            } else acc += v[i].x + v[i].y;
            if (MODE != 2) { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rdn, voff_in, i * SIN, AUX); nx[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y)); }
            else nx[i] = make_float2(v[i].x + 1.f, v[i].y);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
    }
    if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}

// row pass reading PP halves, gang of 2: blocks b and b + 8 take the two polarisations of the same (channel, k1) rows
template <int DST>
__global__ __launch_bounds__(512) void k_row_gang(const float2* __restrict__ src, float2* __restrict__ dst, const float* __restrict__ phase) {
    const int tau = threadIdx.x;
    constexpr int M = N2;
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;      // li 0..31
    const uint32_t pairslot = xg * 16 + (li >> 1), pol = li & 1;    // 128 pair slots
    constexpr uint32_t NPAIR = NCH * N1;
    const int voff_pl = tau * 16;
    const int voff_pp = (tau >> 2) * 128 + (tau & 3) * 16;
    constexpr int STEP_PL = (M / 16) * 8, STEP_PP = 2 * STEP_PL;
    auto src_rsrc = [&](uint32_t u) {
        if (u >= NPAIR) return make_rsrc(src, 0);
        return make_rsrc(src + (int64_t)u * 2 * M + pol * 8, (uint32_t)(2 * M * 8 - pol * 64));
    };
    auto dst_rsrc = [&](uint32_t u) {
        const uint32_t chan = u / N1, k1 = u % N1;
        return make_rsrc(dst + (((int64_t)chan * 2 + pol) * N1 + k1) * M, (uint32_t)(M * 8));
    };
    u32x4 v[16];
    uint32_t u = pairslot;
    {
        const rsrc_t r = src_rsrc(u);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_pp, j * STEP_PP, 0);
    }
    float acc = 0.f;
    for (; u < NPAIR; u += 128) {
        const rsrc_t rp = make_rsrc(phase + (int64_t)u * M, (uint32_t)(M * 4));
        u32x4 ph[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (512 * 16), 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __uint_as_float(ph[j].x);
        const rsrc_t rdn = src_rsrc(u + 128), wr = dst_rsrc(u);
        u32x4 nx[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32x4 y = v[j];
            y.x = __float_as_uint(__uint_as_float(y.x) + acc);
            __builtin_amdgcn_raw_buffer_store_b128(y, wr, voff_pl, j * STEP_PL, 0);
            nx[j] = __builtin_amdgcn_raw_buffer_load_b128(rdn, voff_pp, j * STEP_PP, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nx[j];
    }
}


// ---- "Q4": tiles of 4 series (a channel pair x 2 pols) x 4 columns: 32-byte pieces of the input, lines of 4 series x 4 columns -------
// A vector-memory instruction costs the CU's address path ~4 clocks per 128-byte line it touches whatever it uses of the line
// (fd read only: 2 GiB of 16-byte pieces in 0.92 ms = 3.8 clk per piece per CU, cache policy irrelevant), so 16-byte pieces cap a
// pass at ~2.2 TB/s and 32-byte pieces at ~4.4.  Q4 puts 32-byte pieces on BOTH the forward pass's read side and the row pass's.
// line of the workspace Q4[chanpair][k1][n2/4][128 B]: LAY 0: [c0p0][c0p1][c1p0][c1p1] x 4 columns (a channel = one 64-byte sector)
//                                                     LAY 1: [c0p0][c1p0][c0p1][c1p1]            (a polarisation = one sector)
template <int MODE, int LAY, int SCHED>   // SCHED 0: gang of 4 (static), 1: one global counter (chan pair fastest)
__global__ __launch_bounds__(512) void k_fdq(const float2* __restrict__ in, float2* __restrict__ out, unsigned* counters, float* sink) {
    __shared__ unsigned slot;
    const int tid = threadIdx.x, f = tid & 15, tau = tid >> 4;
    const int c = f >> 2, ser = f & 3, chanlo = ser >> 1, pol = ser & 1;
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;
    const uint32_t gang = xg * 8 + (li >> 2), member = li & 3;    // 64 gangs of 4
    constexpr uint32_t NG = N2 / 4;                               // 4096 column groups of 4
    constexpr uint32_t NT = NG * 4;
    const int voff_in = ((tau * N2 + c) * S + ser) * 8;
    const int lofs = LAY == 0 ? (chanlo * 2 + pol) * 4 + c : (pol * 2 + chanlo) * 4 + c;
    const int voff_out = (tau * 4 * N2 + lofs) * 8;               // a chan pair's row k1 holds N2/4 lines = 4 N2 elements
    constexpr int SIN = 32 * N2 * S * 8, SOUT = 32 * 4 * N2 * 8;
    constexpr uint32_t in_span = (uint32_t)(((int64_t)(N1 - 1) * N2 * S + 3 * S + 4) * 8);
    constexpr uint32_t out_span = (uint32_t)(((int64_t)(N1 - 1) * 4 * N2 + 16) * 8);
    auto tile = [&](uint32_t k, uint32_t& g, uint32_t& j) {      // k-th tile of this block
        if (SCHED == 0) { g = k * 64 + gang; j = member; }
        else { g = k >> 2; j = k & 3; }
    };
    auto in_rsrc = [&](uint32_t k) {
        uint32_t g, j; tile(k, g, j);
        if (g >= NG) return make_rsrc(in, 0);
        return make_rsrc(in + ((int64_t)g * 4 * S + 4 * j), in_span);
    };
    auto out_rsrc = [&](uint32_t k) {
        uint32_t g, j; tile(k, g, j);
        if (g >= NG) return make_rsrc(out, 0);
        return make_rsrc(out + ((int64_t)j * N1 * 4 * N2 + (int64_t)g * 16), out_span);
    };
    uint32_t k = 0, kn = 1;
    if (SCHED == 1) {
        if (tid == 0) slot = atomicAdd(&counters[0], 1u);
        __syncthreads();
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
        if (tid == 0) slot = atomicAdd(&counters[0], 1u);
        __syncthreads();
        kn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
    }
    float2 v[32];
    float acc = 0.f;
    {
        const rsrc_t rd = in_rsrc(k);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE != 2) { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rd, voff_in, i * SIN, 0); v[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y)); }
            else v[i] = make_float2((float)tid, (float)i);
        }
    }
    while (true) {
        { uint32_t g, j; tile(k, g, j); if (g >= NG) break; }
        unsigned fetched = 0;
        if (SCHED == 1 && tid == 0) fetched = atomicAdd(&counters[0], 1u);
        const rsrc_t rdn = in_rsrc(kn);
        const rsrc_t wr = out_rsrc(k);
        float2 nx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE != 1) {
                u32x2 y;
                y.x = __float_as_uint(v[i].x + 1.0f);
                y.y = __float_as_uint(v[i].y);
                __builtin_amdgcn_raw_buffer_store_b64(y, wr, voff_out, i * SOUT, 0);
            } else acc += v[i].x + v[i].y;
            if (MODE != 2) { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rdn, voff_in, i * SIN, 0); nx[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y)); }
            else nx[i] = make_float2(v[i].x + 1.f, v[i].y);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
        if (SCHED == 1) {
            if (tid == 0) slot = fetched;
            __syncthreads();
            k = kn;
            kn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
            __syncthreads();
        } else { k = kn; kn = kn + 1; }
    }
    if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}


// fdq with 16-byte memory instructions (round 4, after the four-pass schedule was in): the same pieces and lines as k_fdq
// (32-byte pieces of the caller's lines in, whole Q4 lines out, gang of 4), but LW / SW = bytes per lane of the loads / stores.
//   LW 16: lanes (2 l, 2 l + 1) take the two halves of a piece -- 32 pieces per wave instruction instead of 16, half the instructions
//   SW 16: a lane stores columns (c, c + 1) of one series -- 8 lines per wave instruction instead of 4
// (the real kernel would pay a lane-pair exchange for either: v_permlane32_swap, one per float)
// SUB = log2 of the columns of a sub-block: the same pieces with the rows of a tile 2^SUB x 128 B apart instead of 2 MiB (sub-blocks of
// (1024, 2^SUB, 16)): is the cost of a
// partial-line load the LINE, or the 2-MiB page each of its rows lives in?
template <int MODE, int LW, int SW, int SUB = 0, int WC = 0, int STAUX = 0>   // STAUX: cache policy bits of the stores (1 sc0, 2 nt, 16 sc1)   // WC (8-byte stores): 1 = a tile's outputs as ONE contiguous 128-KiB block, 2 = rows of a tile 1 KiB apart in 8-row chunks
__global__ __launch_bounds__(512) void k_fdqw(const float2* __restrict__ in, float2* __restrict__ out, float* sink) {
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;
    const uint32_t gang = xg * 8 + (li >> 2), member = li & 3;    // 64 gangs of 4
    constexpr uint32_t NG = N2 / 4;
    constexpr int NLD = LW == 16 ? 16 : 32, NST = SW == 16 ? 16 : 32;
    int voff_in, voff_out;
    constexpr int N2E = SUB ? (1 << SUB) : N2;
    if (LW == 16) { const int h = tid & 1, c = (tid >> 1) & 3, t = tid >> 3; voff_in = ((t * N2E + c) * S + 2 * h) * 8; }
    else          { const int ser = tid & 3, c = (tid >> 2) & 3, t = tid >> 4; voff_in = ((t * N2E + c) * S + ser) * 8; }
    if (SW == 16) { const int cp = tid & 1, ser = (tid >> 1) & 3, k = tid >> 3; voff_out = (k * 4 * N2 + ser * 4 + cp * 2) * 8; }
    else          { const int c = tid & 3, ser = (tid >> 2) & 3, k = tid >> 4; voff_out = (k * 4 * N2 + ser * 4 + c) * 8; }
    if (WC == 1) voff_out = tid * 8;
    if (WC == 2) { const int c = tid & 3, ser = (tid >> 2) & 3, k = tid >> 4; voff_out = ((k >> 3) * 4 * N2 * 8 + (k & 7) * 16 + ser * 4 + c) * 8; }   // [k1/8][n2/4][k1%8][16]
    constexpr uint32_t SIN = (uint32_t)(1024 / NLD) * N2E * S * 8, SOUT = WC == 1 ? 4096u : (uint32_t)(1024 / NST) * 4 * N2 * 8;
    constexpr uint32_t in_span = (uint32_t)(((int64_t)(N1 - 1) * N2E * S + 3 * S + 4) * 8);
    constexpr uint32_t out_span = (uint32_t)(((int64_t)(N1 - 1) * 4 * N2 + 16) * 8);
    auto in_rsrc = [&](uint32_t k) {
        const uint32_t g = k * 64 + gang;
        if (g >= NG) return make_rsrc(in, 0);
        if (SUB) return make_rsrc(in + ((int64_t)(g / (N2E / 4)) * N1 * N2E * S + (int64_t)(g % (N2E / 4)) * 4 * S + 4 * member), in_span);
        return make_rsrc(in + ((int64_t)g * 4 * S + 4 * member), in_span);
    };
    auto out_rsrc = [&](uint32_t k) {
        const uint32_t g = k * 64 + gang;
        if (g >= NG) return make_rsrc(out, 0);
        if (WC == 1) return make_rsrc(out + ((int64_t)g * 4 + member) * 16384, 131072u);
        if (WC == 2) return make_rsrc(out + ((int64_t)member * N1 * 4 * N2 + (int64_t)g * 128), out_span);
        return make_rsrc(out + ((int64_t)member * N1 * 4 * N2 + (int64_t)g * 16), out_span);
    };
    float v[64], nx[64];
    float acc = 0.f;
    auto load_all = [&](rsrc_t rd, float* d) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (LW == 16) { u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rd, voff_in, (int)(i * SIN), 0); d[4 * i] = __uint_as_float(x.x); d[4 * i + 1] = __uint_as_float(x.y); d[4 * i + 2] = __uint_as_float(x.z); d[4 * i + 3] = __uint_as_float(x.w); }
            else { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rd, voff_in, (int)(i * SIN), 0); d[2 * i] = __uint_as_float(x.x); d[2 * i + 1] = __uint_as_float(x.y); }
        }
    };
    if (MODE != 2) load_all(in_rsrc(0), v);
    else {
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = (float)(tid + i);
    }
    for (uint32_t k = 0; k * 64 + gang < NG; ++k) {
        const rsrc_t rdn = in_rsrc(k + 1);
        const rsrc_t wr = out_rsrc(k);
        // stores and loads interleaved in proportion (as the kernels pace them)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE != 1) {
                if (SW == 16) { if (i % 2 == 0) { u32x4 y; y.x = __float_as_uint(v[2 * i] + 1.f); y.y = __float_as_uint(v[2 * i + 1]); y.z = __float_as_uint(v[2 * i + 2]); y.w = __float_as_uint(v[2 * i + 3]);
                                                  __builtin_amdgcn_raw_buffer_store_b128(y, wr, voff_out, (int)((i / 2) * SOUT), 0); } }
                else { u32x2 y; y.x = __float_as_uint(v[2 * i] + 1.f); y.y = __float_as_uint(v[2 * i + 1]); __builtin_amdgcn_raw_buffer_store_b64(y, wr, voff_out, (int)(i * SOUT), STAUX); }
            } else acc += v[2 * i] + v[2 * i + 1];
            if (MODE != 2) {
                if (LW == 16) { if (i % 2 == 0) { const int j = i / 2; u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rdn, voff_in, (int)(j * SIN), 0); nx[4 * j] = __uint_as_float(x.x); nx[4 * j + 1] = __uint_as_float(x.y); nx[4 * j + 2] = __uint_as_float(x.z); nx[4 * j + 3] = __uint_as_float(x.w); } }
                else { u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rdn, voff_in, (int)(i * SIN), 0); nx[2 * i] = __uint_as_float(x.x); nx[2 * i + 1] = __uint_as_float(x.y); }
            } else { nx[2 * i] = v[2 * i] + 1.f; nx[2 * i + 1] = v[2 * i + 1]; }
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = nx[i];
    }
    if (MODE == 1 && acc == 123.456f) sink[0] = acc;
}

// row pass over Q4: a workgroup takes a channel (both pols one after the other, one phase row), 32-byte pieces of every line
template <int LAY, int SCHED>   // SCHED 0: gang of 2 static (the two channels of a pair side by side), 1: global counter, channel fastest
__global__ __launch_bounds__(512) void k_rowq(const float2* __restrict__ src, float2* __restrict__ dst, const float* __restrict__ phase, unsigned* counters) {
    __shared__ unsigned slot;
    const int tau = threadIdx.x;
    constexpr int M = N2;
    constexpr uint32_t NPAIR = NCH * N1;                           // (channel, k1) pairs
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;
    const int voff_pl = tau * 16;
    constexpr int STEP_PL = (M / 16) * 8, STEP_Q = 4 * STEP_PL;   // 1024 bins = 256 lines
    auto unit = [&](uint32_t k, uint32_t& chan, uint32_t& k1) {   // k-th unit of this block
        if (SCHED == 0) { const uint32_t slotp = xg * 16 + (li >> 1); const uint32_t u = k * 128 + slotp; chan = (u / N1) * 2 + (li & 1); k1 = u % N1; if (u >= NPAIR / 2) chan = NCH; }
        else { chan = (k & 1) + 2 * ((k >> 1) / N1); k1 = (k >> 1) % N1; if (k >= NPAIR) chan = NCH; }
    };
    auto src_rsrc = [&](uint32_t chan, uint32_t k1, int pol) {
        if (chan >= NCH) return make_rsrc(src, 0);
        const int chanlo = chan & 1;
        const int sofs = (LAY == 0 ? (chanlo * 2 + pol) : (pol * 2 + chanlo)) * 4;   // elements into the line
        return make_rsrc(src + ((int64_t)(chan >> 1) * N1 + k1) * 4 * M + sofs, (uint32_t)((4 * M - sofs) * 8));
    };
    auto dst_rsrc = [&](uint32_t chan, uint32_t k1, int pol) {
        return make_rsrc(dst + (((int64_t)chan * 2 + pol) * N1 + k1) * M, (uint32_t)(M * 8));
    };
    const int voff_q = (tau >> 1) * 128 + (tau & 1) * 16;
    uint32_t k = 0, kn = 1;
    if (SCHED == 1) {
        if (tau == 0) slot = atomicAdd(&counters[0], 1u);
        __syncthreads();
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
        if (tau == 0) slot = atomicAdd(&counters[0], 1u);
        __syncthreads();
        kn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
    }
    uint32_t chan, k1;
    unit(k, chan, k1);
    if (chan >= NCH) return;
    int pol = 0;
    u32x4 v[16];
    {
        const rsrc_t r = src_rsrc(chan, k1, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_q, j * STEP_Q, 0);
    }
    float acc = 0.f;
    while (true) {
        unsigned fetched = 0;
        if (pol == 0) {
            const rsrc_t rp = make_rsrc(phase + ((int64_t)chan * N1 + k1) * M, (uint32_t)(M * 4));
            u32x4 ph[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (512 * 16), 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __uint_as_float(ph[j].x);
            if (SCHED == 1 && tau == 0) { fetched = atomicAdd(&counters[0], 1u); slot = fetched; }
        }
        uint32_t chn = chan, k1n = k1;
        if (pol == 1) unit(kn, chn, k1n);
        const rsrc_t rdn = src_rsrc(chn, k1n, pol ^ 1), wr = dst_rsrc(chan, k1, pol);
        u32x4 nx[16];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32x4 y = v[j];
            y.x = __float_as_uint(__uint_as_float(y.x) + acc);
            __builtin_amdgcn_raw_buffer_store_b128(y, wr, voff_pl, j * STEP_PL, 0);
            nx[j] = __builtin_amdgcn_raw_buffer_load_b128(rdn, voff_q, j * STEP_Q, 0);
        }
        if (chn >= NCH) break;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nx[j];
        if (pol == 1) {
            k = kn;
            kn = SCHED == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)slot) : kn + 1;
            chan = chn; k1 = k1n;
        }
        pol ^= 1;
        __syncthreads();
    }
}


// row pass over Q4, gang of 4: the four series of a (channel pair, k1) line set side by side, ONE series per workgroup (no pol
// pairing: the two pols of a channel read the same phase row at the same time -- one of them from L2)
template <int DST>
__global__ __launch_bounds__(512) void k_rowq4(const float2* __restrict__ src, float2* __restrict__ dst, const float* __restrict__ phase) {
    const int tau = threadIdx.x;
    constexpr int M = N2;
    const uint32_t b = blockIdx.x, xg = b & 7u, li = b >> 3;
    const uint32_t gang = xg * 8 + (li >> 2), ser = li & 3;         // 64 gangs
    const int chanlo = ser >> 1, pol = ser & 1;
    constexpr uint32_t NU = (NCH / 2) * N1;                           // (channel pair, k1) units
    const int voff_pl = tau * 16;
    const int voff_q = (tau >> 1) * 128 + (tau & 1) * 16;
    constexpr int STEP_PL = (M / 16) * 8, STEP_Q = 4 * STEP_PL;
    auto src_rsrc = [&](uint32_t u) {
        if (u >= NU) return make_rsrc(src, 0);
        return make_rsrc(src + (int64_t)u * 4 * M + ser * 4, (uint32_t)((4 * M - ser * 4) * 8));
    };
    auto dst_rsrc = [&](uint32_t u) {
        const uint32_t cp = u / N1, k1 = u % N1;
        return make_rsrc(dst + ((((int64_t)cp * 2 + chanlo) * 2 + pol) * N1 + k1) * M, (uint32_t)(M * 8));
    };
    u32x4 v[16];
    uint32_t u = gang;
    {
        const rsrc_t r = src_rsrc(u);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, voff_q, j * STEP_Q, 0);
    }
    float acc = 0.f;
    for (; u < NU; u += 64) {
        const uint32_t cp = u / N1, k1 = u % N1;
        const rsrc_t rp = make_rsrc(phase + ((int64_t)(cp * 2 + chanlo) * N1 + k1) * M, (uint32_t)(M * 4));
        u32x4 ph[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (512 * 16), 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += __uint_as_float(ph[j].x);
        const rsrc_t rdn = src_rsrc(u + 64), wr = dst_rsrc(u);
        u32x4 nx[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32x4 y = v[j];
            y.x = __float_as_uint(__uint_as_float(y.x) + acc);
            __builtin_amdgcn_raw_buffer_store_b128(y, wr, voff_pl, j * STEP_PL, 0);
            nx[j] = __builtin_amdgcn_raw_buffer_load_b128(rdn, voff_q, j * STEP_Q, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nx[j];
    }
}

// ---- the pair the fd pass replaces, as bare patterns ----------------------------------------------------------------
// planar column copy in place: tiles of 16 columns x 1024 rows of one series (128-byte pieces 128 KiB apart), persistent
__global__ __launch_bounds__(512) void k_colcopy(float2* data, unsigned* counters) {
    __shared__ unsigned slot;
    const int tid = threadIdx.x, f = tid & 15, tau = tid >> 4;
    const uint32_t G = gridDim.x;
    const int voff = (tau * N2 + f) * 8;
    constexpr int STEP = 32 * N2 * 8;
    constexpr uint32_t span = (uint32_t)(((int64_t)(N1 - 1) * N2 + 16) * 8);
    auto rs = [&](uint32_t t) {
        if (t >= NTILE) return make_rsrc(data, 0);
        const uint32_t s = t / (N2 / 16), g = t % (N2 / 16);
        return make_rsrc(data + ((int64_t)s * N1 * N2 + (int64_t)g * 16), span);
    };
    uint32_t t = blockIdx.x, tn = t + G;
    float2 v[32];
    {
        const rsrc_t rd = rs(t);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rd, voff, i * STEP, 0);
            v[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
        }
    }
    while (true) {
        unsigned fetched = NTILE;
        if (tid == 0) fetched = 2 * G + atomicAdd(&counters[0], 1u);
        const rsrc_t rdn = rs(tn), wr = rs(t);
        float2 nx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            u32x2 y;
            y.x = __float_as_uint(v[i].x + 1.0f);
            y.y = __float_as_uint(v[i].y);
            __builtin_amdgcn_raw_buffer_store_b64(y, wr, voff, i * STEP, 0);
            u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rdn, voff, i * STEP, 0);
            nx[i] = make_float2(__uint_as_float(x.x), __uint_as_float(x.y));
        }
        if (tid == 0) slot = fetched;
        __syncthreads();
        const uint32_t tnn = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        __syncthreads();
        if (tn >= NTILE) break;
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = nx[i];
        t = tn;
        tn = tnn;
    }
}

// ---- row pass traffic -------------------------------------------------------------------------------------------------
// unit of work: a (channel, k1) pair = 2 polarisation rows of 2^14 points + one phase row (float32)
template <int SRC, int DST, int PIPE>
__global__ __launch_bounds__(512) void k_row(const float2* __restrict__ src, float2* __restrict__ dst, const float* __restrict__ phase,
                                             unsigned* counters) {
    __shared__ unsigned slot;
    const int tau = threadIdx.x;
    const uint32_t G = gridDim.x, NPAIR = NCH * N1;
    constexpr int M = N2;
    // SRC 0: planar rows [chan][pol][k1][M]; SRC 1: PP [chan][k1][M/8][pol][8]
    const int voff_pl = tau * 16;
    const int voff_pp = (tau >> 2) * 128 + (tau & 3) * 16;
    constexpr int STEP_PL = (M / 16) * 8, STEP_PP = 2 * STEP_PL;
    auto src_rsrc = [&](uint32_t u, int pol) {
        if (u >= NPAIR) return make_rsrc(src, 0);
        const uint32_t chan = u / N1, k1 = u % N1;
        if (SRC == 0) return make_rsrc(src + (((int64_t)chan * 2 + pol) * N1 + k1) * M, (uint32_t)(M * 8));
        return make_rsrc(src + ((int64_t)chan * N1 + k1) * 2 * M + pol * 8, (uint32_t)(2 * M * 8 - pol * 64));
    };
    auto dst_rsrc = [&](uint32_t u, int pol) {
        if (u >= NPAIR) return make_rsrc(dst, 0);
        const uint32_t chan = u / N1, k1 = u % N1;
        return make_rsrc(dst + (((int64_t)chan * 2 + pol) * N1 + k1) * M, (uint32_t)(M * 8));
    };
    auto load_tile = [&](rsrc_t r, u32x4* v) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, SRC == 0 ? voff_pl : voff_pp, j * (SRC == 0 ? STEP_PL : STEP_PP), 0);
    };
    uint32_t u = blockIdx.x;
    if (u >= NPAIR) return;
    uint32_t unx = u + G;
    int pol = 0;
    u32x4 v[16];
    load_tile(src_rsrc(u, 0), v);
    float acc = 0.f;
    while (true) {
        unsigned fetched = NPAIR;
        u32x4 ph[8];
        if (pol == 0) {
            const rsrc_t rp = make_rsrc(phase + (int64_t)u * M, (uint32_t)(M * 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) ph[j] = __builtin_amdgcn_raw_buffer_load_b128(rp, tau * 16, j * (512 * 16), 0);
            if (tau == 0) fetched = 2 * G + atomicAdd(&counters[0], 1u);
            if (tau == 0) slot = fetched;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += __uint_as_float(ph[j].x);
        }
        const bool last_pol = pol == 1;
        const uint32_t un = last_pol ? unx : u;
        const int poln = last_pol ? 0 : 1;
        const rsrc_t rdn = src_rsrc(un, poln);
        const rsrc_t wr = dst_rsrc(u, pol);
        u32x4 nx[16];
        if (PIPE == 1) load_tile(rdn, nx);        // the next tile's samples on their way before this tile's stores
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32x4 y = v[j];
            y.x = __float_as_uint(__uint_as_float(y.x) + acc);
            __builtin_amdgcn_raw_buffer_store_b128(y, wr, voff_pl, j * STEP_PL, 0);
            if (PIPE == 2) nx[j] = __builtin_amdgcn_raw_buffer_load_b128(rdn, SRC == 0 ? voff_pl : voff_pp, j * (SRC == 0 ? STEP_PL : STEP_PP), 0);
        }
        if (PIPE == 0) load_tile(rdn, nx);
        if (un >= NPAIR) break;
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = nx[j];
        if (last_pol) {
            u = unx;
            unx = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        }
        pol = poln;
        __syncthreads();
    }
}

// de-interleave as a bare pattern: (N, 16) -> planar, tiles of 256 samples x 16 series (32 KiB), 16-byte accesses through LDS
__global__ __launch_bounds__(256) void k_deint(const float4* __restrict__ in, float4* __restrict__ out) {
    __shared__ float2 lds[16][256 + 2];
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 256;
    // load: 256 samples x 128 B = 2048 float4, 8 per thread; lane -> consecutive float4 (a sample's 8 float4 are 16 series)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;              // float4 index within the tile
        const float4 x = in[n0 * 8 + e];
        const int n = e >> 3, s = (e & 7) * 2;
        lds[s][n] = make_float2(x.x, x.y);
        lds[s + 1][n] = make_float2(x.z, x.w);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = tid + 256 * k;              // series = e / 128, pair of samples = e % 128
        const int s = e >> 7, n = (e & 127) * 2;
        const float2 a = lds[s][n], b = lds[s][n + 1];
        out[((int64_t)s * N1 * N2 + n0) / 2 + (e & 127)] = make_float4(a.x, a.y, b.x, b.y);
    }
}

template <typename F>
static float timeit(const char* name, double bytes, F launch, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-64s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms * 1e-6);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms;
}

// allocation: hipMalloc, or a physical allocation (hipMemCreate) mapped at a virtual address with a chosen alignment
static void* alloc_dev(size_t bytes, int vmm_align_log2) {
    void* p = nullptr;
    if (vmm_align_log2 <= 0) { CK(hipMalloc(&p, bytes)); return p; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t sz = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, sz, &prop, 0));
    CK(hipMemAddressReserve(&p, sz, (size_t)1 << vmm_align_log2, nullptr, 0));
    CK(hipMemMap(p, sz, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(p, sz, &acc, 1));
    return p;
}

int main(int argc, char** argv) {
    int vmm = 0;
    bool quick = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--vmm") && i + 1 < argc) vmm = atoi(argv[++i]);
        if (!strcmp(argv[i], "--quick")) quick = true;
    }
    const size_t bytes = (size_t)N1 * N2 * S * 8;   // 2 GiB
    float2 *in, *a, *b; float* ph; unsigned* cnt;
    in = (float2*)alloc_dev(bytes, vmm); a = (float2*)alloc_dev(bytes, vmm); b = (float2*)alloc_dev(bytes, vmm); ph = (float*)alloc_dev(bytes / 4, vmm);
    CK(hipMalloc(&cnt, 4096));
    printf("alloc %s: in %p a %p b %p ph %p\n", vmm ? "vmm" : "hipMalloc", (void*)in, (void*)a, (void*)b, (void*)ph);
    CK(hipMemset(in, 0, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(ph, 0, bytes / 4));
    auto zero = [&] { CK(hipMemsetAsync(cnt, 0, 4096, 0)); };
    const double two = 2.0 * bytes, rowb = 2.0 * bytes + bytes / 4.0;
    if (argc > 1 && !strcmp(argv[1], "--shift-sweep")) {
        // does the per-process "mode" of a (read buffer, write buffer) pair follow their relative ADDRESS?  shift the write
        // buffer inside a larger allocation and watch the de-interleave and fdq patterns
        float2* big = (float2*)alloc_dev(bytes + (256u << 20), 0);
        CK(hipMemset(big, 0, bytes + (256u << 20)));
        const size_t shifts[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1u << 20, 2u << 20, 4u << 20, 8u << 20, 16u << 20, 32u << 20, 64u << 20, 128u << 20, 0};
        for (size_t sh : shifts) {
            float2* aa = big + sh / 8;
            char nm[96];
            snprintf(nm, 96, "shift %9zu: de-interleave", sh);
            timeit(nm, two, [&] { hipLaunchKernelGGL(k_deint, dim3(N1 * N2 / 256), dim3(256), 0, 0, (const float4*)in, (float4*)aa); });
            snprintf(nm, 96, "shift %9zu: fdq gang of 4", sh);
            timeit(nm, two, [&] { hipLaunchKernelGGL((k_fdq<0, 0, 0>), dim3(256), dim3(512), 0, 0, in, aa, cnt, (float*)cnt); });
            snprintf(nm, 96, "shift %9zu: rowq4 (a shifted -> b)", sh);
            timeit(nm, rowb + bytes / 4.0, [&] { hipLaunchKernelGGL((k_rowq4<1>), dim3(256), dim3(512), 0, 0, aa, b, ph); });
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "--widths")) {
        for (int rep = 0; rep < 2; ++rep) {
            timeit("fdq x2 loads, x2 stores (k_fdq)", two, [&] { hipLaunchKernelGGL((k_fdq<0, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
#define W(M, L, St, name, by) timeit(name, by, [&] { hipLaunchKernelGGL((k_fdqw<M, L, St>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
            W(0, 8, 8, "fdqw  8-byte loads,  8-byte stores", two)
            W(0, 16, 8, "fdqw 16-byte loads,  8-byte stores", two)
            W(0, 8, 16, "fdqw  8-byte loads, 16-byte stores", two)
            W(0, 16, 16, "fdqw 16-byte loads, 16-byte stores", two)
            W(1, 8, 8, "fdqw  8-byte loads only", 1.0 * bytes)
            W(1, 16, 8, "fdqw 16-byte loads only", 1.0 * bytes)
            W(2, 8, 8, "fdqw  8-byte stores only", 1.0 * bytes)
            W(2, 8, 16, "fdqw 16-byte stores only", 1.0 * bytes)
#define W2(WCm, M, name, by) timeit(name, by, [&] { hipLaunchKernelGGL((k_fdqw<M, 8, 8, 0, WCm>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
            W2(1, 2, "stores only, tile = one contiguous 128-KiB block", 1.0 * bytes)
            W2(2, 2, "stores only, 8-row chunks (1 KiB runs)", 1.0 * bytes)
            W2(1, 0, "loads + stores, tile out = contiguous block", two)
            W2(2, 0, "loads + stores, 8-row chunks", two)
#undef W2
#define W3(AUX, M, name, by) timeit(name, by, [&] { hipLaunchKernelGGL((k_fdqw<M, 8, 8, 0, 0, AUX>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
            W3(2, 2, "stores only, nt", 1.0 * bytes)
            W3(16, 2, "stores only, sc1", 1.0 * bytes)
            W3(17, 2, "stores only, sc0 sc1", 1.0 * bytes)
            W3(3, 2, "stores only, sc0 nt", 1.0 * bytes)
            W3(2, 0, "loads + stores, nt stores", two)
            W3(17, 0, "loads + stores, sc0 sc1 stores", two)
#undef W3
#define W1(SB, M, L, St, name, by) timeit(name, by, [&] { hipLaunchKernelGGL((k_fdqw<M, L, St, SB>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
            W1(6, 1, 8, 8, "rows   8 KiB apart: 8-byte loads only", 1.0 * bytes)
            W1(8, 1, 8, 8, "rows  32 KiB apart: 8-byte loads only", 1.0 * bytes)
            W1(10, 1, 8, 8, "rows 128 KiB apart: 8-byte loads only", 1.0 * bytes)
            W1(11, 1, 8, 8, "rows 256 KiB apart: 8-byte loads only", 1.0 * bytes)
            W1(12, 1, 8, 8, "rows 512 KiB apart: 8-byte loads only", 1.0 * bytes)
            W1(13, 1, 8, 8, "rows   1 MiB apart: 8-byte loads only", 1.0 * bytes)
            W1(14, 1, 8, 8, "rows   2 MiB apart: 8-byte loads only", 1.0 * bytes)
            W1(8, 0, 8, 8, "rows  32 KiB apart: 8-byte loads, 8-byte stores", two)
            W1(10, 0, 8, 8, "rows 128 KiB apart: 8-byte loads, 8-byte stores", two)
            W1(12, 0, 8, 8, "rows 512 KiB apart: 8-byte loads, 8-byte stores", two)
            W1(13, 0, 8, 8, "rows   1 MiB apart: 8-byte loads, 8-byte stores", two)
#undef W1
#undef W
        }
        return 0;
    }
    if (quick) {
        timeit("ref: de-interleave pattern (N,16) -> planar", two, [&] { hipLaunchKernelGGL(k_deint, dim3(N1 * N2 / 256), dim3(256), 0, 0, (const float4*)in, (float4*)a); });
        timeit("ref: planar column copy, in place, persistent", two, [&] { zero(); hipLaunchKernelGGL(k_colcopy, dim3(256), dim3(512), 0, 0, a, cnt); });
        timeit("fdq gang of 4: I(32) -> Q4 lines", two, [&] { hipLaunchKernelGGL((k_fdq<0, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("fdq gang of 4: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fdq<1, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("rowq4: gang of 4, one series per workgroup", rowb + bytes / 4.0, [&] { hipLaunchKernelGGL((k_rowq4<1>), dim3(256), dim3(512), 0, 0, a, b, ph); });
        timeit("row: planar -> in place, loads after stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 0, 0>), dim3(256), dim3(512), 0, 0, a, a, ph, cnt); });
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        timeit("ref: de-interleave pattern (N,16) -> planar", two, [&] { hipLaunchKernelGGL(k_deint, dim3(N1 * N2 / 256), dim3(256), 0, 0, (const float4*)in, (float4*)a); });
        timeit("ref: planar column copy, in place, persistent", two, [&] { zero(); hipLaunchKernelGGL(k_colcopy, dim3(256), dim3(512), 0, 0, a, cnt); });
        timeit("fd: I(16) -> PP lines, global counter (channel fastest)", two, [&] { zero(); hipLaunchKernelGGL(k_fd<0>, dim3(256), dim3(512), 0, 0, in, a, cnt); });
        timeit("fd: I(16) -> PP lines, one queue per XCD", two, [&] { zero(); hipLaunchKernelGGL(k_fd<1>, dim3(256), dim3(512), 0, 0, in, a, cnt); });
        timeit("fd: I(16) -> PP lines, static stride", two, [&] { zero(); hipLaunchKernelGGL(k_fd<2>, dim3(256), dim3(512), 0, 0, in, a, cnt); });
        timeit("fdq gang of 4: I(32) -> Q4 lines", two, [&] { hipLaunchKernelGGL((k_fdq<0, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("fdq gang of 4: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fdq<1, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("fdq gang of 4: write only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fdq<2, 0, 0>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("fdq global counter: I(32) -> Q4 lines", two, [&] { zero(); hipLaunchKernelGGL((k_fdq<0, 0, 1>), dim3(256), dim3(512), 0, 0, in, a, cnt, (float*)cnt); });
        timeit("rowq lay0 (channel = sector), global counter", rowb, [&] { zero(); hipLaunchKernelGGL((k_rowq<0, 1>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("rowq lay0 (channel = sector), gang of 2", rowb, [&] { hipLaunchKernelGGL((k_rowq<0, 0>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("rowq lay1 (pol = sector), global counter", rowb, [&] { zero(); hipLaunchKernelGGL((k_rowq<1, 1>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("rowq lay1 (pol = sector), gang of 2", rowb, [&] { hipLaunchKernelGGL((k_rowq<1, 0>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("rowq4: gang of 4, one series per workgroup", rowb + bytes / 4.0, [&] { hipLaunchKernelGGL((k_rowq4<1>), dim3(256), dim3(512), 0, 0, a, b, ph); });
        timeit("fd gang of 8 (static): I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 8>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 8>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8: write only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<2, 8>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8, sc1 loads: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 8, 16>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8, nt loads: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 8, 2>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8, sc0 sc1 loads: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 8, 17>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 4: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 4, 0>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 4, sc1 loads: read only", 1.0 * bytes, [&] { hipLaunchKernelGGL((k_fd_gang<1, 4, 16>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 8, sc1 loads: I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 8, 16>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 4, sc1 loads: I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 4, 16>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 4, nt loads: I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 4, 2>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 4: I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 4>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("fd gang of 2: I(16) -> PP lines", two, [&] { hipLaunchKernelGGL((k_fd_gang<0, 2>), dim3(256), dim3(512), 0, 0, in, a, (float*)cnt); });
        timeit("row gang of 2 (pol siblings): PP halves -> out of place", rowb, [&] { hipLaunchKernelGGL((k_row_gang<1>), dim3(256), dim3(512), 0, 0, a, b, ph); });
        timeit("row: planar -> in place, loads after stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 0, 0>), dim3(256), dim3(512), 0, 0, a, a, ph, cnt); });
        timeit("row: planar -> in place, loads before stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 0, 1>), dim3(256), dim3(512), 0, 0, a, a, ph, cnt); });
        timeit("row: planar -> in place, interleaved", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 0, 2>), dim3(256), dim3(512), 0, 0, a, a, ph, cnt); });
        timeit("row: planar -> out of place, loads after stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 1, 0>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("row: planar -> out of place, loads before stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 1, 1>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("row: planar -> out of place, interleaved", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<0, 1, 2>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("row: PP halves -> out of place, loads after stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<1, 1, 0>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("row: PP halves -> out of place, loads before stores", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<1, 1, 1>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
        timeit("row: PP halves -> out of place, interleaved", rowb, [&] { zero(); hipLaunchKernelGGL((k_row<1, 1, 2>), dim3(256), dim3(512), 0, 0, a, b, ph, cnt); });
    }
    return 0;
}
