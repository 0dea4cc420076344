// gatherbench.hip -- address-pattern study for the column passes: what costs time, piece size, stride or direction?
// A tile is 128 KiB (1024 rows x 128 B) moved by 512 threads, like k_col.  Patterns, each with a piece size pB:
//   I(pB)  interleaved (n, 16 series) block, slow-index-first split: rows 2 MiB apart; per row 128/pB adjacent
//          lines, pB bytes of each (the other bytes belong to sibling tiles dealt to the same XCD)
//   F(pB)  interleaved block, fast-index-first split: 1024 adjacent lines, 128/pB groups 128 KiB apart
//   P(pB)  planar [16 series][1024 rows][128 KiB] workspace: 128/pB series, rows 128 KiB apart, pB contiguous bytes
// T = float2 (8 B per lane) or float4 (16 B per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Pat { int kind; int lp; };  // kind 0 = I, 1 = F, 2 = P; lp = log2(elements per piece)

template <int LLE>  // log2(elements per 128-B line): 4 for float2, 3 for float4
__host__ __device__ __forceinline__ size_t elem_addr(Pat p, unsigned tile, unsigned i) {
    const unsigned lq = LLE - p.lp;                 // log2(pieces per row-group)
    const unsigned nsub = 1u << lq;                 // sibling tiles sharing lines (I/F); series per tile (P)
    const unsigned e = i & ((1u << p.lp) - 1);
    if (p.kind == 2) {
        const unsigned lgrp = 4 - lq;               // log2(series groups)
        const unsigned sub = tile & ((1u << lgrp) - 1), g = tile >> lgrp;
        const unsigned r = (i >> p.lp) & 1023u, j = i >> (p.lp + 10);   // j < 128/pB series
        const unsigned ser = (sub << lq) + j;
        return (((((size_t)ser << 10) + r) << (10 + LLE)) + ((size_t)g << p.lp)) + e;
    }
    const unsigned sub = tile & (nsub - 1), g = tile >> lq;
    if (p.kind == 0) {
        const unsigned q = (i >> p.lp) & (nsub - 1), r = i >> LLE;
        return ((((size_t)r << 14) + ((size_t)g << lq) + q) << LLE) + ((size_t)sub << p.lp) + e;
    } else {
        const unsigned r = (i >> p.lp) & 1023u, q = i >> (p.lp + 10);
        return (((size_t)r + 1024 * (((size_t)g << lq) + q)) << LLE) + ((size_t)sub << p.lp) + e;
    }
}

// host check: every pattern must be a bijection of the 2-GiB buffer's elements
template <int LLE>
static bool check_pattern(Pat p) {
    const size_t n = (size_t)1 << (24 + LLE);
    std::vector<bool> seen(n, false);
    for (unsigned t = 0; t < (1u << 14); ++t)
        for (unsigned i = 0; i < (1u << (10 + LLE)); ++i) {
            const size_t ad = elem_addr<LLE>(p, t, i);
            if (ad >= n || seen[ad]) { printf("pattern kind %d lp %d LLE %d BAD at tile %u i %u -> %zu\n", p.kind, p.lp, LLE, t, i, ad); return false; }
            seen[ad] = true;
        }
    return true;
}

template <typename T, int LLE, int MODE>  // MODE 0 copy, 1 read only, 2 write only
__global__ __launch_bounds__(512) void k_tile(const T* __restrict__ in, T* __restrict__ out, Pat pr, Pat pw, unsigned ntile) {
    constexpr int U = (1 << (10 + LLE)) / 512;
    const unsigned b = blockIdx.x;
    const unsigned tile = (b & 7u) * (ntile >> 3) + (b >> 3);
    T v[U];
    if (MODE != 2) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[elem_addr<LLE>(pr, tile, threadIdx.x + 512 * u)];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) { T t{}; t.x = (float)threadIdx.x; t.y = (float)u; v[u] = t; }
    }
    if (MODE != 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) out[elem_addr<LLE>(pw, tile, threadIdx.x + 512 * u)] = v[u];
    } else {
        float acc = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
        if (acc == 123.456f) { T t{}; t.x = acc; out[0] = t; }
    }
}

static void timeit(const char* name, double bytes, std::function<void()> f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms * 1e-6);
    fflush(stdout);
}

static const char* kn[3] = {"I", "F", "P"};
static void* a; static void* b;
static const size_t bytes = 2ull << 30;
static const unsigned ntile = 1u << 14;

template <typename T, int LLE>
static void sweep(const char* tn) {
    const int es = (int)sizeof(T);
    auto nm = [&](char* s, const char* what, Pat p, Pat q, bool two) {
        if (two) snprintf(s, 96, "%s %s  %s(%d) -> %s(%d)", tn, what, kn[p.kind], es << p.lp, kn[q.kind], es << q.lp);
        else snprintf(s, 96, "%s %s %s(%d)", tn, what, kn[p.kind], es << p.lp);
    };
    char s[96];
    for (int kind = 0; kind < 3; ++kind)
        for (int lp = 0; lp <= LLE; ++lp) {
            Pat p{kind, lp};
            nm(s, "read ", p, p, false);
            timeit(s, 1.0 * bytes, [&] { k_tile<T, LLE, 1><<<ntile, 512>>>((const T*)a, (T*)b, p, p, ntile); });
            nm(s, "write", p, p, false);
            timeit(s, 1.0 * bytes, [&] { k_tile<T, LLE, 2><<<ntile, 512>>>((const T*)a, (T*)b, p, p, ntile); });
        }
    // transposing pairs: pass 1 reads I/F(pB) and writes P(1024/pB); pass 3 the reverse
    for (int lp = 0; lp <= LLE; ++lp) {
        const int lq = LLE - lp + (LLE == 3 ? 1 : 0) - (LLE == 3 ? 1 : 0);
        for (int kind = 0; kind < 2; ++kind) {
            Pat pi{kind, lp}, pp{2, LLE - lp};
            (void)lq;
            nm(s, "copy ", pi, pp, true);
            timeit(s, 2.0 * bytes, [&] { k_tile<T, LLE, 0><<<ntile, 512>>>((const T*)a, (T*)b, pi, pp, ntile); });
            nm(s, "copy ", pp, pi, true);
            timeit(s, 2.0 * bytes, [&] { k_tile<T, LLE, 0><<<ntile, 512>>>((const T*)a, (T*)b, pp, pi, ntile); });
        }
    }
    for (int kind = 0; kind < 3; ++kind) {
        Pat p{kind, LLE};
        nm(s, "copy ", p, p, true);
        timeit(s, 2.0 * bytes, [&] { k_tile<T, LLE, 0><<<ntile, 512>>>((const T*)a, (T*)b, p, p, ntile); });
    }
}

int main(int argc, char** argv) {
    bool ok = true;
    for (int kind = 0; kind < 3; ++kind) {
        for (int lp = 0; lp <= 4; ++lp) ok = ok && check_pattern<4>(Pat{kind, lp});
        for (int lp = 0; lp <= 3; ++lp) ok = ok && check_pattern<3>(Pat{kind, lp});
    }
    if (!ok) return 1;
    if (argc > 1 && !strcmp(argv[1], "--check")) { printf("patterns ok\n"); return 0; }
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    sweep<float2, 4>("b64 ");
    sweep<float4, 3>("b128");
    return 0;
}
