// classprobe2.hip -- round 4, second look at the allocation "classes" (DESIGN 6d d): full-size (2 GiB) copies work -> candidate,
// candidates obtained in ways classprobe.hip did not try: behind ODD spacers (its spacers were multiples of 4 GiB), with odd sizes,
// as windows of one large allocation, from the stream-ordered pool, from hipMemCreate + hipMemMap.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 a = in[base], b = in[base + 256], c = in[base + 512], d = in[base + 768];
    out[base] = a; out[base + 256] = b; out[base + 512] = c; out[base + 768] = d;
}
static const size_t G = 1ull << 30, BYTES = 2 * G;
static float copy_ms(const void* a, void* b) {   // median of 7 full-size copies
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(BYTES / 16 / 1024);
    std::vector<float> t;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) t.push_back(ms);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
static void report(const char* what, const void* work, const void* x, void* c) {
    CK(hipMemset(c, 0, BYTES));
    const float tw = copy_ms(work, c), tx = copy_ms(x, c), tb = copy_ms(c, const_cast<void*>(work));
    printf("%-46s %p  work->c %.4f  x->c %.4f  c->work %.4f\n", what, c, tw, tx, tb);
    fflush(stdout);
}
int main(int argc, char** argv) {
    void *x, *tmp, *ph, *work;
    CK(hipMalloc(&x, BYTES)); CK(hipMalloc(&tmp, BYTES)); CK(hipFree(tmp)); CK(hipMalloc(&ph, G / 2)); CK(hipMalloc(&work, BYTES));
    CK(hipMemset(x, 0, BYTES)); CK(hipMemset(work, 0, BYTES));
    printf("x %p work %p   x->work %.4f  work->x %.4f ms per 2 GiB copy\n", x, work, copy_ms(x, work), copy_ms(work, x));
    if (argc > 1 && atoi(argv[1]) > 0) {   // classprobe2 N: N plain 2-GiB blocks in a row, all kept: is the class periodic in the memory taken?
        const int n = atoi(argv[1]);
        std::vector<void*> all;
        for (int i = 0; i < n; ++i) {
            void* c; if (hipMalloc(&c, BYTES) != hipSuccess) { printf("out of memory after %d blocks\n", i); break; }
            CK(hipMemset(c, 0, BYTES));
            const float tw = copy_ms(work, c), tx = copy_ms(x, c);
            printf("block %3d (%3d GiB taken) %p  work->c %.4f  x->c %.4f  %s\n", i, 2 * i, c, tw, tx, tw < tx ? "W" : "x");
            fflush(stdout);
            all.push_back(c);
        }
        for (void* c : all) CK(hipFree(c));
        return 0;
    }
    printf("-- six plain candidates, kept (what the library does)\n");
    std::vector<void*> keep;
    for (int i = 0; i < 6; ++i) { void* c; CK(hipMalloc(&c, BYTES)); char nm[64]; snprintf(nm, 64, "plain %d", i); report(nm, work, x, c); keep.push_back(c); }
    for (void* c : keep) CK(hipFree(c));
    keep.clear();
    printf("-- a candidate behind an odd spacer (spacer freed again after the candidate exists)\n");
    const size_t spacers[] = {2ull << 20, 64ull << 20, 256ull << 20, 512ull << 20, G, G + G / 2, 3 * G, 5 * G, 7 * G + (256ull << 20)};
    for (size_t sp : spacers) {
        void *s, *c;
        CK(hipMalloc(&s, sp)); CK(hipMalloc(&c, BYTES));
        char nm[64]; snprintf(nm, 64, "behind a %.3f-GiB spacer", (double)sp / G); report(nm, work, x, c);
        CK(hipFree(c)); CK(hipFree(s));
    }
    printf("-- the same with the spacers KEPT (cumulative)\n");
    std::vector<void*> sps;
    for (size_t sp : spacers) {
        void *s, *c;
        CK(hipMalloc(&s, sp)); sps.push_back(s); CK(hipMalloc(&c, BYTES));
        char nm[64]; snprintf(nm, 64, "after + %.3f GiB kept", (double)sp / G); report(nm, work, x, c);
        CK(hipFree(c));
    }
    for (void* s : sps) CK(hipFree(s));
    printf("-- windows of one 8-GiB allocation\n");
    {
        void* big; CK(hipMalloc(&big, 4 * BYTES));
        for (int w = 0; w < 4; ++w) { char nm[64]; snprintf(nm, 64, "window %d of 8 GiB", w); report(nm, work, x, (char*)big + (size_t)w * BYTES); }
        report("window at +1 GiB", work, x, (char*)big + G);
        CK(hipFree(big));
    }
    printf("-- odd sizes\n");
    for (size_t extra : {(size_t)(2ull << 20), (size_t)(G / 2), (size_t)G}) {
        void* c; CK(hipMalloc(&c, BYTES + extra));
        char nm[64]; snprintf(nm, 64, "size 2 GiB + %.3f GiB", (double)extra / G); report(nm, work, x, c);
        CK(hipFree(c));
    }
    printf("-- stream-ordered pool (hipMallocAsync)\n");
    {
        void* c = nullptr;
        if (hipMallocAsync(&c, BYTES, 0) == hipSuccess) { CK(hipStreamSynchronize(0)); report("hipMallocAsync", work, x, c); CK(hipFreeAsync(c, 0)); CK(hipStreamSynchronize(0)); }
        else printf("hipMallocAsync failed\n");
    }
    printf("-- hipMemCreate + hipMemMap (virtual alignment 2 MiB / 1 GiB / 4 GiB)\n");
    for (int al : {21, 30, 32}) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        hipMemGenericAllocationHandle_t h; void* p = nullptr;
        if (hipMemCreate(&h, BYTES, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
        CK(hipMemAddressReserve(&p, BYTES, (size_t)1 << al, nullptr, 0));
        CK(hipMemMap(p, BYTES, 0, h, 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(p, BYTES, &acc, 1));
        char nm[64]; snprintf(nm, 64, "vmm, va aligned 2^%d", al); report(nm, work, x, p);
        CK(hipMemUnmap(p, BYTES)); CK(hipMemAddressFree(p, BYTES)); CK(hipMemRelease(h));
    }
    printf("-- six plain candidates again\n");
    for (int i = 0; i < 6; ++i) { void* c; CK(hipMalloc(&c, BYTES)); char nm[64]; snprintf(nm, 64, "plain %d", i); report(nm, work, x, c); keep.push_back(c); }
    for (void* c : keep) CK(hipFree(c));
    return 0;
}
