// classprobe.hip -- how does one get an allocation of the OTHER class?  (bufprobe.hip: copies between large allocations run
// at one of two speeds, a property of the pair of ALLOCATIONS.)  After a few ordinary allocations (as a process has them when a
// plan is created) this tries candidates for a second work buffer in several ways and prints the copy time work -> candidate
// next to the copy inside work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 a = in[base], b = in[base + 256], c = in[base + 512], d = in[base + 768];
    out[base] = a; out[base + 256] = b; out[base + 512] = c; out[base + 768] = d;
}
static void pair(const void* a, void* b1, void* b2, size_t bytes, float* t1, float* t2) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(bytes / 16 / 1024);
    *t1 = *t2 = 1e9f;
    for (int rep = 0; rep < 7; ++rep)
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)(w ? b2 : b1));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) { float* t = w ? t2 : t1; if (ms < *t) *t = ms; }
        }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}
int main() {
    const size_t G = 1ull << 30, bytes = 2 * G;
    // a process's earlier allocations: input-sized blocks, a temporary freed again, the chirp
    void *x, *tmp, *ph, *work;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&tmp, bytes)); CK(hipFree(tmp)); CK(hipMalloc(&ph, G / 2)); CK(hipMalloc(&work, bytes));
    CK(hipMemset(x, 0, bytes)); CK(hipMemset(work, 0, bytes));
    float t, ts;
    pair(x, work, (char*)x + G, G, &t, &ts);
    printf("x -> work %.4f   x -> x+1G %.4f   ratio %.3f\n", t, ts, t / ts);
    printf("-- plain candidates, all kept alive\n");
    std::vector<void*> keep;
    for (int i = 0; i < 10; ++i) {
        void* c; CK(hipMalloc(&c, bytes)); CK(hipMemset(c, 0, bytes));
        float tx, txs;
        pair(work, c, (char*)work + G, G, &t, &ts);
        pair(x, c, (char*)x + G, G, &tx, &txs);
        printf("candidate %2d at %p: work -> cand %.4f (inside work %.4f, ratio %.3f)   x -> cand %.4f (inside x %.4f, ratio %.3f)\n", i, c, t, ts, t / ts, tx, txs, tx / txs);
        keep.push_back(c);
    }
    printf("-- candidate x candidate (ratio to the copy inside the source)\n");
    for (size_t i = 0; i < keep.size(); ++i) {
        for (size_t j = 0; j < keep.size(); ++j) {
            if (i == j) { printf("   -  "); continue; }
            pair(keep[i], keep[j], (char*)keep[i] + G, G, &t, &ts);
            printf(" %.3f", t / ts);
        }
        printf("\n");
    }
    for (void* c : keep) CK(hipFree(c));
    printf("-- candidates behind growing spacers (spacers kept): does the class follow the amount of memory already taken?\n");
    {
        std::vector<void*> sp;
        size_t total = 0;
        for (size_t gib : {4, 4, 8, 8, 8, 16, 16, 32, 32, 64}) {
            void* sps; if (hipMalloc(&sps, gib * G) != hipSuccess) { printf("spacer of %zu GiB failed\n", gib); (void)hipGetLastError(); break; }
            sp.push_back(sps); total += gib;
            void* c; CK(hipMalloc(&c, bytes)); CK(hipMemset(c, 0, bytes));
            float tw, tws, tx, txs;
            pair(work, c, (char*)work + G, G, &tw, &tws);
            pair(x, c, (char*)x + G, G, &tx, &txs);
            size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
            printf("after %3zu GiB of spacers (free %.1f GiB): cand %p  work -> cand ratio %.3f   x -> cand ratio %.3f\n", total, fr / 1073741824.0, c, tw / tws, tx / txs);
            CK(hipFree(c));
        }
        for (void* q : sp) CK(hipFree(q));
    }
    printf("-- candidates of other sizes (a 2-GiB window of each)\n");
    for (size_t gib : {3, 4, 6, 8, 16}) {
        void* c; CK(hipMalloc(&c, gib * G)); CK(hipMemset(c, 0, gib * G));
        for (size_t off = 0; off + 2 <= gib; off += (gib > 4 ? gib / 4 : 1)) {
            pair(work, (char*)c + off * G, (char*)work + G, G, &t, &ts);
            printf("size %2zu GiB at %p + %zu GiB: work -> cand %.4f (inside work %.4f, ratio %.3f)\n", gib, c, off, t, ts, t / ts);
        }
        CK(hipFree(c));
    }
    return 0;
}
