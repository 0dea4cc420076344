// bufprobe.hip -- is "fast" / "slow" a property of a BUFFER (where its pages landed) rather than of a kernel or a process?
// Round 3 saw the layout passes move 0.76 <-> 0.83 ms between processes of one box; round 4 saw the same two modes in bare
// copy patterns, unchanged by shifting one buffer against the other (pp4bench --shift-sweep).  Here: several 2-GiB
// allocations in one process, each read and written alone, then copied pairwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float* sink, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 a = in[base], b = in[base + 256], c = in[base + 512], d = in[base + 768];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ out, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    out[base] = v; out[base + 256] = v; out[base + 512] = v; out[base + 768] = v;
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const float4 a = in[base], b = in[base + 256], c = in[base + 512], d = in[base + 768];
    out[base] = a; out[base + 256] = b; out[base + 512] = c; out[base + 768] = d;
}
template <typename F>
static float timeit(F launch, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = 2ull << 30, n = bytes / 16;
    const unsigned grid = (unsigned)(n / 1024);
    std::vector<float4*> buf(nb);
    float* sink; CK(hipMalloc(&sink, 64));
    for (int i = 0; i < nb; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 0, bytes)); }
    for (int i = 0; i < nb; ++i) {
        const float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf[i], sink, n); });
        const float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf[i], n); });
        printf("buffer %d at %p: read %.4f ms (%.0f GB/s)  write %.4f ms (%.0f GB/s)\n", i, (void*)buf[i], r, bytes / r * 1e-6, w, bytes / w * 1e-6);
    }
    if (argc > 2) {   // big-arena mode: does the pair's mode follow an offset of 256 MiB ... 2 GiB inside one 4-GiB allocation?
        float4* big; CK(hipMalloc(&big, 2 * bytes)); CK(hipMemset(big, 0, 2 * bytes));
        printf("4-GiB allocation at %p\n", (void*)big);
        for (int i = 0; i < nb; ++i) {
            printf("src buffer %d -> big + s:", i);
            for (size_t sh = 0; sh <= 2048; sh += 256) {
                float4* d = big + (sh << 20) / 16;
                const float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, buf[i], d, n); }, 6);
                printf(" %4zuM %.4f", sh, c);
            }
            printf("\n");
        }
        printf("big + 0 -> big + s (inside one allocation):");
        for (size_t sh = 2048; sh >= 2048; sh -= 256) {
            const float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, big, big + (sh << 20) / 16, n); }, 6);
            printf(" %4zuM %.4f", sh, c);
            break;
        }
        printf("\n");
    }
    printf("copy ms, row = source, column = destination\n");
    for (int i = 0; i < nb; ++i) {
        for (int j = 0; j < nb; ++j) {
            if (i == j) { printf("   -   "); continue; }
            const float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, buf[i], buf[j], n); }, 6);
            printf(" %.4f", c);
        }
        printf("\n");
    }
    return 0;
}
