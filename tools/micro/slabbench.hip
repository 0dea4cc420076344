// slabbench.hip -- upper bound of what the 256 MB Infinity Cache can give a "layout pass + column pass" pair.
// Pass A copies in -> work (the de-interleave's traffic: 2 GiB read, 2 GiB written), pass B updates work in place
// (the column pass's traffic: 2 GiB read, 2 GiB written).  Schedules, all with the same streaming kernels:
//   whole      A over everything, then B over everything (what the pipeline does: B re-reads from HBM)
//   slab(W)    for every W-MiB slab: A(slab), B(slab)  -- B finds the slab in the Infinity Cache
//   far(W)     the same launches, but B works on the slab A wrote 1 GiB earlier -- same launch count, no cache reuse:
//              the difference slab - far is the cache's contribution, far - whole the cost of the small launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, int T>
__global__ __launch_bounds__(T) void k_a(const float4* __restrict__ in, float4* __restrict__ out) {
    size_t base = (size_t)blockIdx.x * (T * U);
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * T + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + u * T + threadIdx.x] = v[u];
}
template <int U, int T>
__global__ __launch_bounds__(T) void k_b(float4* __restrict__ w) {
    size_t base = (size_t)blockIdx.x * (T * U);
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = w[base + u * T + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u].x += 1.f; w[base + u * T + threadIdx.x] = v[u]; }
}

int main() {
    const size_t bytes = 2ull << 30;
    float4 *in, *work;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&work, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(work, 0, bytes));
    constexpr int U = 4, T = 256;
    const size_t n = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& f) {
        for (int i = 0; i < 2; ++i) f();
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms per pair of passes (%6.0f GB/s over 8.59 GB)\n", name, ms / reps, 4.0 * bytes / (ms / reps) * 1e-6);
        fflush(stdout);
    };
    timeit("whole", [&] {
        hipLaunchKernelGGL((k_a<U, T>), dim3((unsigned)(n / (U * T))), dim3(T), 0, 0, in, work);
        hipLaunchKernelGGL((k_b<U, T>), dim3((unsigned)(n / (U * T))), dim3(T), 0, 0, work);
    });
    for (size_t mb : {16, 32, 64, 96, 128}) {
        const size_t sn = mb * (1ull << 20) / 16;       // float4 per slab
        const size_t nslab = n / sn;
        const unsigned blocks = (unsigned)(sn / (U * T));
        char nm[64];
        snprintf(nm, 64, "slab(%zu MiB) x%zu", mb, nslab);
        timeit(nm, [&] {
            for (size_t j = 0; j < nslab; ++j) {
                hipLaunchKernelGGL((k_a<U, T>), dim3(blocks), dim3(T), 0, 0, in + j * sn, work + j * sn);
                hipLaunchKernelGGL((k_b<U, T>), dim3(blocks), dim3(T), 0, 0, work + j * sn);
            }
        });
        snprintf(nm, 64, "far(%zu MiB) x%zu", mb, nslab);
        timeit(nm, [&] {
            for (size_t j = 0; j < nslab; ++j) {
                const size_t jb = (j + nslab / 2) % nslab;
                hipLaunchKernelGGL((k_a<U, T>), dim3(blocks), dim3(T), 0, 0, in + j * sn, work + j * sn);
                hipLaunchKernelGGL((k_b<U, T>), dim3(blocks), dim3(T), 0, 0, work + jb * sn);
            }
        });
    }
    return 0;
}
