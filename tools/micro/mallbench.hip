// mallbench.hip -- does a working set that fits the 256 MB Infinity Cache stream faster than HBM?
// In-place pass (read + write of the same buffer, like the column / row passes) repeated back to back over
// working sets from 32 MB to 2 GB; also an out-of-place ping-pong between two buffers of half the size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U, int T>
__global__ __launch_bounds__(T) void pass(const float4* __restrict__ in, float4* __restrict__ out) {
    size_t base = (size_t)blockIdx.x * (T * U);
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * T + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u].x += 1.f; out[base + u * T + threadIdx.x] = v[u]; }
}

int main() {
    const size_t maxb = 2ull << 30;
    float4* buf;
    CK(hipMalloc(&buf, maxb));
    CK(hipMemset(buf, 0, maxb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int U = 8, T = 256;
    const size_t mbs[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048};
    printf("working set MB | in-place GB/s | ping-pong (two halves) GB/s\n");
    for (size_t mb : mbs) {
        const size_t n = mb * (1ull << 20) / 16;
        const unsigned blocks = (unsigned)(n / (U * T));
        const int reps = mb <= 256 ? 200 : 40;
        float ms[2];
        for (int mode = 0; mode < 2; ++mode) {
            const unsigned bl = mode ? blocks / 2 : blocks;
            float4 *a = buf, *b = mode ? buf + n / 2 : buf;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((pass<U, T>), dim3(bl), dim3(T), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pass<U, T>), dim3(bl), dim3(T), 0, 0, (i & 1) ? b : a, (i & 1) ? a : b);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
            ms[mode] /= reps;
        }
        printf("%5zu | %8.0f | %8.0f\n", mb, 2.0 * mb * 1.048576e6 / ms[0] / 1e6, 1.0 * mb * 1.048576e6 / ms[1] / 1e6);
        fflush(stdout);
    }
    return 0;
}
