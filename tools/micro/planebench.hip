// planebench.hip -- does the distance between the 16 planes of the planar work buffer matter to the de-interleave pattern?
// (round 3)  A workgroup reads one contiguous 32-KiB piece of the sample-major block (256 samples x 16 series x 8 B) and writes
// sixteen 2-KiB pieces, one per plane, at the same offset inside every plane -- k_deinterleave_p2<16>'s memory pattern without
// its LDS transpose (the values written are whatever the thread loaded: only addresses matter here).  Plane pitch: N elements
// (128 MiB, the product's) or N + pad.  Also the reverse direction (k_reinterleave_p2's pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 256 threads; tile = 256 samples.  in: float4 index = tile * 2048 + j * 256 + tid (j < 8): 32 KiB contiguous.
// out: plane s (s < 16), 2 KiB = 128 float4 per plane and tile: thread tid writes planes (tid >> 7) + 2 j', float4 (tid & 127)
template <bool REV>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ in, float4* __restrict__ out, size_t plane4, int ntile) {
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        float4 v[8];
        if (!REV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = in[(size_t)t * 2048 + j * 256 + tid];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[(size_t)((tid >> 7) + 2 * j) * plane4 + (size_t)t * 128 + (tid & 127)] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = in[(size_t)((tid >> 7) + 2 * j) * plane4 + (size_t)t * 128 + (tid & 127)];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[(size_t)t * 2048 + j * 256 + tid] = v[j];
        }
    }
}

int main() {
    const size_t N = 1ull << 24;                 // samples per series
    const int ntile = (int)(N / 256);
    const size_t maxpad = 1 << 16;               // elements
    float4 *a, *b;
    CK(hipMalloc(&a, 16 * (N + maxpad) * 8));
    CK(hipMalloc(&b, 16 * (N + maxpad) * 8));
    CK(hipMemset(a, 0, 16 * (N + maxpad) * 8));
    CK(hipMemset(b, 0, 16 * (N + maxpad) * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t pads[] = {0, 32, 256, 512, 4096 + 32, 16384 + 256, 65536 - 256};   // elements of 8 bytes
    for (int rev = 0; rev < 2; ++rev)
        for (size_t pad : pads)
            for (int grid : {2048, 65536}) {
                const size_t plane4 = (N + pad) / 2;   // float4 units
                auto launch = [&] {
                    if (rev) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, a, b, plane4, ntile);
                    else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, a, b, plane4, ntile);
                };
                for (int i = 0; i < 3; ++i) launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= 10;
                printf("%s plane pitch N + %6zu elements, grid %5d: %7.3f ms  %6.0f GB/s\n", rev ? "planar -> sample-major" : "sample-major -> planar",
                       pad, grid, ms, 2.0 * 16 * N * 8 / ms * 1e-6);
                fflush(stdout);
            }
    return 0;
}
