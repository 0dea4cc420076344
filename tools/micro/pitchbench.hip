// pitchbench.hip -- does the power-of-two row pitch of the planar work buffer cost the column passes bandwidth?
// A column tile is 1024 rows x 128 bytes of one series; rows are `pitch` bytes apart (128 KiB in the pipeline: every row of
// a tile differs from the previous one only in address bits >= 17).  In-place copy (read tile, write it back) of a 16-series
// x 1024-row x 128-KiB block with tiles handed out in order, for several pitches: 128 KiB (product), + 128 B, + 256 B, ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 512 threads move a tile: thread = (row group, 16-byte piece): 8 pieces per 128-byte row piece, 64 rows per pass
__global__ __launch_bounds__(512) void k_tile(float4* buf, size_t plane16, size_t pitch16, int ngrp, int inplace, float4* out) {
    const unsigned t = blockIdx.x;
    const unsigned s = t / ngrp, g = t % ngrp;
    float4* base = buf + s * plane16 + (size_t)g * 8;
    float4* obase = (inplace ? buf : out) + s * plane16 + (size_t)g * 8;
    const int piece = threadIdx.x & 7, r0 = threadIdx.x >> 3;   // 64 rows per pass
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = base[(size_t)(r0 + 64 * i) * pitch16 + piece];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i].x += 1.f; obase[(size_t)(r0 + 64 * i) * pitch16 + piece] = v[i]; }
}

int main() {
    const int S = 16, rows = 1024, ngrp = 1024;           // 1024 column groups of 128 B = 128 KiB per row
    const size_t row_bytes = 128 << 10;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t pads[] = {0, 128, 256, 512, 1024, 2048, 4096, 8192, 128 + 4096, 65536};
    size_t maxbytes = (size_t)S * rows * (row_bytes + 65536) + (1 << 20);
    float4 *a, *b;
    CK(hipMalloc(&a, maxbytes)); CK(hipMalloc(&b, maxbytes));
    CK(hipMemset(a, 0, maxbytes)); CK(hipMemset(b, 0, maxbytes));
    for (int inplace = 1; inplace >= 0; --inplace)
        for (size_t pad : pads) {
            const size_t pitch = row_bytes + pad, plane = pitch * rows;
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_tile, dim3(S * ngrp), dim3(512), 0, 0, a, plane / 16, pitch / 16, ngrp, inplace, b);
            CK(hipDeviceSynchronize());
            const int reps = 10;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_tile, dim3(S * ngrp), dim3(512), 0, 0, a, plane / 16, pitch / 16, ngrp, inplace, b);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            printf("%s  row pitch 128 KiB + %6zu B: %7.3f ms  %6.0f GB/s\n", inplace ? "in place    " : "out of place", pad, ms,
                   2.0 * S * rows * row_bytes / ms * 1e-6);
            fflush(stdout);
        }
    return 0;
}
