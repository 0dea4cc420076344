// rowcopy.hip -- what do the row pass's memory instructions cost by themselves?  One persistent 512-thread workgroup per CU
// walks over 128-KiB rows in place (load, +1, store), like k_rowp without the transforms, in three forms:
//   b64   32 loads + 32 stores of 8 bytes per lane (lane stride 8 B: the tile FFT's natural distribution tau + 512 i)
//   b128  16 loads + 16 stores of 16 bytes per lane (adjacent pairs 2 tau, 2 tau + 1)
//   b64 with the next row's loads issued before this row's stores (software pipelined, as the product does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float2* data, int nrows, float2* dst = nullptr) {
    const int t = threadIdx.x;
    if (MODE == 0) {
        for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
            float2* row = data + (size_t)r * 16384;
            float2 v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = row[t + 512 * i];
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i].x += 1.f; row[t + 512 * i] = v[i]; }
        }
    } else if (MODE == 1) {
        for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
            float4* row = reinterpret_cast<float4*>(data + (size_t)r * 16384);
            float4* wrow = reinterpret_cast<float4*>((dst ? dst : data) + (size_t)r * 16384);
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = row[t + 512 * i];
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i].x += 1.f; wrow[t + 512 * i] = v[i]; }
        }
    } else if (MODE == 2) {
        int r = blockIdx.x;
        if (r >= nrows) return;
        float2 v[32], nx[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = data[(size_t)r * 16384 + t + 512 * i];
        while (true) {
            const int rn = r + gridDim.x;
            const bool more = rn < nrows;
            if (more) {
#pragma unroll
                for (int i = 0; i < 32; ++i) nx[i] = data[(size_t)rn * 16384 + t + 512 * i];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i].x += 1.f; data[(size_t)r * 16384 + t + 512 * i] = v[i]; }
            if (!more) break;
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = nx[i];
            r = rn;
        }
    } else {
        int r = blockIdx.x;
        if (r >= nrows) return;
        float4 v[16], nx[16];
        float4* base = reinterpret_cast<float4*>(data);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = base[(size_t)r * 8192 + t + 512 * i];
        while (true) {
            const int rn = r + gridDim.x;
            const bool more = rn < nrows;
            if (more) {
#pragma unroll
                for (int i = 0; i < 16; ++i) nx[i] = base[(size_t)rn * 8192 + t + 512 * i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i].x += 1.f; base[(size_t)r * 8192 + t + 512 * i] = v[i]; }
            if (!more) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = nx[i];
            r = rn;
        }
    }
}

int main() {
    const int nrows = 16384;                       // 2 GiB
    float2* d; CK(hipMalloc(&d, (size_t)nrows * 16384 * 8)); CK(hipMemset(d, 0, (size_t)nrows * 16384 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"b64,  load all then store all", "b128, load all then store all", "b64,  next row's loads before the stores", "b128, next row's loads before the stores"};
    for (int mode = 0; mode < 4; ++mode)
        for (int grid : {256, 512}) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, nrows);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, nrows);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, d, nrows);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, d, nrows);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 10;
            printf("%-44s grid %4d: %7.3f ms  %6.0f GB/s\n", names[mode], grid, ms, 2.0 * nrows * 16384 * 8 / ms * 1e-6);
            fflush(stdout);
        }
    // round 3: the b128 form OUT OF PLACE (a second 2-GiB buffer receives the rows)
    float2* d2; CK(hipMalloc(&d2, (size_t)nrows * 16384 * 8)); CK(hipMemset(d2, 0, (size_t)nrows * 16384 * 8));
    for (int grid : {256, 512}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, nrows, d2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, nrows, d2);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("%-44s grid %4d: %7.3f ms  %6.0f GB/s\n", "b128, load all then store all, OUT OF PLACE", grid, ms, 2.0 * nrows * 16384 * 8 / ms * 1e-6);
    }
    return 0;
}
