// colcopy.hip -- what do the COLUMN pass's memory instructions cost by themselves?  (round 3; companion of rowcopy.hip)
// A column tile is 1024 rows x 16 columns of one series of the planar work buffer: 1024 pieces of 128 bytes, `pitch` elements
// apart (2^14 elements = 128 KiB in the product).  One persistent 512-thread workgroup per CU (or two of them) walks over the
// 16384 tiles of a 2-GiB buffer in place (load, +1, store) with the thread <-> point map of the tile FFT
// (thread = column f + 16 tau, rows tau + 32 i) in these forms:
//   b64     32 loads + 32 stores of 8 bytes per lane (16 lanes per 128-byte piece, 4 pieces per wave instruction)
//   b128    16 loads + 16 stores of 16 bytes per lane: the lanes of a pair (columns 2p, 2p + 1) fetch the even / odd rows of BOTH
//           columns (8 lanes per piece, 8 pieces per wave instruction); in a transform the halves would change hands by DPP
//   *_pipe  the next tile's loads issued before this tile's stores (software pipelined, as the product does)
// pitch 16384 (the product's) and 16384 + 16 (rows no longer a power of two apart).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int N1 = 1024, NCOL = 16;

// tile t -> element offset of its first piece: series s = t / groups, column group g = t % groups
__device__ __forceinline__ size_t tile_base(int t, int groups, size_t plane, int pitch) {
    const int s = t / groups, g = t - s * groups;
    return (size_t)s * plane + (size_t)g * NCOL;
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// buffer (SRD) addressing as in the product: wave-uniform tile base in SGPRs, one per-lane offset VGPR, scalar row steps
template <int MODE>
__global__ __launch_bounds__(512) void k(float2* data, float2* dst, int ntiles, int groups, size_t plane, int pitch, unsigned* counter) {
    const int f = threadIdx.x & 15, tau = threadIdx.x >> 4;
    __shared__ int next;
    auto take = [&]() {   // atomic tile hand-out, like the product (consecutive tiles stay in a tight window)
        __syncthreads();
        if (threadIdx.x == 0) next = (int)atomicAdd(counter, 1u) + (int)gridDim.x;
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(next);
    };
    const uint32_t span = (uint32_t)((size_t)N1 * pitch * 8);
    auto rs = [&](int t) { return make_rsrc(data + tile_base(t, groups, plane, pitch), span); };
    auto ws = [&](int t) { return make_rsrc(dst + tile_base(t, groups, plane, pitch), span); };   // dst == data: in place
    const int step = 32 * pitch * 8;                  // bytes between a thread's rows (b64) / half of it per b128 row pair
    const int par = f & 1, c2 = f >> 1;
    const int v64 = (tau * pitch + f) * 8, v128 = ((tau + 32 * par) * pitch + 2 * c2) * 8;
    if (MODE == 0) {
        for (int t = blockIdx.x; t < ntiles; t = take()) {
            const rsrc_t r = rs(t), w = ws(t);
            u32x2 v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(r, v64, i * step, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i].x += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], w, v64, i * step, 0); }
        }
    } else if (MODE == 1) {
        for (int t = blockIdx.x; t < ntiles; t = take()) {
            const rsrc_t r = rs(t), w = ws(t);
            u32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, v128, i * 2 * step, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i].x += 1; __builtin_amdgcn_raw_buffer_store_b128(v[i], w, v128, i * 2 * step, 0); }
        }
    } else if (MODE == 2) {
        int t = blockIdx.x;
        if (t >= ntiles) return;
        u32x2 v[32], nx[32];
        rsrc_t r = rs(t), w = ws(t);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b64(r, v64, i * step, 0);
        while (true) {
            const int tn = take();
            const bool more = tn < ntiles;
            const rsrc_t rn = more ? rs(tn) : make_rsrc(data, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) nx[i] = __builtin_amdgcn_raw_buffer_load_b64(rn, v64, i * step, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i].x += 1; __builtin_amdgcn_raw_buffer_store_b64(v[i], w, v64, i * step, 0); }
            if (!more) break;
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = nx[i];
            r = rn; w = ws(tn);
        }
    } else {
        int t = blockIdx.x;
        if (t >= ntiles) return;
        u32x4 v[16], nx[16];
        rsrc_t r = rs(t), w = ws(t);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, v128, i * 2 * step, 0);
        while (true) {
            const int tn = take();
            const bool more = tn < ntiles;
            const rsrc_t rn = more ? rs(tn) : make_rsrc(data, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) nx[i] = __builtin_amdgcn_raw_buffer_load_b128(rn, v128, i * 2 * step, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[i].x += 1; __builtin_amdgcn_raw_buffer_store_b128(v[i], w, v128, i * 2 * step, 0); }
            if (!more) break;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = nx[i];
            r = rn; w = ws(tn);
        }
    }
}

int main() {
    const int S = 16, groups = 1024;                 // 16 series x 2^24 samples = 2 GiB; 1024 column groups per series
    const int ntiles = S * groups;
    const size_t maxplane = (size_t)N1 * (16384 + 16);
    float2* d; CK(hipMalloc(&d, S * maxplane * 8)); CK(hipMemset(d, 0, S * maxplane * 8));
    float2* d2; CK(hipMalloc(&d2, S * maxplane * 8)); CK(hipMemset(d2, 0, S * maxplane * 8));
    unsigned* ctr; CK(hipMalloc(&ctr, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"b64,  load all then store all", "b128, load all then store all", "b64,  next tile's loads before the stores", "b128, next tile's loads before the stores"};
    for (int oop = 0; oop < 2; ++oop)
    for (int pitch : {16384, 16384 + 16})
        for (int mode = 0; mode < 4; ++mode)
            for (int grid : {256}) {
                float2* dst = oop ? d2 : d;
                const size_t plane = (size_t)N1 * pitch;
                auto launch = [&] {
                    CK(hipMemsetAsync(ctr, 0, 4, 0));
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, d, dst, ntiles, groups, plane, pitch, ctr);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, d, dst, ntiles, groups, plane, pitch, ctr);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, d, dst, ntiles, groups, plane, pitch, ctr);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(512), 0, 0, d, dst, ntiles, groups, plane, pitch, ctr);
                };
                for (int i = 0; i < 3; ++i) launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= 10;
                printf("%s pitch %5d  %-44s grid %4d: %7.3f ms  %6.0f GB/s\n", oop ? "out of place" : "in place    ", pitch, names[mode], grid, ms,
                       2.0 * ntiles * N1 * NCOL * 8 / ms * 1e-6);
                fflush(stdout);
            }
    return 0;
}
