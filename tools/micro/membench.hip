// membench.hip -- streaming-copy shapes on MI355X (calibration for the hot-path kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int U>
__global__ __launch_bounds__(256) void copy_gs(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) out[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) out[i] = in[i];
}
// one contiguous chunk per block (CHUNK float4 per block), U loads in flight per thread
template <int U, int T>
__global__ __launch_bounds__(T) void copy_chunk(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t base = (size_t)blockIdx.x * (T * U);
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * T + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + u * T + threadIdx.x] = v[u];
}
template <int U, int T>
__global__ __launch_bounds__(T) void copy_chunk_nt(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    size_t base = (size_t)blockIdx.x * (T * U);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* pin = reinterpret_cast<const f4*>(in);
    f4* pout = reinterpret_cast<f4*>(out);
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&pin[base + u * T + threadIdx.x]);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &pout[base + u * T + threadIdx.x]);
}
// 8-byte-per-lane version (what the FFT kernels do today)
template <int U, int T>
__global__ __launch_bounds__(T) void copy_chunk8(const float2* __restrict__ in, float2* __restrict__ out, size_t n) {
    size_t base = (size_t)blockIdx.x * (T * U);
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + u * T + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + u * T + threadIdx.x] = v[u];
}
__global__ __launch_bounds__(256) void read_only(const float4* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_only(float4* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    printf("%-34s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
    const size_t bytes = 2ull << 30;
    const size_t n = bytes / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "copy_gs<1> grid=%d", g);
        timeit(nm, 2.0 * bytes, [&] { copy_gs<1><<<g, 256>>>(a, b, n); });
        snprintf(nm, 64, "copy_gs<4> grid=%d", g);
        timeit(nm, 2.0 * bytes, [&] { copy_gs<4><<<g, 256>>>(a, b, n); });
    }
    timeit("copy_chunk<4,256>", 2.0 * bytes, [&] { copy_chunk<4, 256><<<n / (4 * 256), 256>>>(a, b, n); });
    timeit("copy_chunk<8,256>", 2.0 * bytes, [&] { copy_chunk<8, 256><<<n / (8 * 256), 256>>>(a, b, n); });
    timeit("copy_chunk<16,256>", 2.0 * bytes, [&] { copy_chunk<16, 256><<<n / (16 * 256), 256>>>(a, b, n); });
    timeit("copy_chunk<16,512>", 2.0 * bytes, [&] { copy_chunk<16, 512><<<n / (16 * 512), 512>>>(a, b, n); });
    timeit("copy_chunk<8,1024>", 2.0 * bytes, [&] { copy_chunk<8, 1024><<<n / (8 * 1024), 1024>>>(a, b, n); });
    timeit("copy_chunk_nt<8,256>", 2.0 * bytes, [&] { copy_chunk_nt<8, 256><<<n / (8 * 256), 256>>>(a, b, n); });
    timeit("copy_chunk_nt<16,512>", 2.0 * bytes, [&] { copy_chunk_nt<16, 512><<<n / (16 * 512), 512>>>(a, b, n); });
    timeit("copy_chunk8<16,256> (8B/lane)", 2.0 * bytes, [&] { copy_chunk8<16, 256><<<2 * n / (16 * 256), 256>>>((float2*)a, (float2*)b, 2 * n); });
    timeit("copy_chunk8<32,512> (8B/lane)", 2.0 * bytes, [&] { copy_chunk8<32, 512><<<2 * n / (32 * 512), 512>>>((float2*)a, (float2*)b, 2 * n); });
    timeit("read_only grid=4096", 1.0 * bytes, [&] { read_only<<<4096, 256>>>(a, (float*)b, n); });
    timeit("write_only grid=4096", 1.0 * bytes, [&] { write_only<<<4096, 256>>>(b, n); });
    return 0;
}
