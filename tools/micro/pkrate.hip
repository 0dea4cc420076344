// pkrate.hip -- issue rate of packed-f32 VALU ops vs scalar ones on gfx950 (is v_pk_* worth it for butterflies?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    v2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (v2){(float)threadIdx.x + i, 1.0f + i};
    v2 w = {0.999f, 0.001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {  // scalar fma x2
                asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3" : "+v"(a[i].x), "+v"(a[(i + 1) & 15].y) : "v"(w.x), "v"(w.y), "v"(w.y), "v"(w.x));
            } else if (MODE == 1) {  // pk fma
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(w));
            } else if (MODE == 2) {  // pk add
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
            } else if (MODE == 3) {  // scalar add x2
                asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %2, %3, %2" : "+v"(a[i].x), "+v"(a[i].y) : "v"(a[(i + 5) & 15].x), "v"(a[(i + 5) & 15].y));
            } else if (MODE == 4) {  // pk fma with op_sel / neg (complex multiply second half)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(w));
            } else if (MODE == 5) {  // pk mul
                asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(w));
            } else if (MODE == 6) {  // pk add with swap/neg
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
            }
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, int per_iter_instr, int flop_per_instr) {
    float* d; CK(hipMalloc(&d, 64));
    const int iters = 4096, grid = 256 * 4;   // 4 waves... 512 thr = 8 waves per WG; 1024 WGs
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<grid, 512>>>(d, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    k<MODE><<<grid, 512>>>(d, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: grid*8 waves / (256 CU * 4 SIMD) * iters * per_iter_instr
    double winstr = (double)grid * 8 / 1024.0 * iters * per_iter_instr;
    double cyc = ms * 1e-3 * 2.4e9 / winstr;
    double tflops = (double)grid * 512 * iters * per_iter_instr * flop_per_instr / (ms * 1e-3) * 1e-12;
    printf("%-36s %8.3f ms  %5.2f clk/wave-instr (at 2.4 GHz)  %7.1f TFLOP/s\n", name, ms, cyc, tflops);
}

int main() {
    run<0>("v_fma_f32 (x2 per item)", 32, 2);
    run<1>("v_pk_fma_f32", 16, 4);
    run<4>("v_pk_fma_f32 op_sel+neg", 16, 4);
    run<5>("v_pk_mul_f32 op_sel_hi", 16, 2);
    run<3>("v_add_f32 (x2 per item)", 32, 1);
    run<2>("v_pk_add_f32", 16, 2);
    run<6>("v_pk_add_f32 op_sel+neg", 16, 2);
    return 0;
}
