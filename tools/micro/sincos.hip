// sincos.hip -- accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) against float64, for a chirp kept as an f32 phase
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(const float* ph, float2* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2(__builtin_amdgcn_cosf(ph[i]), __builtin_amdgcn_sinf(ph[i]));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i - 2048) / 4096.0);   // exact fractions incl. 0, +-0.25, -0.5
    float* d; float2* o;
    CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&o, n * 8));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    k<<<n / 256, 256>>>(d, o, n);
    std::vector<float2> r(n);
    CK(hipMemcpy(r.data(), o, n * 8, hipMemcpyDeviceToHost));
    double maxe = 0, sum2 = 0; int arg = 0;
    for (int i = 0; i < n; ++i) {
        double a = 2 * M_PI * (double)h[i];
        double ec = r[i].x - cos(a), es = r[i].y - sin(a);
        double e = sqrt(ec * ec + es * es);
        sum2 += e * e;
        if (e > maxe) { maxe = e; arg = i; }
    }
    printf("v_cos/v_sin over [-0.5, 0.5) revolutions: max |err| %.3e at phase %.9g, rms %.3e (f32 eps = 5.96e-8)\n", maxe, h[arg], sqrt(sum2 / n));
    return 0;
}
