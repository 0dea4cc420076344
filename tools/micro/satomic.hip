// Scalar atomic add (s_atomic_add, lgkmcnt-tracked) as a work-queue counter on gfx950: returns the pre-op value into an SGPR
// without touching vmcnt.  Check: 4096 workgroups x 8 waves each take one ticket; tickets must be a permutation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* ctr, unsigned* out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = v;
}
int main() {
    const int nb = 4096, nw = 8;
    unsigned *ctr, *out;
    hipMalloc(&ctr, 4); hipMalloc(&out, 4 * nb * nw);
    hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64 * nw), 0, 0, ctr, out);
    std::vector<unsigned> h(nb * nw); unsigned c = 0;
    hipMemcpy(h.data(), out, 4 * nb * nw, hipMemcpyDeviceToHost);
    hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == (unsigned)(nb * nw);
    for (int i = 0; i < nb * nw; ++i) ok &= h[i] == (unsigned)i;
    printf("counter %u (expected %d), tickets %s\n", c, nb * nw, ok ? "a permutation of 0..n-1: OK" : "WRONG");
    return ok ? 0 : 1;
}
