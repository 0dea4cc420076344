// pbhip_measure.hip -- unit 3 of 3: per-kernel profile of a plan and the streaming yardstick (pbhip_internal.hpp lists the units).
#include "pbhip_internal.hpp"

#include "bench_kernels.hpp"

namespace PBH_NS {
extern "C" {

// ---- measurement ----------------------------------------------------------------------------------------------------
int pbh_plan_profile(pbh_plan* p, const void* in_dev, void* out_dev, int iters, float* ms_per_kernel, int* nkernel,
                     const char** names) {
    if (!p || !in_dev || !out_dev || !ms_per_kernel || !nkernel) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp");
    if (iters <= 0) return fail(PBH_ERR_INVALID, "iters must be positive");
    HIPCHECK(hipSetDevice(p->device));
    auto steps = build_steps(p, (const cf*)in_dev, (cf*)out_dev);
    const int nk = (int)steps.size();
    // steps that share a name (the depth-first schedule launches each middle pass once per series) report as one entry
    std::vector<const char*> uniq;
    std::vector<int> slot(nk);
    for (int k = 0; k < nk; ++k) {
        int j = 0;
        while (j < (int)uniq.size() && strcmp(uniq[j], steps[k].name) != 0) ++j;
        if (j == (int)uniq.size()) uniq.push_back(steps[k].name);
        slot[k] = j;
    }
    if ((int)uniq.size() > PBH_MAX_KERNELS) return fail(PBH_ERR_INVALID, "too many kernels");
    std::vector<hipEvent_t> ev(nk + 1);
    for (auto& e : ev) HIPCHECK(hipEventCreate(&e));
    std::vector<double> acc(nk, 0.0);
    int rc = PBH_OK;
    // every event call is checked: these durations are what bench.py's `roofline` is computed from
    auto evok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == PBH_OK) rc = fail(PBH_ERR_HIP, std::string("profile: ") + what + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (int it = 0; it < iters && rc == PBH_OK; ++it) {
        evok(hipEventRecord(ev[0], p->stream), "hipEventRecord");
        for (int k = 0; k < nk && rc == PBH_OK; ++k) {
            rc = steps[k].launch(p->stream);
            if (rc == PBH_OK) evok(hipEventRecord(ev[k + 1], p->stream), "hipEventRecord");
        }
        evok(hipStreamSynchronize(p->stream), "hipStreamSynchronize");
        for (int k = 0; k < nk && rc == PBH_OK; ++k) {
            float ms = 0.f;
            if (evok(hipEventElapsedTime(&ms, ev[k], ev[k + 1]), "hipEventElapsedTime")) acc[k] += ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    PBHCHECK(rc);
    for (int j = 0; j < (int)uniq.size(); ++j) {
        ms_per_kernel[j] = 0.f;
        if (names) names[j] = uniq[j];
    }
    for (int k = 0; k < nk; ++k) ms_per_kernel[slot[k]] += (float)(acc[k] / iters);
    *nkernel = (int)uniq.size();
    return PBH_OK;
}

#ifndef PBH_F64
// mode 0: copy a -> b (two buffers of `bytes`); mode 1: read-modify-write of ONE buffer in place (what the three middle
// passes do to the planar work buffer).  Mean milliseconds per launch over `iters` launches, HIP events on the null stream.
int pbh_stream_bench(int device, int64_t bytes, int iters, int mode, float* ms_mean) {
    if (!ms_mean || bytes < 16 || iters <= 0 || mode < 0 || mode > 1) return fail(PBH_ERR_INVALID, "bad argument");
    HIPCHECK(hipSetDevice(device));
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = dev_alloc(nullptr, &a, (size_t)bytes);
    if (rc == PBH_OK && mode == 0) rc = dev_alloc(nullptr, &b, (size_t)bytes);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return x == hipSuccess; };
    float ms = 0.f;
    if (rc == PBH_OK) {
        const int64_t n = bytes / 16;
        const unsigned grid = (unsigned)((n + 1023) / 1024);
        auto launch = [&] {
            if (mode == 0) hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n);
            else hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)a, n);
        };
        ok(hipMemset(a, 0, (size_t)bytes));
        ok(hipEventCreate(&e0));
        ok(hipEventCreate(&e1));
        launch();
        launch();
        ok(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch();
        ok(hipGetLastError());
        ok(hipEventRecord(e1, 0));
        ok(hipEventSynchronize(e1));
        if (e == hipSuccess) ok(hipEventElapsedTime(&ms, e0, e1));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc != PBH_OK) return rc;
    if (e != hipSuccess) return fail(PBH_ERR_HIP, std::string("stream bench: ") + hipGetErrorString(e));
    *ms_mean = ms / iters;
    return PBH_OK;
}
int pbh_copy_bench(int device, int64_t bytes, int iters, float* ms_mean) { return pbh_stream_bench(device, bytes, iters, 0, ms_mean); }

#endif  // !PBH_F64

}  // extern "C"
}  // namespace PBH_NS
