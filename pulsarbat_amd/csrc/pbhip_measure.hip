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
// passes of the five-pass schedule do to the planar work buffer).  Mean milliseconds per launch over `iters` launches, HIP
// events on the null stream.  A copy between two allocations of one class is 4-5 % slower than between two classes (DESIGN.md
// 6d d), and consecutive allocations usually share theirs: for buffers of 1 GiB and more mode 0 therefore walks through up to
// twelve destination candidates (all held, two timed copies each) and measures with the fastest -- the ceiling is what a copy
// CAN do on this part, as the passes of the four-pass schedule are arranged to.
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
        if (mode == 0 && bytes >= ((int64_t)1 << 30) && hipMemset(a, 0, (size_t)bytes) == hipSuccess) {
            constexpr int NC = 12;
            void* cand[NC] = {b};
            float t[NC];
            int nc = 1, best = 0;
            float tmax = 0.f;
            hipEvent_t c0 = nullptr, c1 = nullptr;
            if (hipEventCreate(&c0) == hipSuccess && hipEventCreate(&c1) == hipSuccess) {
                size_t mfree = 0, mtotal = 0;
                (void)hipMemGetInfo(&mfree, &mtotal);
                while (true) {
                    float tb = -1.f;
                    for (int rep = 0; rep < 3; ++rep) {
                        (void)hipEventRecord(c0, 0);
                        hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)cand[nc - 1], n);
                        (void)hipEventRecord(c1, 0);
                        float x = 0.f;
                        if (hipEventSynchronize(c1) == hipSuccess && hipEventElapsedTime(&x, c0, c1) == hipSuccess && rep > 0 && (tb < 0 || x < tb)) tb = x;
                    }
                    t[nc - 1] = tb;
                    if (tb > tmax) tmax = tb;
                    if (tb > 0 && tb < t[best]) best = nc - 1;
                    // two groups seen and this one in the fast one, or out of candidates / memory
                    if ((t[best] > 0 && t[best] < 0.97f * tmax) || nc == NC || (size_t)bytes * (size_t)(nc + 1) > mfree / 4) break;
                    if (hipMalloc(&cand[nc], (size_t)bytes) != hipSuccess) break;
                    ++nc;
                }
            }
            if (c0) (void)hipEventDestroy(c0);
            if (c1) (void)hipEventDestroy(c1);
            (void)hipGetLastError();
            for (int i = 0; i < nc; ++i)
                if (i != best) (void)hipFree(cand[i]);
            b = cand[best];
        }
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
