// pbhip_internal.hpp -- what the translation units of libpbhip.so share: the per-precision names of the C ABI, the plan, the
// step list, error plumbing and the prototypes of the functions one unit calls in another.  Units (each compiled twice, float32
// and -DPBH_F64):  pbhip.hip  plans, kernel launch sequences, chirp, the hot-path and next-row entry points, stand-alone transforms;
//                 pbhip_stream.hip   the overlap-save streaming drivers (BASELINE configs[3]);
//                 pbhip_measure.hip  per-kernel profile and the streaming yardstick.
// Everything that is not extern "C" lives in namespace PBH_NS (pbh32 / pbh64): the two precision builds are linked into one
// library and must not share a symbol.
#pragma once
// Compiled twice (float32, and float64 with -DPBH_F64): every public name below is renamed to the
// per-precision prefix (pbh32_* / pbh64_*); pbhip_api.cpp owns the public pbh_* symbols and dispatches
// on the plan's dtype.
#include "pbh_config.hpp"
#define pbh_plan PBH_FN(plan)
#define pbh_device_count PBH_FN(device_count)
#define pbh_last_error PBH_FN(last_error)
#define pbh_version PBH_FN(version)
#define pbh_plan_create PBH_FN(plan_create)
#define pbh_plan_destroy PBH_FN(plan_destroy)
#define pbh_plan_set_stream PBH_FN(plan_set_stream)
#define pbh_plan_set_variant PBH_FN(plan_set_variant)
#define pbh_plan_info PBH_FN(plan_info)
#define pbh_chirp_generate PBH_FN(chirp_generate)
#define pbh_chirp_upload PBH_FN(chirp_upload)
#define pbh_chirp_upload_as PBH_FN(chirp_upload_as)
#define pbh_chirp_download PBH_FN(chirp_download)
#define pbh_chirp_function PBH_FN(chirp_function)
#define pbh_chirp_special PBH_FN(chirp_special)
#define pbh_mix PBH_FN(mix)
#define pbh_zero_edges PBH_FN(zero_edges)
#define pbh_pol_basis PBH_FN(pol_basis)
#define pbh_decimate2 PBH_FN(decimate2)
#define pbh_incoherent PBH_FN(incoherent)
#define pbh_incoherent_series PBH_FN(incoherent_series)
#define pbh_transfer PBH_FN(transfer)
#define pbh_decode PBH_FN(decode)
#define pbh_trim PBH_FN(trim)
#define pbh_relayout PBH_FN(relayout)
#define pbh_dedisperse_stream_raw PBH_FN(dedisperse_stream_raw)
#define pbh_dedisperse PBH_FN(dedisperse)
#define pbh_dedisperse_layout PBH_FN(dedisperse_layout)
#define pbh_dedisperse_slice PBH_FN(dedisperse_slice)
#define pbh_dedisperse_slices PBH_FN(dedisperse_slices)
#define pbh_dedisperse_mix PBH_FN(dedisperse_mix)
#define pbh_place PBH_FN(place)
#define pbh_dedisperse_detect_layout PBH_FN(dedisperse_detect_layout)
#define pbh_dedisperse_detect PBH_FN(dedisperse_detect)
#define pbh_dedisperse_stream PBH_FN(dedisperse_stream)
#define pbh_stream_stats PBH_FN(stream_stats)
#define pbh_plan_stream_detect PBH_FN(plan_stream_detect)
#define pbh_dedisperse_istft PBH_FN(dedisperse_istft)
#define pbh_real_to_complex PBH_FN(real_to_complex)
#define pbh_detect PBH_FN(detect)
#define pbh_fft_c2c PBH_FN(fft_c2c)
#define pbh_plan_profile PBH_FN(plan_profile)
#define pbh_copy_bench PBH_FN(copy_bench)
#define pbh_stream_bench PBH_FN(stream_bench)
#define pbh_plan_buffer_class PBH_FN(plan_buffer_class)
#include "../../include/pbhip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <algorithm>
#include <vector>

#include "fft_core.hpp"
#include "host_sched.hpp"


namespace PBH_NS {
// ---- environment switches ---------------------------------------------------------------------------------
// SHIPPED (read by every build, documented in README.md "Environment"): PBH_FD4 (0: five-pass schedule only), PBH_CLASS
// (0: no allocation-class probing), PBH_TRACE_ALLOC (allocations and placement decisions on stderr), PBH_STREAM_WINDOW_MB and
// PBH_STREAM_EPOCH (device window of the streaming drivers), PBH_QMAX (rows of a column tile: forces the split column transform
// at small sizes), PBH_ROW_GRID (workgroups of the persistent kernels; default one per CU), PBH_MIXED (7-smooth lengths: 0
// padded convolution, 1 one-level plans, 2 two-level too).  Everything else is an EXPERIMENT switch of the A/B runs recorded
// in DESIGN.md 6-6d and exists only in builds made with PBH_EXTRA_FLAGS="-DPBH_DIAGNOSTIC": a product build takes the default
// (tests/test_abi.py counts the getenv calls of the shipped sources).
static inline const char* diag_env(const char* name) {
#ifdef PBH_DIAGNOSTIC
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- error plumbing ------------------------------------------------------------------------------------
extern thread_local std::string g_err;   // (defined in pbhip.hip; pbh_last_error returns it)

inline int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHECK(expr)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(PBH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)
#define PBHCHECK(expr)              \
    do {                            \
        int r_ = (expr);            \
        if (r_ != PBH_OK) return r_; \
    } while (0)

// ---- plan ------------------------------------------------------------------------------------------------
constexpr size_t kCounterBytes = 4096;   // tile hand-out counters of the persistent kernels, behind the stage twiddle table
constexpr int kCounters = (int)(kCounterBytes / sizeof(unsigned));
struct Step {
    const char* name;
    std::function<int(hipStream_t)> launch;
};

}  // namespace PBH_NS
using namespace PBH_NS;

// (the plan is the C ABI's opaque `pbh_plan`: global scope, like its declaration in include/pbhip.h)
struct pbh_plan {
    int device = 0;
    int64_t N = 0, start = 0, stop = 0;
    int nchan = 0, npol = 0, S = 0;
    int N1 = 1, N2 = 1;
    int P = 1;  // N1 = P * Q: column transform split into a radix-P stage (k_radix_p) and P blocks of Q rows (k_colq)
    int variant = PBH_VARIANT_AUTO;
    hipStream_t stream = nullptr;
    bool has_chirp = false;
    int perm_w = 0;  // chirp row order: 0 natural, 8 = wave-decoupled row kernel (k_row2)

    cf* work = nullptr;      // planar workspace, S * N
    cf* work2 = nullptr;     // second planar workspace: the middle passes of the power-of-two planar pipeline ping-pong (oop_ok)
    // Allocation "classes" (pair_class below): a pass streaming from one large allocation into another runs ~5 % faster
    // when the two are of different class.  work2 is chosen of the class opposite to work's when it is allocated;
    // t_same / t_diff are the probe copy's times for a same-class and a different-class pair, the cache remembers what
    // the caller's buffers turned out to be (0 = work's class, 1 = the other, -1 = not decidable).
    float cls_t_same = 0.f, cls_t_diff = 0.f;
    size_t cls_len = 0;
    struct ClsEntry { const void* ptr = nullptr; size_t bytes = 0; int cls = -1; };
    ClsEntry cls_cache[8];
    int cls_next = 0, cls_probes = 0;   // (at most kClassProbes probe copies per plan: a caller with ever new arrays is not probed for ever)
    // four-pass schedule: which of the two work buffers holds the Q4 intermediate, per (input, output) pair -- decided by
    // timing both assignments on the first call with that pair (fd4_roles)
    struct RoleEntry { const void* in = nullptr; const void* out = nullptr; int swap = 0; };
    RoleEntry role_cache[8];
    int role_next = 0;
    int role_tunes = 0, role_last = 0;   // pairs timed so far (at most kRoleTunes per plan), the last decision
    int fd4_force = -1;   // >= 0 while fd4_roles times an assignment
    real* det_part = nullptr;  // detect tail fused into the inverse column pass: per-tile power sums (ColpParams::det_part),
    size_t det_bytes = 0;      // S * N / 16 floats + S * N1 * (N2 / nscrunch) for the groups with a scrunch boundary
    cf* chirp = nullptr;     // plan order, nchan * N, pre-scaled by 1/N
    float* chirp_phase = nullptr;  // same order, revolutions: what k_rowp reads (generated chirps only)
    bool has_phase = false;
    bool phase16 = false;          // ... with every 2^14-bin row in k_rowp16's order (ChirpParams::phase16)
    cf* tw16k = nullptr;     // W_16384^p
    double2* tw_hi = nullptr;
    double2* tw_lo = nullptr;
    int tw_shift = 0;
    double* chan_freq = nullptr;
    double* mix_ft = nullptr;   // per-series mixer frequencies of pbh_dedisperse_mix
    // Bluestein (nsample not a power of two, or < 32): two runs of a power-of-two sub-plan
    bool plain_fft = false;     // plan backs pbh_fft_c2c: Bluestein ring for every length, no chirp buffer
    int64_t bsL = 0;            // ring length, power of two >= 2N-1; 0 = not a Bluestein plan
    pbh_plan* sub = nullptr;    // (bsL, 1 chan, S "pols") plan whose chirp is FFT_L(wrapped conj b)/L
    cf* bs_b = nullptr;         // b[n] = exp(-i pi n^2/N)
    cf* bs_a = nullptr;         // (bsL, S) pipeline input
    cf* bs_conv = nullptr;      // (bsL, S) pipeline output
    // Dedispersion of such lengths is ONE power-of-two run: ifft_N(fft_N(x) H) is the circular convolution
    // of x with h = ifft_N(H), i.e. outputs N-1 .. 2N-2 of the linear convolution of x with the N-periodic
    // h laid out over 2N-1 taps, and that is the pipeline of a (bsL, nchan, npol) plan whose "chirp" is
    // FFT_L of those taps and whose crop is [N-1+start, N-1+stop)  (rebuild_circular_filter)
    pbh_plan* cfilt = nullptr;
    cf* cf_in = nullptr;        // (bsL, S) zero-padded copy of the input
    void* stage_in = nullptr;   // device staging for host inputs
    void* stage_out = nullptr;  // device staging for host outputs
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    void* det_mid = nullptr;    // dedispersed voltages of the two-step detect (plans / scrunch factors without a fused tail)
    size_t det_mid_bytes = 0;
    int64_t owned_bytes = 0;
    // 7-smooth lengths (mixed_kernels.hpp): N = N1 * N2, N2 = 2^k rows of the power-of-two engine, N1 = P * Q any 7-smooth
    // number with P, Q <= kMixMaxLen; both column roles run k_colmix (mixP: the P-point stage, mixQ: the Q-point pass)
    struct MixTable {
        int L = 0, nstage = 0;
        int radix[kMixMaxStages] = {};
        cf* wl = nullptr;
        unsigned short* perm = nullptr;
    };
    bool mixed = false;
    MixTable mixP, mixQ;
    // ... and, when the length has too few factors of two for the 2^k engine's rows, the rows as well (k_rowmix): mixR.perm
    // is then the chirp's row order (position -> bin)
    bool rowmix = false;
    MixTable mixR;
    double stream_stats[PBH_STREAM_NSTATS] = {};   // of the last streaming call (pbh_stream_stats)
    int stream_detect_mode = -1, stream_detect_ns = 1;   // pbh_plan_stream_detect: the streaming calls write detected rows
    double gen_coeff = 0, gen_inv_ndt = 0, gen_inv_ref = 0;   // parameters of the generated chirp (k_rowp16's on-the-fly phase)
    bool chirp_lazy = false;   // the generated chirp exists as phase rows only; `chirp` is filled by materialize_chirp on demand
};

namespace PBH_NS {

struct DetectTail {
    real* out = nullptr;   // non-null: replace the final layout pass by detect + scrunch into `out`
    int mode = 0, nscrunch = 1;
};

struct IoLayout {
    int in_layout = PBH_LAYOUT_SAMPLE_MAJOR, out_layout = PBH_LAYOUT_SAMPLE_MAJOR;
    int64_t in_pitch = 0, out_pitch = 0;
    int64_t in_valid = -1;   // sample-major input: time samples present (the rest of nsample is zero padding); -1 = all
    const double* mix_ft = nullptr;   // sample-major input: per-series mixer frequencies (device), applied by the de-interleave pass
    int64_t out_row_elems = 0;   // sample-major output: elements between consecutive rows (0 = compact, S): the rows are a
                                 // channel slice of a wider (nout, nchan_total, npol) array (pbh_dedisperse_slice)
    // the output rows may be split over several buffers (row-chunks of a destination block, each its own allocation):
    // part i holds output rows [part_row[i], part_row[i+1]) starting at part_ptr[i]; empty = one buffer, `out`
    std::vector<cf*> part_ptr;
    std::vector<int64_t> part_row;
};

// ---- functions of pbhip.hip the other units call ------------------------------------------------------------------------
int dev_alloc(pbh_plan* p, void** ptr, size_t bytes);
std::vector<Step> build_steps(pbh_plan* p, const cf* in, cf* out, DetectTail tail = DetectTail(), IoLayout io = IoLayout());
int run_steps(std::vector<Step>& steps, hipStream_t st);
typedef pbh_host::Span DecodeSpan;   // {b0, b1: first / last block touched; off, len: bytes [off, off + len) of the raw buffer read}
int decode_span(const pbh_raw_layout_t* L, int64_t first, int64_t nsample, int nchan, int npol, size_t raw_bytes, DecodeSpan* sp);
int decode_launch(const unsigned char* draw, int64_t skip, const pbh_raw_layout_t* L, int64_t first, int64_t nsample, int nchan,
                  int npol, const unsigned char* dconj, float scale, void* out_dev, int out_layout, int64_t out_pitch, hipStream_t st);
hipError_t xfer_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st);
hipError_t xfer_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st);
int detect_out_elems(int mode, int npol);
int pin_host_range(void* ptr, size_t bytes);
bool can_fuse_detect(const pbh_plan* p, int nscrunch, int mode);
}  // namespace PBH_NS
