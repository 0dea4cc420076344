// pbhip_api.cpp -- the public C ABI of include/pbhip.h.  The implementation is compiled twice
// (pbhip.hip: pbh32_* for complex64 / float32, pbh64_* for complex128 / float64); this file owns
// the pbh_* symbols and forwards on the plan's dtype.
#include "../../include/pbhip.h"

#include <hip/hip_runtime.h>

#include <unistd.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>

#define PBH_DECLARE_IMPL(P)                                                                                     \
    struct P##plan;                                                                                             \
    extern "C" {                                                                                                \
    const char* P##last_error(void);                                                                            \
    int P##plan_create(P##plan**, int, int64_t, int, int, int, int64_t, int64_t);                               \
    int P##plan_destroy(P##plan*);                                                                              \
    int P##plan_set_stream(P##plan*, void*);                                                                    \
    int P##plan_set_variant(P##plan*, int);                                                                     \
    int P##plan_info(const P##plan*, pbh_plan_info_t*);                                                         \
    int P##plan_buffer_class(P##plan*, const void*, int64_t, int*);                                             \
    int P##chirp_generate(P##plan*, double, double, const double*, double);                                     \
    int P##chirp_upload(P##plan*, const void*, int);                                                            \
    int P##chirp_upload_as(P##plan*, const void*, int, int);                                                    \
    int P##chirp_download(P##plan*, void*, int);                                                                \
    int P##chirp_special(P##plan*, const double*, int);                                                         \
    int P##mix(int, void*, int, const void*, void*, int64_t, int, const double*);                               \
    int P##zero_edges(int, void*, int, void*, int64_t, int, const double*);                                     \
    int P##pol_basis(int, void*, int, const void*, void*, int64_t, int);                                        \
    int P##decimate2(int, void*, int, const void*, void*, int64_t, int);                                        \
    int P##transfer(int, void*, void*, const void*, size_t, int);                                               \
    int P##trim(void);                                                                                          \
    int P##relayout(int, void*, int, const void*, int, int64_t, void*, int, int64_t, int64_t, int);            \
    int P##place(int, void*, int, const void*, int64_t, void*, int64_t, int64_t, int64_t);                     \
    int P##decode(int, void*, const void*, size_t, int, const pbh_raw_layout_t*, int64_t, int64_t, int, int,    \
                  const unsigned char*, float, void*, int, int64_t);                                            \
    int P##dedisperse(P##plan*, const void*, void*, int, int);                                                  \
    int P##dedisperse_layout(P##plan*, const void*, int, int64_t, void*, int, int64_t);                         \
    int P##dedisperse_slice(P##plan*, const void*, void*, int64_t, int64_t);                                    \
    int P##dedisperse_slices(P##plan*, const void*, int, void* const*, const int64_t*, int64_t, int64_t);       \
    int P##dedisperse_mix(P##plan*, const void*, void*, const double*);                                         \
    int P##dedisperse_detect_layout(P##plan*, const void*, int, int64_t, void*, int, int);                      \
    int P##dedisperse_detect(P##plan*, const void*, void*, int, int, int, int);                                 \
    int P##dedisperse_stream(P##plan*, const void*, int64_t, void*, int64_t*, float*);                          \
    int P##stream_stats(const P##plan*, double*, int);                                                          \
    int P##plan_stream_detect(P##plan*, int, int);                                                              \
    int P##dedisperse_stream_raw(P##plan*, const void*, size_t, const pbh_raw_layout_t*, int64_t, int64_t,      \
                                 const unsigned char*, float, void*, int64_t*, float*);                         \
    int P##detect(int, void*, int, const void*, void*, int64_t, int, int, int, int, int, int);                  \
    int P##fft_c2c(int, void*, int, const void*, void*, int64_t, int64_t, int, int, int);                       \
    int P##stft(int, void*, int, const void*, void*, int64_t, int, int, int, int, int, int);                    \
    int P##stft_dedisperse(P##plan*, const void*, int, int, void*, int, int64_t);                               \
    int P##dedisperse_istft(P##plan*, const void*, int, int64_t, int, int, void*);                              \
    int P##plan_profile(P##plan*, const void*, void*, int, float*, int*, const char**);                         \
    }
PBH_DECLARE_IMPL(pbh32_)
PBH_DECLARE_IMPL(pbh64_)
extern "C" {
int pbh32_device_count(void);
const char* pbh32_version(void);
int pbh32_chirp_function(int, void*, double, int64_t, double, double, double, void*, int);
int pbh32_copy_bench(int, int64_t, int, float*);
int pbh32_stream_bench(int, int64_t, int, int, float*);
int pbh32_real_to_complex(int, void*, const void*, void*, int64_t, int);
int pbh32_incoherent(int, void*, const void*, void*, int64_t, int, int, const int64_t*);
int pbh32_incoherent_series(int, void*, const void*, int64_t, void*, int64_t, int64_t, int, int, int, const int64_t*);
}

struct pbh_plan {
    int dtype;
    void* impl;
};

// which implementation produced the most recent failure on this thread (-1: this file)
static thread_local int g_last = 0;
static thread_local std::string g_err;

static int fail_here(int code, const char* msg) {
    g_last = -1;
    g_err = msg;
    return code;
}
static int done(int dtype, int rc) {
    if (rc != PBH_OK) g_last = dtype;
    return rc;
}

#define FORWARD(p, call32, call64)                                            \
    do {                                                                      \
        if (!(p)) return fail_here(PBH_ERR_INVALID, "plan is NULL");          \
        if ((p)->dtype == PBH_C128) return done(PBH_C128, call64);            \
        return done(PBH_C64, call32);                                         \
    } while (0)
#define P32(p) ((pbh32_plan*)(p)->impl)
#define P64(p) ((pbh64_plan*)(p)->impl)

extern "C" {

int pbh_device_count(void) { return pbh32_device_count(); }
const char* pbh_version(void) { return pbh32_version(); }
const char* pbh_last_error(void) {
    if (g_last == -1) return g_err.c_str();
    return g_last == PBH_C128 ? pbh64_last_error() : pbh32_last_error();
}

int pbh_plan_create(pbh_plan** out, int device, int64_t nsample, int nchan, int npol, int dtype,
                    int64_t crop_start, int64_t crop_stop) {
    if (!out) return fail_here(PBH_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dtype != PBH_C64 && dtype != PBH_C128)
        return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
    void* impl = nullptr;
    int rc = dtype == PBH_C128
                 ? pbh64_plan_create((pbh64_plan**)&impl, device, nsample, nchan, npol, dtype, crop_start, crop_stop)
                 : pbh32_plan_create((pbh32_plan**)&impl, device, nsample, nchan, npol, dtype, crop_start, crop_stop);
    if (rc != PBH_OK) return done(dtype, rc);
    *out = new pbh_plan{dtype, impl};
    return PBH_OK;
}

int pbh_plan_destroy(pbh_plan* p) {
    if (!p) return PBH_OK;
    int rc = p->dtype == PBH_C128 ? pbh64_plan_destroy(P64(p)) : pbh32_plan_destroy(P32(p));
    delete p;
    return rc;
}

int pbh_plan_set_stream(pbh_plan* p, void* s) { FORWARD(p, pbh32_plan_set_stream(P32(p), s), pbh64_plan_set_stream(P64(p), s)); }
int pbh_plan_set_variant(pbh_plan* p, int v) { FORWARD(p, pbh32_plan_set_variant(P32(p), v), pbh64_plan_set_variant(P64(p), v)); }
int pbh_plan_info(const pbh_plan* p, pbh_plan_info_t* i) { FORWARD(p, pbh32_plan_info(P32(p), i), pbh64_plan_info(P64(p), i)); }
int pbh_plan_buffer_class(pbh_plan* p, const void* d, int64_t b, int* c) { FORWARD(p, pbh32_plan_buffer_class(P32(p), d, b, c), pbh64_plan_buffer_class(P64(p), d, b, c)); }
int pbh_chirp_generate(pbh_plan* p, double c, double dt, const double* f, double r) {
    FORWARD(p, pbh32_chirp_generate(P32(p), c, dt, f, r), pbh64_chirp_generate(P64(p), c, dt, f, r));
}
int pbh_chirp_upload(pbh_plan* p, const void* c, int loc) { FORWARD(p, pbh32_chirp_upload(P32(p), c, loc), pbh64_chirp_upload(P64(p), c, loc)); }
int pbh_chirp_upload_as(pbh_plan* p, const void* c, int dt, int loc) {
    FORWARD(p, pbh32_chirp_upload_as(P32(p), c, dt, loc), pbh64_chirp_upload_as(P64(p), c, dt, loc));
}
int pbh_chirp_download(pbh_plan* p, void* c, int loc) { FORWARD(p, pbh32_chirp_download(P32(p), c, loc), pbh64_chirp_download(P64(p), c, loc)); }
int pbh_chirp_special(pbh_plan* p, const double* a, int m) { FORWARD(p, pbh32_chirp_special(P32(p), a, m), pbh64_chirp_special(P64(p), a, m)); }
int pbh_mix(int device, void* stream, int dtype, const void* in, void* out, int64_t n, int ns, const double* ft) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_mix(device, stream, dtype, in, out, n, ns, ft));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_mix(device, stream, dtype, in, out, n, ns, ft));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_zero_edges(int device, void* stream, int dtype, void* data, int64_t n, int ns, const double* sh) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_zero_edges(device, stream, dtype, data, n, ns, sh));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_zero_edges(device, stream, dtype, data, n, ns, sh));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_pol_basis(int device, void* stream, int dtype, const void* in, void* out, int64_t np, int tc) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_pol_basis(device, stream, dtype, in, out, np, tc));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_pol_basis(device, stream, dtype, in, out, np, tc));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_decimate2(int device, void* stream, int dtype, const void* in, void* out, int64_t nout, int ns) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_decimate2(device, stream, dtype, in, out, nout, ns));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_decimate2(device, stream, dtype, in, out, nout, ns));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_incoherent(int device, void* stream, const void* in, void* out, int64_t nout, int nchan, int unit, const int64_t* d) {
    return done(PBH_C64, pbh32_incoherent(device, stream, in, out, nout, nchan, unit, d));
}
int pbh_incoherent_series(int device, void* stream, const void* in, int64_t ip, void* out, int64_t op, int64_t nout, int nchan,
                          int spc, int unit, const int64_t* d) {
    return done(PBH_C64, pbh32_incoherent_series(device, stream, in, ip, out, op, nout, nchan, spc, unit, d));
}
int pbh_chirp_function(int device, void* stream, double coeff, int64_t n, double dt, double fc, double fr, void* out, int loc) {
    return done(PBH_C64, pbh32_chirp_function(device, stream, coeff, n, dt, fc, fr, out, loc));
}
int pbh_dedisperse(pbh_plan* p, const void* in, void* out, int il, int ol) {
    FORWARD(p, pbh32_dedisperse(P32(p), in, out, il, ol), pbh64_dedisperse(P64(p), in, out, il, ol));
}
int pbh_decode(int device, void* stream, const void* raw, size_t raw_bytes, int raw_loc, const pbh_raw_layout_t* layout,
               int64_t first, int64_t nsample, int nchan, int npol, const unsigned char* conj_mask, float scale, void* out,
               int out_layout, int64_t out_pitch) {
    return done(PBH_C64, pbh32_decode(device, stream, raw, raw_bytes, raw_loc, layout, first, nsample, nchan, npol, conj_mask,
                                      scale, out, out_layout, out_pitch));
}
int pbh_relayout(int device, void* stream, int dtype, const void* in, int il, int64_t ip, void* out, int ol, int64_t op, int64_t n,
                 int nseries) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_relayout(device, stream, dtype, in, il, ip, out, ol, op, n, nseries));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_relayout(device, stream, dtype, in, il, ip, out, ol, op, n, nseries));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_place(int device, void* stream, int dtype, const void* src, int64_t sp, void* dst, int64_t dp, int64_t nrow, int64_t ncol) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_place(device, stream, dtype, src, sp, dst, dp, nrow, ncol));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_place(device, stream, dtype, src, sp, dst, dp, nrow, ncol));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_trim(void) {
    pbh32_trim();
    pbh64_trim();
    return PBH_OK;
}
int pbh_transfer(int device, void* stream, void* dst, const void* src, size_t bytes, int direction) {
    return done(PBH_C64, pbh32_transfer(device, stream, dst, src, bytes, direction));
}
int pbh_dedisperse_layout(pbh_plan* p, const void* in, int il, int64_t ip, void* out, int ol, int64_t op) {
    FORWARD(p, pbh32_dedisperse_layout(P32(p), in, il, ip, out, ol, op), pbh64_dedisperse_layout(P64(p), in, il, ip, out, ol, op));
}
int pbh_dedisperse_slice(pbh_plan* p, const void* in, void* out, int64_t row, int64_t off) {
    FORWARD(p, pbh32_dedisperse_slice(P32(p), in, out, row, off), pbh64_dedisperse_slice(P64(p), in, out, row, off));
}
int pbh_dedisperse_slices(pbh_plan* p, const void* in, int np, void* const* parts, const int64_t* rows, int64_t row, int64_t off) {
    FORWARD(p, pbh32_dedisperse_slices(P32(p), in, np, parts, rows, row, off), pbh64_dedisperse_slices(P64(p), in, np, parts, rows, row, off));
}
int pbh_dedisperse_mix(pbh_plan* p, const void* in, void* out, const double* ft) {
    FORWARD(p, pbh32_dedisperse_mix(P32(p), in, out, ft), pbh64_dedisperse_mix(P64(p), in, out, ft));
}
int pbh_dedisperse_detect_layout(pbh_plan* p, const void* in, int il, int64_t ip, void* out, int ns, int mode) {
    FORWARD(p, pbh32_dedisperse_detect_layout(P32(p), in, il, ip, out, ns, mode), pbh64_dedisperse_detect_layout(P64(p), in, il, ip, out, ns, mode));
}
int pbh_dedisperse_detect(pbh_plan* p, const void* in, void* out, int ns, int mode, int il, int ol) {
    FORWARD(p, pbh32_dedisperse_detect(P32(p), in, out, ns, mode, il, ol), pbh64_dedisperse_detect(P64(p), in, out, ns, mode, il, ol));
}
int pbh_dedisperse_stream(pbh_plan* p, const void* in, int64_t total, void* out, int64_t* nchunk, float* ms) {
    FORWARD(p, pbh32_dedisperse_stream(P32(p), in, total, out, nchunk, ms), pbh64_dedisperse_stream(P64(p), in, total, out, nchunk, ms));
}
int pbh_stream_stats(const pbh_plan* p, double* out, int n) {
    FORWARD(p, pbh32_stream_stats(P32(p), out, n), pbh64_stream_stats(P64(p), out, n));
}
int pbh_plan_stream_detect(pbh_plan* p, int mode, int nscrunch) {
    FORWARD(p, pbh32_plan_stream_detect(P32(p), mode, nscrunch), pbh64_plan_stream_detect(P64(p), mode, nscrunch));
}
int pbh_dedisperse_stream_raw(pbh_plan* p, const void* raw, size_t raw_bytes, const pbh_raw_layout_t* layout, int64_t first,
                              int64_t total, const unsigned char* conj_mask, float scale, void* out, int64_t* nchunk, float* ms) {
    FORWARD(p, pbh32_dedisperse_stream_raw(P32(p), raw, raw_bytes, layout, first, total, conj_mask, scale, out, nchunk, ms),
            pbh64_dedisperse_stream_raw(P64(p), raw, raw_bytes, layout, first, total, conj_mask, scale, out, nchunk, ms));
}
int pbh_detect(int device, void* stream, int dtype, const void* in, void* out, int64_t n, int nchan, int npol, int mode,
               int ns, int il, int ol) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_detect(device, stream, dtype, in, out, n, nchan, npol, mode, ns, il, ol));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_detect(device, stream, dtype, in, out, n, nchan, npol, mode, ns, il, ol));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_fft_c2c(int device, void* stream, int dtype, const void* in, void* out, int64_t n, int64_t batch, int inverse,
                int il, int ol) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_fft_c2c(device, stream, dtype, in, out, n, batch, inverse, il, ol));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_fft_c2c(device, stream, dtype, in, out, n, batch, inverse, il, ol));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_stft(int device, void* stream, int dtype, const void* in, void* out, int64_t nseg, int nperseg, int nchan,
             int inner, int inverse, int il, int ol) {
    if (dtype == PBH_C128) return done(PBH_C128, pbh64_stft(device, stream, dtype, in, out, nseg, nperseg, nchan, inner, inverse, il, ol));
    if (dtype == PBH_C64) return done(PBH_C64, pbh32_stft(device, stream, dtype, in, out, nseg, nperseg, nchan, inner, inverse, il, ol));
    return fail_here(PBH_ERR_UNSUPPORTED, "dtype must be PBH_C64 or PBH_C128");
}
int pbh_stft_dedisperse(pbh_plan* p, const void* in, int nperseg, int nchan_in, void* out, int ol, int64_t op) {
    FORWARD(p, pbh32_stft_dedisperse(P32(p), in, nperseg, nchan_in, out, ol, op), pbh64_stft_dedisperse(P64(p), in, nperseg, nchan_in, out, ol, op));
}
int pbh_dedisperse_istft(pbh_plan* p, const void* in, int il, int64_t ip, int nperseg, int nchan_out, void* out) {
    FORWARD(p, pbh32_dedisperse_istft(P32(p), in, il, ip, nperseg, nchan_out, out), pbh64_dedisperse_istft(P64(p), in, il, ip, nperseg, nchan_out, out));
}
int pbh_plan_profile(pbh_plan* p, const void* in, void* out, int iters, float* ms, int* nk, const char** names) {
    FORWARD(p, pbh32_plan_profile(P32(p), in, out, iters, ms, nk, names), pbh64_plan_profile(P64(p), in, out, iters, ms, nk, names));
}
int pbh_real_to_complex(int device, void* stream, const void* in, void* out, int64_t nreal, int nseries) {
    return done(PBH_C64, pbh32_real_to_complex(device, stream, in, out, nreal, nseries));
}
int pbh_copy_bench(int device, int64_t bytes, int iters, float* ms) { return done(PBH_C64, pbh32_copy_bench(device, bytes, iters, ms)); }
int pbh_stream_bench(int device, int64_t bytes, int iters, int mode, float* ms) { return done(PBH_C64, pbh32_stream_bench(device, bytes, iters, mode, ms)); }

// ---- node-level sharing of device buffers between the ranks of one node (one process per GPU) -----------------
// The reference gathers chunked results with Signal.compute() (pulsarbat/core.py:298-309).  Here the gather is done by
// the producers: the destination rank allocates the full-band block (pbh_node_alloc), exports it (pbh_node_export), the
// other ranks map it (pbh_node_import, xGMI peer access) and their pipelines write their channel slices into it
// (pbh_dedisperse_slice).  Handle exchange and the closing barrier are the host's business (torch.distributed).
static_assert(sizeof(hipIpcMemHandle_t) == sizeof(pbh_ipc_handle_t), "pbh_ipc_handle_t must hold a hipIpcMemHandle_t");

static int hip_fail(const char* what, hipError_t e) {
    g_last = -1;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return PBH_ERR_HIP;
}

int pbh_node_alloc(int device, size_t bytes, void** dev_ptr) {
    if (!dev_ptr || bytes == 0) return fail_here(PBH_ERR_INVALID, "pbh_node_alloc: bad argument");
    *dev_ptr = nullptr;
    // a peer's hipIpcOpenMemHandle on a large allocation never returns on this ROCm stack (measured between two processes on
    // one device: 2040 MiB maps in 0.1 ms, 2056 MiB hangs; the cross-device case is unobserved).  The limit is the largest
    // size SEEN to map, not "just under 2 GiB": a request of 2^31 - 1 bytes rounds up to exactly 2^31 in the allocator.
    // Refuse here, where it is an error, rather than there, where it is a hang.
    if (bytes > PBH_NODE_MAX_BYTES)
        return fail_here(PBH_ERR_UNSUPPORTED, "pbh_node_alloc: a buffer that peers map must be at most 2040 MiB (beyond 2 GiB the "
                                              "peer's mapping call hangs); build larger blocks from row-chunks (pbh_dedisperse_slices)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    e = hipMalloc(dev_ptr, bytes);   // a whole allocation of its own: exportable, unlike a caching allocator's sub-block
    if (e != hipSuccess) {
        g_last = -1;
        g_err = std::string("pbh_node_alloc: hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        return PBH_ERR_NOMEM;
    }
    return PBH_OK;
}

int pbh_node_free(int device, void* dev_ptr) {
    if (!dev_ptr) return PBH_OK;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // peers may only just have finished writing
    if (e == hipSuccess) e = hipFree(dev_ptr);
    return e == hipSuccess ? PBH_OK : hip_fail("pbh_node_free", e);
}

int pbh_node_export(int device, void* dev_ptr, pbh_ipc_handle_t* handle) {
    if (!dev_ptr || !handle) return fail_here(PBH_ERR_INVALID, "pbh_node_export: NULL argument");
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle), dev_ptr);
    return e == hipSuccess ? PBH_OK : hip_fail("pbh_node_export: hipIpcGetMemHandle", e);
}

int pbh_node_import(int device, const pbh_ipc_handle_t* handle, void** dev_ptr) {
    if (!dev_ptr || !handle) return fail_here(PBH_ERR_INVALID, "pbh_node_import: NULL argument");
    *dev_ptr = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        hipIpcMemHandle_t h;
        memcpy(&h, handle, sizeof(h));
        e = hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
    }
    return e == hipSuccess ? PBH_OK : hip_fail("pbh_node_import: hipIpcOpenMemHandle", e);
}

int pbh_node_release(int device, void* dev_ptr) {
    if (!dev_ptr) return PBH_OK;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcCloseMemHandle(dev_ptr);
    return e == hipSuccess ? PBH_OK : hip_fail("pbh_node_release: hipIpcCloseMemHandle", e);
}

// ---- the same sharing through the virtual-memory-management API (round 3) ----------------------------------------------
// hipIpcOpenMemHandle never returns for allocations beyond 2 GiB on this ROCm stack, which forced destination blocks to be runs of
// <= 1-GiB chunks and a joining copy afterwards.  A physical allocation made with hipMemCreate and exported as a POSIX file
// descriptor has no such limit (tools/micro/vmmprobe.hip: one 3-GiB allocation, and a 3-GiB range of three 1-GiB allocations,
// shared between two processes; 1 GiB: same) -- a destination block is then ONE contiguous buffer of any size.  The descriptor
// travels between the processes over a Unix socket (SCM_RIGHTS; the Python host does that, pulsarbat_amd/node.py).
namespace {
struct SharedMapping { hipMemGenericAllocationHandle_t handle; size_t bytes; };
std::mutex g_shared_mu;
std::map<void*, SharedMapping> g_shared;

int shared_map(int device, hipMemGenericAllocationHandle_t h, size_t bytes, void** dev_ptr) {
    void* va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, bytes, 0, nullptr, 0);
    if (e != hipSuccess) return hip_fail("hipMemAddressReserve", e);
    e = hipMemMap(va, bytes, 0, h, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, bytes, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(va, bytes);
    }
    if (e != hipSuccess) {
        (void)hipMemAddressFree(va, bytes);
        return hip_fail("hipMemMap / hipMemSetAccess", e);
    }
    std::lock_guard<std::mutex> lk(g_shared_mu);
    g_shared[va] = SharedMapping{h, bytes};
    *dev_ptr = va;
    return PBH_OK;
}
size_t shared_round(int device, size_t bytes, hipMemAllocationProp* prop) {
    prop->type = hipMemAllocationTypePinned;
    prop->requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    prop->location.type = hipMemLocationTypeDevice;
    prop->location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
        (void)hipGetLastError();
        gran = 2u << 20;
    }
    return (bytes + gran - 1) / gran * gran;
}
}  // namespace

int pbh_node_share_alloc(int device, size_t bytes, void** dev_ptr, int* fd_out) {
    if (!dev_ptr || !fd_out || bytes == 0) return fail_here(PBH_ERR_INVALID, "pbh_node_share_alloc: bad argument");
    *dev_ptr = nullptr;
    *fd_out = -1;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    hipMemAllocationProp prop{};
    const size_t size = shared_round(device, bytes, &prop);
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, size, &prop, 0);
    if (e != hipSuccess) {
        g_last = -1;
        g_err = std::string("pbh_node_share_alloc: hipMemCreate(") + std::to_string(size) + "): " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PBH_ERR_NOMEM : PBH_ERR_HIP;
    }
    int fd = -1;
    e = hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0);
    if (e != hipSuccess) {
        (void)hipMemRelease(h);
        return hip_fail("pbh_node_share_alloc: hipMemExportToShareableHandle", e);
    }
    const int rc = shared_map(device, h, size, dev_ptr);
    if (rc != PBH_OK) {
        close(fd);
        (void)hipMemRelease(h);
        return rc;
    }
    *fd_out = fd;
    return PBH_OK;
}

int pbh_node_share_import(int device, int fd, size_t bytes, void** dev_ptr) {
    if (dev_ptr) *dev_ptr = nullptr;   // (also on the error paths: found by tests/sanitizer/api_nodevice_test.cpp)
    if (!dev_ptr || fd < 0 || bytes == 0) return fail_here(PBH_ERR_INVALID, "pbh_node_share_import: bad argument");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    hipMemAllocationProp prop{};
    const size_t size = shared_round(device, bytes, &prop);
    hipMemGenericAllocationHandle_t h;
    // The HIP runtimes in use disagree about `osHandle`: 7.0 (the one torch 2.10+rocm7.0 bundles and loads first) takes a POINTER to
    // the descriptor and dies with SIGSEGV on CUDA's form, the descriptor's value cast to a pointer; 7.2 takes the value.  The form
    // is chosen from the version of the runtime that is actually loaded (hipRuntimeGetVersion: major * 10^7 + minor * 10^5 + patch):
    //   7.0.x      pointer form ONLY (the value form would be dereferenced: a crash, never an error code)
    //   >= 7.2     value form, then the pointer form as a fallback (a pointer handed to a value-taking runtime is merely a bad
    //              descriptor number: an error)
    //   otherwise  (7.1, older, unreadable) the round-3 probe: pointer form first, value form on error
    int rv = 0;
    if (hipRuntimeGetVersion(&rv) != hipSuccess) { (void)hipGetLastError(); rv = 0; }
    const int major = rv / 10000000, minor = (rv / 100000) % 100;
    const bool value_first = major > 7 || (major == 7 && minor >= 2);
    const bool pointer_only = major == 7 && minor == 0;
    int fdv = fd;
    void* const as_pointer = (void*)&fdv;
    void* const as_value = (void*)(uintptr_t)fd;
    e = hipMemImportFromShareableHandle(&h, value_first ? as_value : as_pointer, hipMemHandleTypePosixFileDescriptor);
    if (e != hipSuccess && !pointer_only) {
        (void)hipGetLastError();
        e = hipMemImportFromShareableHandle(&h, value_first ? as_pointer : as_value, hipMemHandleTypePosixFileDescriptor);
    }
    if (e != hipSuccess) return hip_fail("pbh_node_share_import: hipMemImportFromShareableHandle", e);
    const int rc = shared_map(device, h, size, dev_ptr);
    if (rc != PBH_OK) (void)hipMemRelease(h);
    return rc;
}

int pbh_node_share_free(int device, void* dev_ptr) {
    if (!dev_ptr) return PBH_OK;
    SharedMapping m;
    {
        std::lock_guard<std::mutex> lk(g_shared_mu);
        auto it = g_shared.find(dev_ptr);
        if (it == g_shared.end()) return fail_here(PBH_ERR_INVALID, "pbh_node_share_free: not a shared mapping of this process");
        m = it->second;
        g_shared.erase(it);
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // peers may only just have finished writing
    hipError_t e2 = hipMemUnmap(dev_ptr, m.bytes);
    if (e == hipSuccess) e = e2;
    e2 = hipMemAddressFree(dev_ptr, m.bytes);
    if (e == hipSuccess) e = e2;
    e2 = hipMemRelease(m.handle);
    if (e == hipSuccess) e = e2;
    return e == hipSuccess ? PBH_OK : hip_fail("pbh_node_share_free", e);
}

}  // extern "C"
