// pbhip_stream.hip -- unit 2 of 3: the overlap-save streaming drivers (BASELINE configs[3]; pbhip_internal.hpp lists the units).
// The schedules themselves -- epochs, device windows, what every chunk uploads and takes over -- are csrc/host_sched.hpp, which
// the CPU box runs under the sanitizers; this file executes them with streams, events and pinned host memory.
#include "pbhip_internal.hpp"

namespace PBH_NS {
extern "C" {

// ---- streaming overlap-save (BASELINE configs[3]) ---------------------------------------------------------------------
// A long host-resident block is dedispersed in chunks of plan->N samples that overlap by N - hop,
// hop = stop - start: chunk k covers input rows [k*hop, k*hop + N) and yields output rows
// [k*hop, (k+1)*hop) -- exactly `concatenate([coherent_dedispersion(z[k*hop : k*hop+N]) for k])` of the
// reference (each chunk is one reference call; its crop is the valid region of an overlap-save step:
// dedispersion.py:127-133, transforms.py:59-148).
//
// Every input row crosses PCIe ONCE.  The stream's samples exist once on the host (transforms.py:101-110; readers are
// offset-addressed, readers/_base.py:298-333) and the N - hop rows two consecutive chunks share stay in HBM: the device
// holds a WINDOW of N + (B-1)*hop consecutive rows; chunk k of an epoch of B chunks reads rows [j*hop, j*hop + N) of it
// (j = k mod B) and only the hop rows it adds are uploaded.  Two windows alternate between epochs: the first chunk of an
// epoch gets the N - hop rows it shares with its predecessor by one device-to-device copy out of the other window's tail
// (on the compute stream), so uploads never wait for kernels except for the window of two epochs ago.  H2D, kernels and D2H
// run on three streams chained by events.  Host memory is page-locked for the duration of the call (hipHostRegister); a
// range that cannot be page-locked and is not pinned already is an error -- pageable memory is never handed to
// hipMemcpyAsync (see "host <-> device transfers" above).
namespace {
struct StreamRig {
    hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    hipEvent_t ev_epoch[2] = {nullptr, nullptr};   // window w has been read for the last time (chunks and the slide out of it)
    hipEvent_t t[6] = {};                          // timing: whole call, H2D stream, D2H stream (first / last)
    std::vector<hipEvent_t> kev;                   // timing: around every chunk's kernels
    void* reg[2] = {nullptr, nullptr};             // host ranges registered by this call
    int rc = PBH_OK;
    const char* who;

    explicit StreamRig(const char* name) : who(name) {}
    bool ok(hipError_t e, const char* what) {
        if (e != hipSuccess && rc == PBH_OK) rc = fail(PBH_ERR_HIP, std::string(who) + ": " + what + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    }
    void pin(int slot, void* ptr, size_t bytes) {
        if (rc != PBH_OK) return;
        const int r = pin_host_range(ptr, bytes);
        if (r == 0) reg[slot] = ptr;
        if (r < 0)
            rc = fail(PBH_ERR_HIP, std::string(who) + ": the host buffers cannot be page-locked (hipHostRegister failed); "
                                   "pageable memory is never handed to asynchronous copies");
    }
    void create(int64_t nchunk) {
        if (rc != PBH_OK) return;
        ok(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking), "hipStreamCreate");
        ok(hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking), "hipStreamCreate");
        ok(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking), "hipStreamCreate");
        for (int b = 0; b < 2; ++b) {
            ok(hipEventCreateWithFlags(&ev_in[b], hipEventDisableTiming), "hipEventCreate");
            ok(hipEventCreateWithFlags(&ev_cmp[b], hipEventDisableTiming), "hipEventCreate");
            ok(hipEventCreateWithFlags(&ev_out[b], hipEventDisableTiming), "hipEventCreate");
            ok(hipEventCreateWithFlags(&ev_epoch[b], hipEventDisableTiming), "hipEventCreate");
        }
        for (auto& e : t) ok(hipEventCreate(&e), "hipEventCreate");
        kev.assign((size_t)(2 * nchunk), nullptr);
        for (auto& e : kev) ok(hipEventCreate(&e), "hipEventCreate");
    }
    // after the last chunk has been enqueued: drain, read the clocks (stats: see pbh_stream_stats in pbhip.h)
    void finish(double* stats, float* ms_total) {
        ok(hipStreamSynchronize(s_in), "hipStreamSynchronize");
        ok(hipStreamSynchronize(s_cmp), "hipStreamSynchronize");
        ok(hipEventRecord(t[1], s_out), "hipEventRecord");
        ok(hipStreamSynchronize(s_out), "hipStreamSynchronize");
        if (rc != PBH_OK) return;
        float ms = 0.f;
        if (ok(hipEventElapsedTime(&ms, t[0], t[1]), "hipEventElapsedTime")) stats[5] = ms;
        if (ms_total) *ms_total = ms;
        if (ok(hipEventElapsedTime(&ms, t[2], t[3]), "hipEventElapsedTime")) stats[2] = ms;
        if (ok(hipEventElapsedTime(&ms, t[4], t[5]), "hipEventElapsedTime")) stats[3] = ms;
        double kms = 0.0;
        for (size_t k = 0; k + 1 < kev.size() && rc == PBH_OK; k += 2)
            if (ok(hipEventElapsedTime(&ms, kev[k], kev[k + 1]), "hipEventElapsedTime")) kms += ms;
        stats[4] = kms;
    }
    ~StreamRig() {
        for (void* r : reg)
            if (r) (void)hipHostUnregister(r);
        for (int b = 0; b < 2; ++b) {
            if (ev_in[b]) (void)hipEventDestroy(ev_in[b]);
            if (ev_cmp[b]) (void)hipEventDestroy(ev_cmp[b]);
            if (ev_out[b]) (void)hipEventDestroy(ev_out[b]);
            if (ev_epoch[b]) (void)hipEventDestroy(ev_epoch[b]);
        }
        for (auto e : t)
            if (e) (void)hipEventDestroy(e);
        for (auto e : kev)
            if (e) (void)hipEventDestroy(e);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_cmp) (void)hipStreamDestroy(s_cmp);
        if (s_out) (void)hipStreamDestroy(s_out);
    }
};

// chunks per epoch: as many as keep a window within `PBH_STREAM_WINDOW_MB` (default 2048 MiB), at most 64, or exactly
// `PBH_STREAM_EPOCH` (both read per call so that tests can force many short epochs); `step_bytes` = what one more chunk
// adds to the window
static size_t stream_window_cap() {
    const char* e = getenv("PBH_STREAM_WINDOW_MB");
    return (size_t)(e && atoll(e) > 0 ? atoll(e) : 2048) << 20;
}
static int64_t stream_epoch_max() {
    const char* e = getenv("PBH_STREAM_EPOCH");
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1 && v <= 64 ? v : 0;   // 0: not forced
}
}  // namespace

// Detected output of the streaming calls (pbh_plan_stream_detect): every chunk ends in the fused detect tail and what goes
// back to the host is float32 (hop / nscrunch, nchan[, npol | 4]) rows -- a filterbank stream; the chunks' valid regions
// must be whole scrunch blocks (hop % nscrunch == 0) for the concatenation to be the scrunched stream.
int pbh_plan_stream_detect(pbh_plan* p, int mode, int nscrunch) {
    if (!p) return fail(PBH_ERR_INVALID, "plan is NULL");
    if (mode < 0) {
        p->stream_detect_mode = -1;
        p->stream_detect_ns = 1;
        return PBH_OK;
    }
    if (nscrunch <= 0) return fail(PBH_ERR_INVALID, "nscrunch must be positive");
    if (!detect_out_elems(mode, p->npol)) return fail(PBH_ERR_INVALID, "bad detect mode");
    if (mode != PBH_DETECT_INTENSITY && p->npol != 2) return fail(PBH_ERR_INVALID, "Stokes modes need npol == 2");
    if ((p->stop - p->start) % nscrunch != 0)
        return fail(PBH_ERR_INVALID, "the plan's valid region (crop_stop - crop_start) must be a multiple of nscrunch");
    if (!can_fuse_detect(p, nscrunch, mode))
        return fail(PBH_ERR_UNSUPPORTED, "no fused detect tail for this plan and nscrunch (multi-pass plans; nscrunch % 64 == 0, or 1)");
    p->stream_detect_mode = mode;
    p->stream_detect_ns = nscrunch;
    return PBH_OK;
}
// bytes one chunk of a streaming call leaves, and the tail that makes them
static size_t stream_out_bytes(const pbh_plan* p) {
    const int64_t hop = p->stop - p->start;
    if (p->stream_detect_mode < 0) return sizeof(cf) * (size_t)p->S * (size_t)hop;
    return sizeof(real) * (size_t)(hop / p->stream_detect_ns) * (size_t)p->nchan * (size_t)detect_out_elems(p->stream_detect_mode, p->npol);
}
static DetectTail stream_tail(const pbh_plan* p, void* dout) {
    DetectTail t;
    if (p->stream_detect_mode >= 0) {
        t.out = (real*)dout;
        t.mode = p->stream_detect_mode;
        t.nscrunch = p->stream_detect_ns;
    }
    return t;
}

int pbh_stream_stats(const pbh_plan* p, double* out, int n) {
    if (!p || !out || n < 0) return fail(PBH_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n; ++i) out[i] = i < PBH_STREAM_NSTATS ? p->stream_stats[i] : 0.0;
    return PBH_OK;
}

int pbh_dedisperse_stream(pbh_plan* p, const void* host_in, int64_t total_nsample, void* host_out,
                          int64_t* nchunk_out, float* ms_total) {
    if (!p || !host_in || !host_out) return fail(PBH_ERR_INVALID, "NULL argument");
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    const int64_t N = p->N, hop = p->stop - p->start;
    if (hop <= 0) return fail(PBH_ERR_INVALID, "plan has an empty valid region (stop <= start)");
    if (total_nsample < N) return fail(PBH_ERR_INVALID, "total_nsample is shorter than one chunk");
    HIPCHECK(hipSetDevice(p->device));
    const size_t row = sizeof(cf) * (size_t)p->S;
    // the schedule (epochs, windows, what every chunk uploads and takes over): host_sched.hpp
    pbh_host::RowStream sched;
    if (!pbh_host::row_stream(N, hop, total_nsample, row, stream_window_cap(), stream_epoch_max(), &sched))
        return fail(PBH_ERR_INVALID, "bad stream geometry");
    const int64_t nchunk = sched.nchunk;
    const size_t out_bytes = stream_out_bytes(p);
    const size_t host_in_bytes = row * (size_t)total_nsample, host_out_bytes = out_bytes * (size_t)nchunk;
    const size_t win_bytes = sched.win_bytes;
    const int nwin = sched.nwin;

    void* dwin[2] = {nullptr, nullptr};
    void* dout[2] = {nullptr, nullptr};
    StreamRig rig("pbh_dedisperse_stream");
    int& rc = rig.rc;
    double* stats = p->stream_stats;
    for (int i = 0; i < PBH_STREAM_NSTATS; ++i) stats[i] = 0.0;
    for (int b = 0; b < 2 && rc == PBH_OK; ++b) {
        if (b < nwin && (rc = dev_alloc(nullptr, &dwin[b], win_bytes)) != PBH_OK) break;
        rc = dev_alloc(nullptr, &dout[b], out_bytes);
    }
    rig.pin(0, const_cast<void*>(host_in), host_in_bytes);
    rig.pin(1, host_out, host_out_bytes);
    rig.create(nchunk);
    if (rc == PBH_OK) {
        // the plan's own stream may hold pending work (chirp generation): order after it
        rig.ok(hipStreamSynchronize(p->stream), "hipStreamSynchronize");
        rig.ok(hipEventRecord(rig.t[0], rig.s_in), "hipEventRecord");
        rig.ok(hipEventRecord(rig.t[2], rig.s_in), "hipEventRecord");
        rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.t[0], 0), "hipStreamWaitEvent");
        for (int64_t k = 0; k < nchunk && rc == PBH_OK; ++k) {
            const pbh_host::RowChunk c = pbh_host::row_chunk(sched, k);
            const int64_t e = c.epoch, j = c.j;
            const int w = c.win, b = (int)(k & 1);
            char* win = (char*)dwin[w];
            char* dst = (char*)host_out + (size_t)k * out_bytes;
            // upload the rows this chunk adds: all N for the first chunk, afterwards rows [(k-1)*hop + N, k*hop + N)
            if (j == 0 && e >= 2) rig.ok(hipStreamWaitEvent(rig.s_in, rig.ev_epoch[w], 0), "hipStreamWaitEvent");
            rig.ok(hipMemcpyAsync(win + c.up_dst, (const char*)host_in + c.up_src, c.up_bytes, hipMemcpyHostToDevice, rig.s_in),
                   "hipMemcpyAsync H2D");
            stats[0] += (double)c.up_bytes;
            rig.ok(hipEventRecord(rig.ev_in[b], rig.s_in), "hipEventRecord");
            rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.ev_in[b], 0), "hipStreamWaitEvent");
            if (k >= 2) rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.ev_out[b], 0), "hipStreamWaitEvent");  // out[b] downloaded
            rig.ok(hipEventRecord(rig.kev[(size_t)(2 * k)], rig.s_cmp), "hipEventRecord");
            if (c.handover) {   // a new epoch: the shared rows come from the tail of the other window, which is then free
                if (c.ho_bytes > 0)
                    rig.ok(hipMemcpyAsync(win, (const char*)dwin[w ^ 1] + c.ho_src, c.ho_bytes, hipMemcpyDeviceToDevice, rig.s_cmp),
                           "hipMemcpyAsync D2D");
                stats[7] += (double)c.ho_bytes;
                rig.ok(hipEventRecord(rig.ev_epoch[w ^ 1], rig.s_cmp), "hipEventRecord");
            }
            if (rc == PBH_OK) {
                const DetectTail tail = stream_tail(p, dout[b]);
                auto steps = build_steps(p, (const cf*)(win + c.win_off), tail.out ? nullptr : (cf*)dout[b], tail);
                rc = run_steps(steps, rig.s_cmp);
            }
            rig.ok(hipEventRecord(rig.kev[(size_t)(2 * k + 1)], rig.s_cmp), "hipEventRecord");
            rig.ok(hipEventRecord(rig.ev_cmp[b], rig.s_cmp), "hipEventRecord");
            rig.ok(hipStreamWaitEvent(rig.s_out, rig.ev_cmp[b], 0), "hipStreamWaitEvent");
            if (k == 0) rig.ok(hipEventRecord(rig.t[4], rig.s_out), "hipEventRecord");
            rig.ok(hipMemcpyAsync(dst, dout[b], out_bytes, hipMemcpyDeviceToHost, rig.s_out), "hipMemcpyAsync D2H");
            stats[1] += (double)out_bytes;
            rig.ok(hipEventRecord(rig.ev_out[b], rig.s_out), "hipEventRecord");
        }
        rig.ok(hipEventRecord(rig.t[3], rig.s_in), "hipEventRecord");
        rig.ok(hipEventRecord(rig.t[5], rig.s_out), "hipEventRecord");
        rig.finish(stats, ms_total);
        stats[6] = (double)nchunk;
    }
    for (int b = 0; b < 2; ++b) {
        if (dwin[b]) (void)hipFree(dwin[b]);
        if (dout[b]) (void)hipFree(dout[b]);
    }
    if (rc == PBH_OK && nchunk_out) *nchunk_out = nchunk;
    return rc;
}

// The same overlap-save stream fed with RAW payload bytes (reader-side decode in front): the bytes of the blocks that hold
// the stream's samples cross PCIe once (2 bytes per 8-bit complex sample instead of 8) into the same two-window scheme, now
// over FILE BYTES: an epoch's window holds the bytes from the (16-byte aligned) start of its first chunk's span to the end of
// its last chunk's span, a chunk uploads only the bytes beyond its predecessor's span and a new epoch takes the bytes it
// shares with the previous chunk from the other window.  k_decode writes each chunk series-major on the device, and the
// chunk runs the pipeline without its de-interleave pass.
int pbh_dedisperse_stream_raw(pbh_plan* p, const void* host_raw, size_t raw_bytes, const pbh_raw_layout_t* L,
                              int64_t first, int64_t total_nsample, const unsigned char* conj_mask, float scale, void* host_out,
                              int64_t* nchunk_out, float* ms_total) {
    if (!p || !host_raw || !host_out || !L) return fail(PBH_ERR_INVALID, "NULL argument");
#ifdef PBH_F64
    return fail(PBH_ERR_UNSUPPORTED, "raw streaming decodes to complex64");
#else
    if (!p->has_chirp) return fail(PBH_ERR_STATE, "no chirp: call pbh_chirp_generate or pbh_chirp_upload first");
    if (L->ncomp != 2) return fail(PBH_ERR_INVALID, "raw streaming needs complex samples (ncomp = 2)");
    if (first < 0) return fail(PBH_ERR_INVALID, "first must be non-negative");
    const int64_t hop = p->stop - p->start;
    if (hop <= 0) return fail(PBH_ERR_INVALID, "plan has an empty valid region (stop <= start)");
    if (total_nsample < p->N) return fail(PBH_ERR_INVALID, "total_nsample is shorter than one chunk");
    const int64_t nchunk = (total_nsample - p->N) / hop + 1;
    // spans of all chunks up front: bounds checks, then the epochs and the size of the device windows (host_sched.hpp)
    pbh_host::SpanStream sched;
    {
        std::vector<DecodeSpan> spans((size_t)nchunk);
        for (int64_t k = 0; k < nchunk; ++k)
            PBHCHECK(decode_span(L, first + k * hop, p->N, p->nchan, p->npol, raw_bytes, &spans[(size_t)k]));
        pbh_host::span_stream(std::move(spans), stream_epoch_max() ? stream_epoch_max() : 64,
                              stream_epoch_max() ? SIZE_MAX : stream_window_cap(), &sched);
    }
    const size_t win_bytes = sched.win_bytes;
    HIPCHECK(hipSetDevice(p->device));
    const bool sm = !(p->bsL || p->mixed || p->N1 == 1 || p->N1 / p->P > kTilePoints || p->N2 % (kTilePoints / (p->N1 / p->P)) != 0 ||
                      p->N >= (1LL << 31)) && p->S > 1;
    const size_t row = sizeof(cf) * (size_t)p->S;
    const size_t out_bytes = stream_out_bytes(p), host_out_bytes = out_bytes * (size_t)nchunk;
    const int nwin = sched.nwin;

    void* dwin[2] = {nullptr, nullptr};
    void* dout[2] = {nullptr, nullptr};
    void *dec = nullptr, *dconj = nullptr;
    StreamRig rig("pbh_dedisperse_stream_raw");
    int& rc = rig.rc;
    double* stats = p->stream_stats;
    for (int i = 0; i < PBH_STREAM_NSTATS; ++i) stats[i] = 0.0;
    for (int b = 0; b < 2 && rc == PBH_OK; ++b) {
        if (b < nwin && (rc = dev_alloc(nullptr, &dwin[b], win_bytes + 16)) != PBH_OK) break;
        rc = dev_alloc(nullptr, &dout[b], out_bytes);
    }
    if (rc == PBH_OK) rc = dev_alloc(nullptr, &dec, row * (size_t)p->N);
    bool any_conj = false;
    if (conj_mask)
        for (int i = 0; i < p->S; ++i) any_conj |= conj_mask[i] != 0;
    if (rc == PBH_OK && any_conj) {
        rc = dev_alloc(nullptr, &dconj, (size_t)p->S);
        if (rc == PBH_OK) rig.ok(xfer_h2d(dconj, conj_mask, (size_t)p->S, p->stream), "mask copy");
    }
    rig.pin(0, const_cast<void*>(host_raw), raw_bytes);
    rig.pin(1, host_out, host_out_bytes);
    rig.create(nchunk);
    if (rc == PBH_OK) {
        rig.ok(hipStreamSynchronize(p->stream), "hipStreamSynchronize");
        rig.ok(hipEventRecord(rig.t[0], rig.s_in), "hipEventRecord");
        rig.ok(hipEventRecord(rig.t[2], rig.s_in), "hipEventRecord");
        rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.t[0], 0), "hipStreamWaitEvent");
        IoLayout io;
        if (sm) {
            io.in_layout = PBH_LAYOUT_SERIES_MAJOR;
            io.in_pitch = p->N;
        }
        for (int64_t k = 0; k < nchunk && rc == PBH_OK; ++k) {
            const pbh_host::SpanChunk c = pbh_host::span_chunk(sched, k);
            const int e = c.epoch;
            const bool head = c.head;                           // first chunk of its epoch
            const int w = c.win, b = (int)(k & 1);
            unsigned char* win = (unsigned char*)dwin[w];
            char* dst = (char*)host_out + (size_t)k * out_bytes;
            // file bytes [up_lo, up_hi) are new to the device; [base, up_lo) of a new epoch come out of the other window
            if (head && e >= 2) rig.ok(hipStreamWaitEvent(rig.s_in, rig.ev_epoch[w], 0), "hipStreamWaitEvent");
            if (c.up_hi > c.up_lo) {
                rig.ok(hipMemcpyAsync(win + (c.up_lo - c.base), (const char*)host_raw + c.up_lo, c.up_hi - c.up_lo, hipMemcpyHostToDevice, rig.s_in),
                       "hipMemcpyAsync H2D");
                stats[0] += (double)(c.up_hi - c.up_lo);
            }
            rig.ok(hipEventRecord(rig.ev_in[b], rig.s_in), "hipEventRecord");
            rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.ev_in[b], 0), "hipStreamWaitEvent");
            if (k >= 2) rig.ok(hipStreamWaitEvent(rig.s_cmp, rig.ev_out[b], 0), "hipStreamWaitEvent");  // out[b] downloaded
            rig.ok(hipEventRecord(rig.kev[(size_t)(2 * k)], rig.s_cmp), "hipEventRecord");
            if (head && k > 0) {
                if (c.handover) {
                    rig.ok(hipMemcpyAsync(win, (const unsigned char*)dwin[w ^ 1] + c.ho_src, c.ho_bytes, hipMemcpyDeviceToDevice, rig.s_cmp),
                           "hipMemcpyAsync D2D");
                    stats[7] += (double)c.ho_bytes;
                }
                rig.ok(hipEventRecord(rig.ev_epoch[w ^ 1], rig.s_cmp), "hipEventRecord");
            }
            if (rc == PBH_OK)
                rc = decode_launch(win, (int64_t)c.base, L, first + k * hop, p->N, p->nchan, p->npol, (const unsigned char*)dconj, scale, dec,
                                   sm ? PBH_LAYOUT_SERIES_MAJOR : PBH_LAYOUT_SAMPLE_MAJOR, p->N, rig.s_cmp);
            if (rc == PBH_OK) {
                const DetectTail tail = stream_tail(p, dout[b]);
                auto steps = build_steps(p, (const cf*)dec, tail.out ? nullptr : (cf*)dout[b], tail, io);
                rc = run_steps(steps, rig.s_cmp);
            }
            rig.ok(hipEventRecord(rig.kev[(size_t)(2 * k + 1)], rig.s_cmp), "hipEventRecord");
            rig.ok(hipEventRecord(rig.ev_cmp[b], rig.s_cmp), "hipEventRecord");
            rig.ok(hipStreamWaitEvent(rig.s_out, rig.ev_cmp[b], 0), "hipStreamWaitEvent");
            if (k == 0) rig.ok(hipEventRecord(rig.t[4], rig.s_out), "hipEventRecord");
            rig.ok(hipMemcpyAsync(dst, dout[b], out_bytes, hipMemcpyDeviceToHost, rig.s_out), "hipMemcpyAsync D2H");
            stats[1] += (double)out_bytes;
            rig.ok(hipEventRecord(rig.ev_out[b], rig.s_out), "hipEventRecord");
        }
        rig.ok(hipEventRecord(rig.t[3], rig.s_in), "hipEventRecord");
        rig.ok(hipEventRecord(rig.t[5], rig.s_out), "hipEventRecord");
        rig.finish(stats, ms_total);
        stats[6] = (double)nchunk;
    }
    for (int b = 0; b < 2; ++b) {
        if (dwin[b]) (void)hipFree(dwin[b]);
        if (dout[b]) (void)hipFree(dout[b]);
    }
    if (dec) (void)hipFree(dec);
    if (dconj) (void)hipFree(dconj);
    if (rc == PBH_OK && nchunk_out) *nchunk_out = nchunk;
    return rc;
#endif
}

}  // extern "C"
}  // namespace PBH_NS
