/*
 * pbhip.h -- C ABI of libpbhip.so: the MI355X (gfx950) coherent-dedispersion hot path.
 *
 * The reference (theXYZT/pulsarbat) is pure Python and has no FFI/plugin registry; the
 * drop-in boundary is its Python call surface.  Each entry point below replaces one
 * numpy/scipy expression of the reference (cited file:line, relative to the reference
 * root) and is what a binding for that expression would call.  INTEGRATION.md shows the
 * ctypes stub a pulsarbat maintainer would add.
 *
 * Conventions
 *  - plain C: pointers and sizes only, no C++/torch types.
 *  - every function returns PBH_OK (0) or a negative pbh_status; it never throws and never
 *    aborts.  pbh_last_error() returns a thread-local message for the last failure.
 *  - arrays are C-contiguous, time (sample) axis first: (nsample, nchan, npol) complex64 or
 *    complex128 (the plan's dtype) exactly as BasebandSignal.data (pulsarbat/core.py:704-764);
 *    parameter names say c64/f32 for the common case.
 *  - a pointer's residency is given by a pbh_loc argument.  Host buffers are borrowed for
 *    the duration of the call.  Device buffers must live on the plan's device.
 *  - a plan is not re-entrant (one in-flight call per plan); distinct plans may be used from
 *    distinct threads.  All work of a plan is enqueued on its stream (default: the null
 *    stream; pbh_plan_set_stream installs a caller stream, e.g. torch's current stream).
 *    Calls with host buffers synchronise before returning; calls with device-only buffers
 *    are asynchronous with respect to the host.
 */
#ifndef PBHIP_H
#define PBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbh_plan pbh_plan;

typedef enum {
    PBH_OK = 0,
    PBH_ERR_INVALID = -1,     /* bad argument (NULL, non-positive size, bad enum)            */
    PBH_ERR_UNSUPPORTED = -2, /* valid request this build cannot serve (e.g. complex128 data)    */
    PBH_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString)    */
    PBH_ERR_NOMEM = -4,       /* device or host allocation failed                             */
    PBH_ERR_STATE = -5        /* call sequence error (e.g. dedisperse before a chirp is set)  */
} pbh_status;

typedef enum { PBH_HOST = 0, PBH_DEVICE = 1 } pbh_loc;

/* complex64: float32 arithmetic; complex128: float64 arithmetic -- as scipy.fft does for each
 * (dtype in = dtype out: reference tests/test_fft.py:53-54; both accepted: pulsarbat/core.py:742).
 * Detection outputs follow: float32 / float64 (tests/test_radio_signal.py:142-172).  The chirp is
 * complex64 for both (the reference rounds it: dedispersion.py:23).                              */
typedef enum { PBH_C64 = 0, PBH_C128 = 1 } pbh_dtype;

/* Detection modes (pulsarbat/core.py:766-774, 930-966). Output float32. */
typedef enum {
    PBH_DETECT_INTENSITY = 0,       /* re^2+im^2 per element: (n, nchan, npol)   core.py:773      */
    PBH_DETECT_STOKES_I = 1,        /* |A|^2+|B|^2:           (n, nchan)         core.py:948/960  */
    PBH_DETECT_STOKES_LINEAR = 2,   /* I,Q,U,V linear basis:  (n, nchan, 4)      core.py:941-951  */
    PBH_DETECT_STOKES_CIRCULAR = 3  /* I,Q,U,V circular:      (n, nchan, 4)      core.py:953-963  */
} pbh_detect_mode;

/* Pass-structure choices (pbh_plan_set_variant).  See DESIGN.md "Kernels". */
typedef enum {
    PBH_VARIANT_AUTO = 0,
    PBH_VARIANT_PLANAR5 = 1, /* de-interleave, column FFT, fused row pass, column IFFT, re-interleave+crop */
    PBH_VARIANT_DIRECT3 = 2, /* column FFT straight from the interleaved block, fused row pass,
                                column IFFT straight into the cropped interleaved output */
    PBH_VARIANT_BLOCK3 = 3   /* as DIRECT3 with column tiles of 4 series x (F/4) time columns: 32-byte
                                pieces on both sides, line-sharing tiles co-located on one XCD        */
} pbh_variant;

typedef struct {
    int64_t nsample, crop_start, crop_stop;
    int32_t nchan, npol, device;
    int32_t n1, n2;             /* nsample = n1 * n2 (n1 = 1: single-pass plan)              */
    int32_t variant;            /* resolved pbh_variant                                       */
    int32_t nkernel;            /* kernels per pbh_dedisperse call                            */
    int64_t workspace_bytes;    /* device memory owned by the plan (workspace+chirp+tables)   */
    double  alg_bytes_per_sample; /* SURVEY.md 8(d) accounting figure for this plan           */
} pbh_plan_info_t;

#define PBH_MAX_KERNELS 16

/* ---- library ---------------------------------------------------------------------------- */
int pbh_device_count(void);
const char* pbh_last_error(void);
const char* pbh_version(void);

/* ---- plan ------------------------------------------------------------------------------- */
/* One plan = one (nsample, nchan, npol) block geometry on one device + its chirp.  crop_start /
 * crop_stop are the reference's start/stop (pulsarbat/transforms/dedispersion.py:130-131); the
 * crop is fused into the last kernel, so only stop-start rows are written.
 * nsample: any length in [2, 2^28] (powers of two) / [2, 2^27] (others), as scipy.fft accepts any (pulsarbat/fft.py:36-38).
 * Powers of two, m * 2^k (m = 3, 5, 7) and 7-smooth lengths -- what pulsarbat.utils.next_fast_len / prev_fast_len return
 * (utils.py:68-130) -- are transformed directly (pbh_plan_info: nsample = n1 * n2); any other length runs one padded
 * convolution of about twice the length (n1 = 1, n2 = nsample), exact.                                                  */
int pbh_plan_create(pbh_plan** out, int device, int64_t nsample, int nchan, int npol,
                    int dtype, int64_t crop_start, int64_t crop_stop);
int pbh_plan_destroy(pbh_plan* plan);
int pbh_plan_set_stream(pbh_plan* plan, void* hip_stream);
int pbh_plan_set_variant(pbh_plan* plan, int variant);
int pbh_plan_info(const pbh_plan* plan, pbh_plan_info_t* info);

/* Placement query (no reference counterpart; measurement-driven).  On MI355X every large device allocation belongs
 * to a "class" (runs of 8 / 32 GiB of the allocator's heap), and a pass that streams from one allocation into another of the SAME
 * class is ~5 % slower.  On its first call a plan looks for a second work buffer of another class than its first one, the
 * caller's input and the caller's output (reading them only), so nothing needs to be done by the host; this entry point merely
 * reports: *cls = 0 when the `bytes` at `dev_ptr` are of the class of the plan's first work buffer, 1 when of that of the second,
 * -1 otherwise or when it cannot be told (buffers under 1 GiB, plans that run in one work buffer).  Costs one timed 1-GiB device
 * copy (reads dev_ptr, writes plan scratch) the first time a pointer is seen; synchronises the plan's stream.             */
int pbh_plan_buffer_class(pbh_plan* plan, const void* dev_ptr, int64_t bytes, int* cls);

/* ---- chirp (transfer function) ------------------------------------------------------------- */
/* Replaces _transfer_function for every channel of a signal, i.e. DispersionMeasure.
 * chirp_from_signal (dedispersion.py:19-23, 59-75): float64 phase
 *   f = chan_freq + fftfreq(nsample, dt);  phi = coeff * f * (1/ref - 1/f)^2  [cycles]
 *   chirp = complex64(exp(-2 pi i phi))
 * with coeff_hz = dispersion_constant * DM in s*Hz^2 (= DM / 2.41e-4 * 1e12).  The result stays
 * device-resident inside the plan (in the plan's internal order).                                 */
int pbh_chirp_generate(pbh_plan* plan, double coeff_hz, double dt_s,
                       const double* chan_freq_hz /* [nchan] */, double ref_freq_hz);
/* User-supplied chirp= of coherent_dedispersion (dedispersion.py:121-124): (nsample, nchan) c64. */
int pbh_chirp_upload(pbh_plan* plan, const void* chirp_c64, int loc);
/* Same with the chirp's dtype stated: PBH_C64, or PBH_C128 for a complex128 plan.  The reference multiplies by
 * whatever array it is handed (`fft(z.data) * chirp`, dedispersion.py:124-125), so a complex128 chirp keeps its
 * precision there; the host promotes the data dtype the way numpy's product would.  A chirp that differs between
 * polarisations is a plan over nchan*npol "channels" with npol = 1 (the host reshapes).                       */
int pbh_chirp_upload_as(pbh_plan* plan, const void* chirp, int chirp_dtype, int loc);
/* The plan's chirp in natural order, (nsample, nchan) c64: what chirp_from_signal returns.       */
int pbh_chirp_download(pbh_plan* plan, void* chirp_c64, int loc);
/* Stand-alone DispersionMeasure.chirp_function (dedispersion.py:44-57): one channel, (nsample,). */
int pbh_chirp_function(int device, void* hip_stream, double coeff_hz, int64_t nsample, double dt_s,
                       double center_freq_hz, double ref_freq_hz, void* chirp_c64, int loc);

/* Transfer functions of the sibling transforms that share the FFT * H * IFFT skeleton
 * (pulsarbat/transforms/transforms.py): mode 0 = time_shift's phase ramp complex64(exp(-2 pi i shift_c f_k)),
 * f = fftfreq(nsample, 1) (transforms.py:266-270), arg = shift in samples per channel; mode 1 = freq_shift's
 * out-of-band mask in fftshifted order (transforms.py:350-359), arg = ft * nsample per channel.  Then
 * mode 2 = real_to_complex's analytic-signal weights (pulsarbat/utils.py:52-57), arg ignored.  Then
 * pbh_dedisperse runs ifft(fft(x) * H) with the plan's crop.                                              */
int pbh_chirp_special(pbh_plan* plan, const double* arg /* [nchan] */, int mode);
/* freq_shift's mixer (transforms.py:346): out[n, s] = in[n, s] * exp(2 pi i ft[s] n); device arrays (n, s). */
int pbh_mix(int device, void* hip_stream, int dtype, const void* in_dev, void* out_dev, int64_t nsample,
            int nseries, const double* ft /* [nseries], host */);
/* time_shift's zero fill of wrapped samples (transforms.py:274-286) on a device (nsample, nseries) array.  */
int pbh_zero_edges(int device, void* hip_stream, int dtype, void* data_dev, int64_t nsample, int nseries,
                   const double* shift /* [nseries], host */);

/* real_to_complex's tail (pulsarbat/utils.py:59-65: * exp(-i pi/2 n), then [::2]): out[m, s] = (-1)^m in[2m, s]. */
int pbh_decimate2(int device, void* hip_stream, int dtype, const void* in_dev, void* out_dev, int64_t nout,
                  int nseries);
/* DualPolarizationSignal.to_circular / to_linear (pulsarbat/core.py:882-928) on device (n, nchan, 2)
 * data: npairs = n * nchan.  to_circular: L = (X - iY)/sqrt2, R = (X + iY)/sqrt2; else the inverse.       */
int pbh_pol_basis(int device, void* hip_stream, int dtype, const void* in_dev, void* out_dev, int64_t npairs,
                  int to_circular);
/* incoherent_dedispersion's gather (pulsarbat/transforms/dedispersion.py:171): out[n, c, :] =
 * in[n + delay[c], c, :] for n < nout, on device data of any dtype with unit_words 4-byte words per
 * (sample, channel).  delay: host array of non-negative sample offsets.                                   */
int pbh_incoherent(int device, void* hip_stream, const void* in_dev, void* out_dev, int64_t nout, int nchan,
                   int unit_words, const int64_t* delay);
/* The same gather on series-major device arrays (time fastest: series q = chan*series_per_chan + j starts at
 * base + q*pitch; pitches in 4-byte words; an element of a series is unit_words = 1, 2 or 4 words): one shifted
 * contiguous copy per series, out[q][t] = in[q][t + delay[chan]].                                            */
int pbh_incoherent_series(int device, void* hip_stream, const void* in_dev, int64_t in_pitch_words, void* out_dev,
                          int64_t out_pitch_words, int64_t nout, int nchan, int series_per_chan, int unit_words,
                          const int64_t* delay);

/* pbh_fft_c2c and pbh_stft keep, per calling thread, up to two Bluestein and two multi-pass transform plans
 * (twiddle tables and a workspace the size of the data) for reuse by the next call of the same shape.
 * pbh_trim() releases the calling thread's cached plans.                                                   */
int pbh_trim(void);

/* Blocking copy between caller (host) memory and device memory, direction 0 = host->device, 1 =
 * device->host, ordered on hip_stream.  Goes through the library's own pinned bounce buffers, like every
 * PBH_HOST argument of the calls below: pageable caller memory is never handed to the HIP runtime, whose
 * cache of on-the-fly pins can outlive the caller's allocation ("host transfers").           */
int pbh_transfer(int device, void* hip_stream, void* dst, const void* src, size_t bytes, int direction);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* Replaces  x = ifft(fft(z.data, axis=0) * chirp, axis=0)[start:stop]  (dedispersion.py:125-133).
 * in : (nsample, nchan, npol) c64;  out: (crop_stop-crop_start, nchan, npol) c64.                   */
int pbh_dedisperse(pbh_plan* plan, const void* in_c64, void* out_c64, int in_loc, int out_loc);

/* The same transform on device-resident arrays whose memory layout is stated explicitly.  The
 * reference's Signal.data is duck-typed (core.py:59-97) and numpy arrays carry strides, so a
 * (nsample, nchan, npol) array need not be C-contiguous:
 *   PBH_LAYOUT_SAMPLE_MAJOR  C-contiguous (nsample, nchan, npol); pitch ignored
 *   PBH_LAYOUT_SERIES_MAJOR  time fastest: element (t, chan, pol) at base + (chan*npol + pol)*pitch + t,
 *                            i.e. strides (1, npol*pitch, pitch) in elements, pitch >= the time length
 * With a series-major end the layout pass at that end disappears (5 kernels -> 4 or 3).  For full-line
 * stores pick out_dev and out_pitch so that (out_dev - crop_start*elem) and out_pitch*elem are multiples
 * of 128 bytes (pulsarbat_amd.DeviceArray.empty_series_major does).  Multi-pass power-of-two plans only
 * (PBH_ERR_UNSUPPORTED otherwise); asynchronous on the plan's stream.                                 */
typedef enum { PBH_LAYOUT_SAMPLE_MAJOR = 0, PBH_LAYOUT_SERIES_MAJOR = 1 } pbh_layout;
int pbh_dedisperse_layout(pbh_plan* plan, const void* in_dev, int in_layout, int64_t in_pitch, void* out_dev,
                          int out_layout, int64_t out_pitch);
/* Multi-GPU: the same transform with the (nout, nchan, npol) result written as a channel slice of a wider
 * sample-major array: row t of the result goes to out_dev[t*out_row_elems + out_col_offset + (chan*npol + pol)].
 * A rank of a channel-sharded job (reference: Dask chunks over the non-time axes, core.py:332-345) passes the
 * full-band block -- its own, or a peer GPU's mapped with pbh_node_import -- with out_row_elems =
 * nchan_total*npol and out_col_offset = first_channel*npol, so the gather of Signal.compute()
 * (core.py:298-309) is done by the pipeline's last kernel.  Device-resident, C-contiguous input;
 * asynchronous on the plan's stream.  Fastest when nchan*npol is a power of two <= 128 and the
 * offset / row length are even (16-byte stores); every other geometry is served through one extra pass. */
int pbh_dedisperse_slice(pbh_plan* plan, const void* in_dev, void* out_dev, int64_t out_row_elems,
                         int64_t out_col_offset);
/* The same with the destination's ROWS split over nparts buffers: part i receives output rows
 * [part_row[i], part_row[i+1]) (part_row[0] = 0, part_row[nparts] = stop - start) starting at part_dev[i].  A
 * destination block that peers map must be built from allocations of at most PBH_NODE_MAX_BYTES (larger ones
 * hang in the runtime's IPC mapping on this ROCm stack): the pipeline runs once and only its last pass runs
 * once per part.                                                                                            */
int pbh_dedisperse_slices(pbh_plan* plan, const void* in_dev, int nparts, void* const* part_dev,
                          const int64_t* part_row /* [nparts + 1] */, int64_t out_row_elems, int64_t out_col_offset);

/* Sharing a device buffer between the ranks of one node (one process per GPU).  The destination rank
 * allocates with pbh_node_alloc (a whole device allocation, hence exportable; at most PBH_NODE_MAX_BYTES each:
 * 2040 MiB is the largest size seen to map into a peer process, 2056 MiB never returned from the peer's
 * hipIpcOpenMemHandle -- observed between two processes on ONE device, the cross-device case is unobserved;
 * a larger block is several allocations, see pbh_dedisperse_slices), exports a 64-byte handle,
 * ships it to its peers by any host channel (the Python host uses torch.distributed), and each peer maps
 * it with pbh_node_import (peer access over xGMI is enabled by the mapping) and writes into it with
 * pbh_dedisperse_slice.  The exporter must keep the buffer alive until every importer has called
 * pbh_node_release; a host-side barrier after the writers' streams have drained makes the data visible
 * to the owner.  Replaces the reference's in-process gather of chunk results (core.py:298-309).          */
/* freq_shift (pulsarbat/transforms/transforms.py:337-361): out = IFFT(H * FFT(x[n, s] * exp(2 pi i ft[s] n))) with the
 * plan's filter H (pbh_chirp_special mode 1: the out-of-band mask).  ft: host array of nchan*npol shifts in cycles per
 * sample.  Device-resident C-contiguous in / out; the mixer is folded into the plan's de-interleave pass where it has
 * one (no copy of the caller's data, no separate mixing pass), else it is one out-of-place pass.                    */
int pbh_dedisperse_mix(pbh_plan* plan, const void* in_dev, void* out_dev, const double* ft);
/* pbh_place: 2-D copy between sample-major device arrays (nrow rows of ncol complex elements, row pitches in
 * elements) -- the push of a rank's channel slice into a peer's full-band block in the all-gather form.       */
int pbh_place(int device, void* hip_stream, int dtype, const void* src_dev, int64_t src_row_elems, void* dst_dev,
              int64_t dst_row_elems, int64_t nrow, int64_t ncol);
typedef struct { unsigned char bytes[64]; } pbh_ipc_handle_t;
#define PBH_NODE_MAX_BYTES (2040ull << 20)
int pbh_node_alloc(int device, size_t bytes, void** dev_ptr);
/* The same sharing without the size limit: a physical allocation (hipMemCreate) exported as a POSIX file descriptor and
 * mapped by every process that holds the descriptor (tools/micro/vmmprobe.hip: a single 3-GiB allocation shared between
 * two processes, where hipIpcOpenMemHandle hangs beyond 2 GiB).  pbh_node_share_alloc returns the owner's mapping and the
 * descriptor (the caller sends it to its peers over a Unix socket, SCM_RIGHTS, and closes it); pbh_node_share_import maps
 * a received descriptor (`bytes` as passed to the owner's call); pbh_node_share_free undoes either.  A destination block
 * is then ONE contiguous buffer: no row-chunks, no joining copy.                                                       */
int pbh_node_share_alloc(int device, size_t bytes, void** dev_ptr, int* fd);
int pbh_node_share_import(int device, int fd, size_t bytes, void** dev_ptr);
int pbh_node_share_free(int device, void* dev_ptr);
int pbh_node_free(int device, void* dev_ptr);
int pbh_node_export(int device, void* dev_ptr, pbh_ipc_handle_t* handle);
int pbh_node_import(int device, const pbh_ipc_handle_t* handle, void** dev_ptr);
int pbh_node_release(int device, void* dev_ptr);

/* pbh_dedisperse_detect (below) for a device-resident input with a stated layout; out is the C-contiguous
 * detected array.  A series-major input needs a fused tail (nscrunch % 64 == 0, or nscrunch == 1 where the last
 * layout pass detects; PBH_ERR_UNSUPPORTED otherwise): 4 kernels.                                           */
int pbh_dedisperse_detect_layout(pbh_plan* plan, const void* in_dev, int in_layout, int64_t in_pitch,
                                 void* out_f32_dev, int nscrunch, int mode);

/* Same, followed by detection (core.py:766-774 / 930-966) and an nscrunch-fold sum over time of the
 * cropped samples (tail dropped): out is float32 (nout, nchan[, npol|4]), nout = (stop-start)/nscrunch.
 * The dedispersed voltages are not stored: with nscrunch % 64 == 0 a read pass over the planar workspace detects
 * them, and for |z|^2 / Stokes I at nsample = 2^20 ... 2^24 (64- to 1024-row column tiles, nscrunch a divisor of 2^14) the inverse
 * column pass itself does (4 KiB of partial sums per tile instead of 128 KiB of voltages; PBH_DETECT_COLQ=0 to
 * compare).  The plan then holds S * nsample / 16 floats of partial sums beside its workspace.  With nscrunch == 1
 * (to_intensity / to_stokes at full time resolution) the last layout pass of a multi-pass plan with an even
 * number of series writes the detected rows instead of the voltages (PBH_DETECT_REINT=0 to compare).             */
int pbh_dedisperse_detect(pbh_plan* plan, const void* in_c64, void* out_f32, int nscrunch, int mode,
                          int in_loc, int out_loc);

/* Streaming overlap-save over a long HOST-resident block (BASELINE configs[3]): chunks of the plan's
 * nsample rows every hop = crop_stop - crop_start rows; chunk k is one reference call on
 * in[k*hop : k*hop + nsample] (dedispersion.py:81-133) and fills out rows [k*hop, (k+1)*hop) --
 * concatenate() of the chunk results (transforms.py:59-148).  Every input row crosses PCIe once: the rows
 * consecutive chunks share stay in a device window (the stream's samples exist once,
 * transforms.py:101-110; readers are offset-addressed, readers/_base.py:298-333).  H2D / kernels / D2H
 * run on three HIP streams.  in : (total_nsample, nchan, npol) host;  out: (nchunk*hop, nchan, npol)
 * host, nchunk = (total_nsample - nsample) / hop + 1 (returned).  ms_total (optional): HIP-event time.
 * Both host ranges are page-locked for the duration of the call; a range that cannot be page-locked (and
 * is not pinned by the caller already) fails with PBH_ERR_HIP -- there is no pageable-memory fallback.   */
int pbh_dedisperse_stream(pbh_plan* plan, const void* host_in, int64_t total_nsample, void* host_out,
                          int64_t* nchunk, float* ms_total);

/* Figures of the plan's last streaming call (pbh_dedisperse_stream / _raw); out[i] for i < n:
 *   0 bytes host -> device   1 bytes device -> host   2 ms the upload stream was busy (first copy .. last)
 *   3 ms the download stream was busy   4 ms of kernels (sum over chunks, decode and window slides included)
 *   5 ms total   6 chunks   7 bytes moved device -> device between windows                               */
#define PBH_STREAM_NSTATS 8
int pbh_stream_stats(const pbh_plan* plan, double* out, int n);

/* Detected output for the streaming calls: after pbh_plan_stream_detect(plan, mode, nscrunch) every chunk of
 * pbh_dedisperse_stream / _raw ends in the fused detect tail of pbh_dedisperse_detect and host_out receives float32 rows
 * (nchunk * hop / nscrunch, nchan[, npol | 4]) instead of voltages: the concatenation of the chunks' detected, scrunched
 * valid regions = detect + scrunch of the dedispersed stream (to_intensity / to_stokes, core.py:766-774, 930-966, then the
 * nscrunch-fold sum).  Needs hop = crop_stop - crop_start to be a multiple of nscrunch (PBH_ERR_INVALID otherwise) and a
 * fused tail for the plan (multi-pass plans; nscrunch % 64 == 0, or 1: PBH_ERR_UNSUPPORTED otherwise).  mode < 0 switches
 * back to voltages.  The download shrinks by 2 * nscrunch * npol / elements per row: the stream is then bound by its upload. */
int pbh_plan_stream_detect(pbh_plan* plan, int mode, int nscrunch);

/* ---- reader-side decode ----------------------------------------------------------------------- */
/* Replaces the host post-processing of the reference's baseband readers
 * (pulsarbat/readers/_baseband_readers.py:136-153 `_read_baseband`: per-series sideband conjugation and
 * `astype(complex64)`; :223-226 GUPPIRawReader / :268-275 DADAStokesReader `_read_array`: axis transposes and
 * channel flip) together with the integer -> float unpacking that precedes it, so that the raw payload bytes
 * are what crosses PCIe.  The byte range `raw` (host or device, `raw_bytes` long) is a sequence of blocks
 * `blk_stride` bytes apart, each `hdr_bytes` of header followed by a payload holding `blk_samples` time
 * samples; inside a payload, element (t, chan, pol) is element number
 * elem0 + t*stride_t + chan*stride_c + pol*stride_p (strides in elements, may be negative), an element being
 * `ncomp` components (1 real, 2 complex: re, im) of `nbits` bits each, low bits first within a byte.
 *   nbits 8, code 0: two's complement (DADA, GUPPI);  code 1: offset binary, v - 128 (VDIF)
 *   nbits 4: offset binary, v - 8 (VDIF; pass scale = 1/2.95 for the usual normalisation)
 *   nbits 2: 4-level code 0..3 -> -3.316505, -1, +1, +3.316505 (baseband's OPTIMAL_2BIT_HIGH)
 * Output (device): float32 (ncomp 1) or complex64 (ncomp 2) array of logical shape (nsample, nchan, npol) in
 * `out_layout` (pbh_layout; out_pitch in elements for PBH_LAYOUT_SERIES_MAJOR), values times `scale`, imaginary
 * part negated for the series whose entry in conj_mask (host, nchan*npol bytes, may be NULL) is non-zero.
 * `first` is the index of the first wanted sample counted from the first sample of the first block in `raw`. */
typedef struct {
    int nbits;
    int ncomp;
    int code;
    int64_t blk_samples;
    int64_t blk_stride;
    int64_t hdr_bytes;
    int64_t elem0;
    int64_t stride_t, stride_c, stride_p;
} pbh_raw_layout_t;
int pbh_decode(int device, void* hip_stream, const void* raw, size_t raw_bytes, int raw_loc,
               const pbh_raw_layout_t* layout, int64_t first, int64_t nsample, int nchan, int npol,
               const unsigned char* conj_mask, float scale, void* out_dev, int out_layout, int64_t out_pitch);

/* pbh_dedisperse_stream fed with RAW payload bytes (pbh_raw_layout_t, complex samples): chunk k decodes samples
 * [first + k*hop, first + k*hop + nsample) of the payload stream on the device (pbh_decode's unpacking, conjugation mask and
 * scale) and dedisperses them, so that 2 bytes per 8-bit complex sample cross PCIe instead of 8.  The result
 * equals pbh_dedisperse_stream on the decoded array.  complex64 plans only.                                */
int pbh_dedisperse_stream_raw(pbh_plan* plan, const void* host_raw, size_t raw_bytes,
                              const pbh_raw_layout_t* layout, int64_t first, int64_t total_nsample,
                              const unsigned char* conj_mask, float scale, void* host_out,
                              int64_t* nchunk, float* ms_total);

/* Conversion of a device (nsample, nseries) complex array between the two device layouts (what
 * DeviceArray.to_series_major() / .contiguous() do; the reference has one layout, pulsarbat/core.py:59-97): one
 * transposing pass with the pipeline's layout kernels.                                                     */
int pbh_relayout(int device, void* hip_stream, int dtype, const void* in_dev, int in_layout, int64_t in_pitch,
                 void* out_dev, int out_layout, int64_t out_pitch, int64_t nsample, int nseries);

/* Stand-alone detection of device- or host-resident baseband data (to_intensity / to_stokes).       */
int pbh_detect(int device, void* hip_stream, int dtype, const void* in, void* out, int64_t nsample,
               int nchan, int npol, int mode, int nscrunch, int in_loc, int out_loc);

/* Backs pb.fft.fft / pb.fft.ifft for device arrays (pulsarbat/fft.py:30-48 -> scipy.fft.fft/ifft,
 * norm=None): c2c along axis 0 of a C-contiguous (n, batch) array.  n up to one tile: one kernel; 2^k and
 * m * 2^k (m = 3, 5, 7) beyond it: multi-pass transform + natural-order output pass; other n: Bluestein.  */
int pbh_fft_c2c(int device, void* hip_stream, int dtype, const void* in, void* out, int64_t n,
                int64_t batch, int inverse, int in_loc, int out_loc);

/* contrib.stft / istft (pulsarbat/contrib/misc.py:17-93; boxcar window, no overlap, nfft = nperseg): the
 * critically sampled channeliser.  stft : in (nseg*nperseg, nchan, inner) time-ordered -> out
 * (nseg, nchan*nperseg, inner) with out[g, c*n + (k + n/2) % n, e] = FFT_k(in[g*n + t, c, e]) / n
 * (fft, fftshift, /nperseg of misc.py:47-52).  inverse != 0 is istft, its exact inverse (misc.py:83-91).
 * Any nperseg >= 1; powers of two up to one tile run a single fused kernel.                               */
int pbh_stft(int device, void* hip_stream, int dtype, const void* in, void* out, int64_t nseg, int nperseg,
             int nchan, int inner, int inverse, int in_loc, int out_loc);

/* contrib.stft followed by coherent_dedispersion -- the reference's typical channelise-then-dedisperse pipeline
 * (pulsarbat/contrib/misc.py:41-55, then transforms/dedispersion.py:118-133) -- in one call.  `plan` is the
 * dedispersion plan of the CHANNELISED block: nsample = nseg, nchan = nchan_in*nperseg, npol = inner elements per
 * channel, chirp set for the channelised signal's channel frequencies; in_dev is the device-resident C-contiguous
 * (nseg*nperseg, nchan_in, inner) block the channeliser reads.  out: (stop-start, nchan_in*nperseg, inner), sample-major
 * or series-major as in pbh_dedisperse_layout.  Where the geometry allows (complex64, nperseg = 2^m in [32, 1024],
 * nseg beyond one tile) the channeliser writes series-major into the plan's work buffer and the dedispersion starts
 * at its column pass: the channelised block makes one HBM round trip less.  Asynchronous on the plan's stream.     */
int pbh_stft_dedisperse(pbh_plan* plan, const void* in_dev, int nperseg, int nchan_in, void* out_dev, int out_layout,
                        int64_t out_pitch);

/* The way back: coherent_dedispersion followed by contrib.istft (transforms/dedispersion.py:118-133, then
 * pulsarbat/contrib/misc.py:58-93) in one call.  `plan` is again the dedispersion plan of the CHANNELISED block
 * (nsample = nseg, nchan = nchan_out*nperseg, npol = inner); in_dev the device-resident channelised block, sample-major
 * or series-major (in_layout, in_pitch as in pbh_dedisperse_layout); out_dev the C-contiguous
 * ((stop-start)*nperseg, nchan_out, inner) time series.  Where the geometry allows (complex64, nperseg = 2^m in
 * [32, 1024], nseg beyond one tile, at least 64-byte output runs per tile) the dedispersion's last column pass leaves its
 * cropped result series-major in the plan's staging buffer and the synthesis kernel reads that: the channelised result is
 * neither re-interleaved nor read back.  Other geometries run the two steps.  Asynchronous on the plan's stream.      */
int pbh_dedisperse_istft(pbh_plan* plan, const void* in_dev, int in_layout, int64_t in_pitch, int nperseg, int nchan_out,
                         void* out_dev);

/* ---- measurement --------------------------------------------------------------------------------- */
/* Runs the plan's kernel sequence `iters` times on device-resident in/out with hipEvents between the
 * kernels (on the plan's stream) and returns the mean milliseconds of each kernel.                    */
int pbh_plan_profile(pbh_plan* plan, const void* in_dev, void* out_dev, int iters,
                     float* ms_per_kernel /* [PBH_MAX_KERNELS] */, int* nkernel,
                     const char** names /* [PBH_MAX_KERNELS], static strings */);

/* utils.real_to_complex (pulsarbat/utils.py:15-65) of device-resident float32 data: in (nreal, nseries) float32 C order,
 * out (nreal/2, nseries) complex64 C order.  Runs as a HALF-LENGTH complex transform (the real series, time fastest, is
 * the complex series x[2m] + i x[2m+1]; one mirror pass turns its spectrum into the decimated analytic signal's): two
 * transforms of nreal/2 points instead of two of nreal points on a complex copy.  nreal/2 must be a power of two in
 * [2^15, 2^24]; other sizes return PBH_ERR_UNSUPPORTED (the host then takes the full-length route through a filter plan,
 * pbh_chirp_special mode 2 + pbh_decimate2).  Asynchronous on hip_stream.                                             */
int pbh_real_to_complex(int device, void* hip_stream, const void* in_dev, void* out_dev, int64_t nreal, int nseries);

/* Device-to-device streaming copy of `bytes` (float4 per lane): the achievable-HBM reference number. */
int pbh_copy_bench(int device, int64_t bytes, int iters, float* ms_mean);
/* The same yardstick with a mode: 0 = copy between two buffers of `bytes` (= pbh_copy_bench; for 1 GiB and more the destination
 * is the fastest of up to twelve allocations: a copy between two allocations of one class is 4-5 % slower, DESIGN.md 6d d),
 * 1 = read-modify-write of ONE buffer in place -- the ceiling of the three middle passes of the five-pass schedule, which update
 * the planar work buffer where it stands (no reference counterpart: measurement support for bench.py's path_roofline).      */
int pbh_stream_bench(int device, int64_t bytes, int iters, int mode, float* ms_mean);

#ifdef __cplusplus
}
#endif
#endif /* PBHIP_H */
