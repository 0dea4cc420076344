"""ctypes binding of libpbhip.so (the C ABI declared in include/pbhip.h).

This is the only module that talks to the native library.  There is no CPU
fallback anywhere behind it: if the library cannot be loaded, or no HIP device
is present, every hot-path call raises :class:`HipUnavailableError`.
"""

import ctypes as C
import os
import threading
from collections import OrderedDict

import numpy as np

from . import _build

__all__ = ["HipError", "HipUnavailableError", "lib", "available", "Plan", "detect", "decode", "trim", "relayout", "fft_c2c",
           "chirp_function", "copy_bench", "stream_bench"]

HOST, DEVICE = 0, 1
DTYPES = {np.dtype(np.complex64): 0, np.dtype(np.complex128): 1}   # pbh_dtype
REAL_OF = {np.dtype(np.complex64): np.dtype(np.float32), np.dtype(np.complex128): np.dtype(np.float64)}


def _dtype_code(dt):
    try:
        return DTYPES[np.dtype(dt)]
    except KeyError:
        raise TypeError(f"the HIP path takes complex64 or complex128 data, got {np.dtype(dt)}")

DETECT_MODES = {"intensity": 0, "I": 1, "stokes_i": 1, "linear": 2, "circular": 3}
VARIANTS = {"auto": 0, "planar5": 1, "direct3": 2, "block3": 3}
MAX_KERNELS = 16


class HipError(RuntimeError):
    """A libpbhip call failed (message from pbh_last_error)."""


class HipUnavailableError(HipError):
    """The HIP library or a HIP device is missing; the hot path cannot run."""


class _PlanInfo(C.Structure):
    _fields_ = [("nsample", C.c_int64), ("crop_start", C.c_int64), ("crop_stop", C.c_int64),
                ("nchan", C.c_int32), ("npol", C.c_int32), ("device", C.c_int32),
                ("n1", C.c_int32), ("n2", C.c_int32), ("variant", C.c_int32),
                ("nkernel", C.c_int32), ("workspace_bytes", C.c_int64),
                ("alg_bytes_per_sample", C.c_double)]


class _RawLayout(C.Structure):
    _fields_ = [("nbits", C.c_int), ("ncomp", C.c_int), ("code", C.c_int), ("blk_samples", C.c_int64),
                ("blk_stride", C.c_int64), ("hdr_bytes", C.c_int64), ("elem0", C.c_int64),
                ("stride_t", C.c_int64), ("stride_c", C.c_int64), ("stride_p", C.c_int64)]


# name -> (restype, argtypes); every symbol include/pbhip.h declares
SIGNATURES = {
    "pbh_device_count": (C.c_int, []),
    "pbh_last_error": (C.c_char_p, []),
    "pbh_version": (C.c_char_p, []),
    "pbh_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int,
                                  C.c_int64, C.c_int64]),
    "pbh_plan_destroy": (C.c_int, [C.c_void_p]),
    "pbh_plan_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbh_plan_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "pbh_plan_info": (C.c_int, [C.c_void_p, C.POINTER(_PlanInfo)]),
    "pbh_plan_buffer_class": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int)]),
    "pbh_chirp_generate": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_double]),
    "pbh_chirp_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "pbh_chirp_upload_as": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pbh_chirp_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "pbh_chirp_special": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "pbh_mix": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                          C.POINTER(C.c_double)]),
    "pbh_zero_edges": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    "pbh_decimate2": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "pbh_pol_basis": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "pbh_incoherent": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.POINTER(C.c_int64)]),
    "pbh_incoherent_series": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                        C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "pbh_chirp_function": (C.c_int, [C.c_int, C.c_void_p, C.c_double, C.c_int64, C.c_double, C.c_double,
                                     C.c_double, C.c_void_p, C.c_int]),
    "pbh_trim": (C.c_int, []),
    "pbh_relayout": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64,
                               C.c_int64, C.c_int]),
    "pbh_transfer": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "pbh_decode": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(_RawLayout), C.c_int64,
                             C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int64]),
    "pbh_dedisperse_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64]),
    "pbh_dedisperse_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "pbh_dedisperse_slices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int64,
                                        C.c_int64]),
    "pbh_dedisperse_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "pbh_place": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "pbh_node_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pbh_node_free": (C.c_int, [C.c_int, C.c_void_p]),
    "pbh_node_export": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "pbh_node_import": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pbh_node_release": (C.c_int, [C.c_int, C.c_void_p]),
    "pbh_dedisperse_detect_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int]),
    "pbh_dedisperse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pbh_dedisperse_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pbh_dedisperse_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_float)]),
    "pbh_dedisperse_istft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pbh_node_share_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "pbh_node_share_import": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pbh_node_share_free": (C.c_int, [C.c_int, C.c_void_p]),
    "pbh_stream_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "pbh_plan_stream_detect": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pbh_dedisperse_stream_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(_RawLayout), C.c_int64, C.c_int64,
                                            C.c_void_p, C.c_float, C.c_void_p, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_float)]),
    "pbh_detect": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_int, C.c_int]),
    "pbh_fft_c2c": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                              C.c_int, C.c_int]),
    "pbh_stft": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                           C.c_int, C.c_int, C.c_int]),
    "pbh_stft_dedisperse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64]),
    "pbh_plan_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float),
                                   C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "pbh_real_to_complex": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "pbh_copy_bench": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
    "pbh_stream_bench": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """The loaded library (ctypes.CDLL releases the GIL during calls)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = os.environ.get("PBHIP_LIBRARY", _build.LIB)
                if not os.path.exists(path):
                    raise HipUnavailableError(
                        f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(needs hipcc); pulsarbat_amd has no CPU fallback for the hot path")
                # torch bundles its own libamdhip64.so.7; it must be loaded first so that
                # libpbhip.so binds to that same runtime by SONAME (two HIP runtimes in one
                # process leave the second without devices).
                import torch  # noqa: F401
                try:
                    handle = C.CDLL(path)
                except OSError as exc:
                    raise HipUnavailableError(f"cannot load {path}: {exc}") from exc
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def available():
    """True when the library loads and sees at least one HIP device."""
    try:
        return lib().pbh_device_count() > 0
    except HipUnavailableError:
        return False


def _require_device():
    n = lib().pbh_device_count()
    if n <= 0:
        raise HipUnavailableError("no HIP device visible; the hot path needs an MI355X (no CPU fallback)")
    return n


_ERRORS = {-1: ValueError, -2: NotImplementedError, -3: HipError, -4: MemoryError, -5: HipError}


def _check(code):
    if code != 0:
        msg = lib().pbh_last_error().decode(errors="replace")
        raise _ERRORS.get(code, HipError)(f"libpbhip: {msg} (status {code})")


def _ptr_loc(a):
    """(pointer, loc, keepalive) for a numpy array or a DeviceArray."""
    from .device import DeviceArray
    if isinstance(a, DeviceArray):
        return C.c_void_p(a.data_ptr()), DEVICE
    if isinstance(a, np.ndarray):
        if not a.flags.c_contiguous:
            raise ValueError("host arrays must be C-contiguous")
        return C.c_void_p(a.ctypes.data), HOST
    raise TypeError(f"unsupported array type {type(a)!r}")


_FILTER_PLANS = OrderedDict()
_FILTER_PLAN_CACHE_SIZE = 2


def filter_plan(nsample, nseries, lo, hi, device, dtype, shared=False):
    """Cached plan for the FFT * H * IFFT helpers (time_shift, freq_shift, real_to_complex): one filter row per
    series (nchan = nseries, npol = 1), or with ``shared`` ONE row for all series (nchan = 1, npol = nseries: the
    row pass then reads 8/nseries bytes of filter per sample instead of 8).  Creating a plan costs two allocations
    the size of the data, so the last few geometries are kept.  The caller sets H (chirp_special / chirp_upload)
    before every use."""
    key = (int(nsample), int(nseries), int(lo), int(hi), int(device), np.dtype(dtype).str, bool(shared),
           threading.get_ident())
    with _lock:
        plan = _FILTER_PLANS.pop(key, None)
    if plan is None:
        plan = (Plan(nsample, 1, nseries, lo, hi, device=device, dtype=dtype) if shared else
                Plan(nsample, nseries, 1, lo, hi, device=device, dtype=dtype))
    with _lock:
        _FILTER_PLANS[key] = plan
        stale = []
        while len(_FILTER_PLANS) > _FILTER_PLAN_CACHE_SIZE:
            stale.append(_FILTER_PLANS.popitem(last=False)[1])
    for old in stale:
        old.close()
    return plan


def relayout(x, out):
    """Copy device array ``x`` into ``out`` (same shape and complex dtype), either of them C-contiguous or series-major."""
    _require_device()
    n = x.shape[0]
    nser = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    ip = None if x.tensor.is_contiguous() else x.series_major_pitch()
    op = None if out.tensor.is_contiguous() else out.series_major_pitch()
    if (ip is None and not x.tensor.is_contiguous()) or (op is None and not out.tensor.is_contiguous()):
        raise ValueError("arrays must be C-contiguous or series-major")
    _check(lib().pbh_relayout(x.device_index, _stream_ptr(x.device_index), _dtype_code(x.dtype), C.c_void_p(x.raw_ptr()),
                              int(ip is not None), int(ip or 0), C.c_void_p(out.raw_ptr()), int(op is not None), int(op or 0),
                              int(n), nser))
    return out


def trim():
    """Free the calling thread's cached stand-alone transform plans (pbh_trim)."""
    with _lock:
        stale = list(_FILTER_PLANS.values())
        _FILTER_PLANS.clear()
    for old in stale:
        old.close()
    if _lib is not None:
        _check(lib().pbh_trim())


def transfer(device, dst_ptr, src_ptr, nbytes, to_host):
    """Blocking host<->device copy through the library's pinned bounce buffers (pbh_transfer)."""
    _require_device()
    _check(lib().pbh_transfer(int(device), _stream_ptr(int(device)), C.c_void_p(dst_ptr), C.c_void_p(src_ptr),
                              int(nbytes), 1 if to_host else 0))


def _stream_ptr(device):
    """torch's current stream on ``device`` so plan work is ordered with torch ops."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Plan:
    """One (nsample, nchan, npol) block geometry + chirp on one device (pbh_plan)."""

    def __init__(self, nsample, nchan, npol, crop_start, crop_stop, device=0, variant="auto",
                 use_torch_stream=True, dtype=np.complex64):
        _require_device()
        self._h = C.c_void_p()
        self.device = int(device)
        self.dtype = np.dtype(dtype)
        self.real_dtype = REAL_OF[self.dtype] if self.dtype in REAL_OF else None
        _check(lib().pbh_plan_create(C.byref(self._h), self.device, int(nsample), int(nchan), int(npol),
                                     _dtype_code(self.dtype), int(crop_start), int(crop_stop)))
        self.nsample, self.nchan, self.npol = int(nsample), int(nchan), int(npol)
        self.crop_start, self.crop_stop = int(crop_start), max(int(crop_stop), int(crop_start))
        self._use_torch_stream = use_torch_stream
        if variant != "auto":
            _check(lib().pbh_plan_set_variant(self._h, VARIANTS[variant]))

    def close(self):
        h = getattr(self, "_h", None)
        self._h = None
        if h is not None and h.value:
            try:
                lib().pbh_plan_destroy(h)
            except Exception:   # interpreter shutdown: module globals may be gone, the process frees the memory
                pass

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _sync_stream(self):
        if self._use_torch_stream:
            _check(lib().pbh_plan_set_stream(self._h, _stream_ptr(self.device)))

    @property
    def info(self):
        inf = _PlanInfo()
        _check(lib().pbh_plan_info(self._h, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in _PlanInfo._fields_}

    @property
    def nout(self):
        return self.crop_stop - self.crop_start

    def _empty_like_class(self, oshape, x):
        """Output array of a call.  (Round 4 placed it by allocation class here; the library now picks its second work
        buffer so that it streams fast from and to BOTH of the caller's arrays on a plan's first call -- ``ensure_work2`` in
        ``csrc/pbhip.hip`` -- and the output is taken as the allocator hands it out.)"""
        from .device import DeviceArray
        return DeviceArray.empty(oshape, self.dtype, device=self.device)

    def buffer_class(self, x):
        """Allocation class of a device array relative to the plan's two work buffers (pbh_plan_buffer_class): 0 = that of
        the first, 1 = that of the second, -1 = neither or not told apart.  A streaming pass is ~5 % slower between two
        allocations of one class; the library arranges its own buffers around the caller's (DESIGN.md 6d d)."""
        self._sync_stream()
        cls = C.c_int(-1)
        nbytes = int(np.prod(x.shape)) * np.dtype(x.dtype).itemsize
        _check(lib().pbh_plan_buffer_class(self._h, C.c_void_p(x.data_ptr()), nbytes, C.byref(cls)))
        return int(cls.value)

    def chirp_generate(self, coeff_hz, dt_s, chan_freq_hz, ref_freq_hz):
        self._sync_stream()
        freqs = np.ascontiguousarray(chan_freq_hz, dtype=np.float64)
        if freqs.shape != (self.nchan,):
            raise ValueError("chan_freq_hz must have shape (nchan,)")
        _check(lib().pbh_chirp_generate(self._h, float(coeff_hz), float(dt_s),
                                        freqs.ctypes.data_as(C.POINTER(C.c_double)), float(ref_freq_hz)))

    def chirp_special(self, arg, mode):
        """H = time_shift phase ramp (mode 0, arg = shifts in samples) or freq_shift mask (mode 1, arg = ft*N)."""
        self._sync_stream()
        a = np.ascontiguousarray(arg, dtype=np.float64)
        if a.shape != (self.nchan,):
            raise ValueError("arg must have shape (nchan,)")
        _check(lib().pbh_chirp_special(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), int(mode)))

    def chirp_upload(self, chirp):
        """chirp: (nsample, nchan) complex64, or complex128 for a complex128 plan; numpy or DeviceArray."""
        self._sync_stream()
        if tuple(chirp.shape) != (self.nsample, self.nchan) or np.dtype(chirp.dtype) not in DTYPES:
            raise ValueError(f"chirp must be complex64 / complex128 with shape {(self.nsample, self.nchan)}")
        if np.dtype(chirp.dtype) == np.complex128 and self.dtype != np.complex128:
            raise TypeError("a complex128 chirp needs a complex128 plan")
        ptr, loc = _ptr_loc(chirp)
        _check(lib().pbh_chirp_upload_as(self._h, ptr, DTYPES[np.dtype(chirp.dtype)], loc))

    def chirp_download(self, out=None):
        self._sync_stream()
        if out is None:
            out = np.empty((self.nsample, self.nchan), dtype=np.complex64)
        ptr, loc = _ptr_loc(out)
        _check(lib().pbh_chirp_download(self._h, ptr, loc))
        return out

    @property
    def supports_series_major(self):
        """True when pbh_dedisperse_layout accepts series-major ends for this plan."""
        n = self.nsample
        info = self.info
        # 3 / 5 / 7 kernels: multi-pass pipelines (7: long blocks with a stand-alone radix stage); 1 + ...: other lengths
        return (n & (n - 1)) == 0 and info["n1"] > 1 and info["nkernel"] in (3, 5, 7)

    def _check_in(self, x):
        if tuple(x.shape[:1]) != (self.nsample,) or int(np.prod(x.shape[1:])) != self.nchan * self.npol:
            raise ValueError(f"input shape {tuple(x.shape)} does not match plan "
                             f"({self.nsample}, {self.nchan}, {self.npol})")
        if x.dtype != self.dtype:
            raise TypeError(f"input must be {self.dtype} for this plan")

    def dedisperse(self, x, out=None, out_layout=None):
        """x: (nsample, nchan, npol) c64 numpy or DeviceArray -> (stop-start, ...) same container.

        A DeviceArray may be stored series-major (time fastest, ``DeviceArray.series_major_pitch``);
        the result then is too unless ``out_layout`` ("sample" | "series") says otherwise, and the
        layout passes at the series-major ends are skipped (``pbh_dedisperse_layout``)."""
        from .device import DeviceArray
        self._check_in(x)
        self._sync_stream()
        oshape = (self.nout,) + tuple(x.shape[1:])
        if isinstance(x, DeviceArray):
            in_pitch = None if x.tensor.is_contiguous() else x.series_major_pitch()
            if in_pitch is None and not x.tensor.is_contiguous():
                raise ValueError("device input must be C-contiguous or series-major; call .contiguous()")
            if out is not None:
                out_pitch = None if out.tensor.is_contiguous() else out.series_major_pitch()
                if out_pitch is None and not out.tensor.is_contiguous():
                    raise ValueError("device output must be C-contiguous or series-major")
            else:
                want = out_layout if out_layout is not None else ("series" if in_pitch is not None else "sample")
                if want not in ("sample", "series"):
                    raise ValueError("out_layout must be 'sample' or 'series'")
                if want == "series":
                    out = DeviceArray.empty_series_major(oshape, self.dtype, device=self.device, align_start=self.crop_start)
                    out_pitch = out.series_major_pitch()
                else:
                    out = self._empty_like_class(oshape, x) if in_pitch is None else DeviceArray.empty(oshape, self.dtype, device=self.device)
                    out_pitch = None
            if in_pitch is not None or out_pitch is not None:
                if tuple(out.shape) != oshape or out.dtype != self.dtype:
                    raise ValueError("output array does not match the plan")
                _check(lib().pbh_dedisperse_layout(self._h, C.c_void_p(x.raw_ptr()), int(in_pitch is not None),
                                                   int(in_pitch or 0), C.c_void_p(out.raw_ptr()),
                                                   int(out_pitch is not None), int(out_pitch or 0)))
                return out
        if out is None:
            if isinstance(x, DeviceArray):
                out = self._empty_like_class(oshape, x)
            else:
                out = np.empty(oshape, dtype=self.dtype)
        pin, lin = _ptr_loc(x)
        pout, lout = _ptr_loc(out)
        _check(lib().pbh_dedisperse(self._h, pin, pout, lin, lout))
        return out

    def stft_dedisperse(self, x, nperseg, out_layout="sample"):
        """``coherent_dedispersion(stft(x, nperseg))`` in one call (``pbh_stft_dedisperse``): ``x`` is the device-resident
        ``(nseg*nperseg, nchan_in, ...)`` block in front of the channeliser, this plan the dedispersion plan of the
        channelised block ``(nseg, nchan_in*nperseg, ...)``."""
        from .device import DeviceArray
        if not isinstance(x, DeviceArray):
            raise TypeError("stft_dedisperse takes a device-resident input")
        nperseg = int(nperseg)
        nchan_in = x.shape[1]
        inner = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
        if x.shape[0] != self.nsample * nperseg or nchan_in * nperseg != self.nchan or inner != self.npol or x.dtype != self.dtype:
            raise ValueError(f"input {tuple(x.shape)} {x.dtype} does not match a ({self.nsample}, {self.nchan}, {self.npol}) "
                             f"plan with nperseg = {nperseg}")
        self._sync_stream()
        oshape = (self.nout, self.nchan) + tuple(x.shape[2:])
        if self.nout == 0:
            return DeviceArray.empty(oshape, self.dtype, device=self.device)
        if out_layout == "series":
            out = DeviceArray.empty_series_major(oshape, self.dtype, device=self.device, align_start=self.crop_start)
            pitch = out.series_major_pitch()
        elif out_layout == "sample":
            out, pitch = DeviceArray.empty(oshape, self.dtype, device=self.device), 0
        else:
            raise ValueError("out_layout must be 'sample' or 'series'")
        _check(lib().pbh_stft_dedisperse(self._h, C.c_void_p(x.contiguous().data_ptr()), nperseg, int(nchan_in),
                                         C.c_void_p(out.raw_ptr()), int(out_layout == "series"), int(pitch)))
        return out

    def dedisperse_istft(self, x, nperseg):
        """``istft(coherent_dedispersion(x), nperseg)`` in one call (``pbh_dedisperse_istft``): ``x`` is the device-resident
        channelised block ``(nseg, nchan_out*nperseg, ...)`` this plan dedisperses (sample-major or series-major), the
        result the ``((stop-start)*nperseg, nchan_out, ...)`` time series."""
        from .device import DeviceArray
        if not isinstance(x, DeviceArray):
            raise TypeError("dedisperse_istft takes a device-resident input")
        nperseg = int(nperseg)
        self._check_in(x)
        if nperseg < 1 or self.nchan % nperseg:
            raise ValueError(f"{self.nchan} channels are not a whole number of {nperseg}-bin channels")
        nchan_out = self.nchan // nperseg
        self._sync_stream()
        out = DeviceArray.empty((self.nout * nperseg, nchan_out) + tuple(x.shape[2:]), self.dtype, device=self.device)
        if self.nout == 0:
            return out
        pitch = None if x.tensor.is_contiguous() else x.series_major_pitch()
        if pitch is None:
            x = x.contiguous()
        _check(lib().pbh_dedisperse_istft(self._h, C.c_void_p(x.raw_ptr()), int(pitch is not None), int(pitch or 0), nperseg,
                                          int(nchan_out), C.c_void_p(out.raw_ptr())))
        return out

    def dedisperse_mix(self, x, ft):
        """``IFFT(H * FFT(x * exp(2 pi i ft n)))`` on a C-contiguous device array, ``ft`` per series in cycles per sample
        (freq_shift: the mixer rides in the plan's de-interleave pass, ``pbh_dedisperse_mix``)."""
        from .device import DeviceArray
        if not isinstance(x, DeviceArray):
            raise TypeError("dedisperse_mix takes a device-resident input")
        self._check_in(x)
        self._sync_stream()
        a = np.ascontiguousarray(ft, dtype=np.float64)
        if a.shape != (self.nchan * self.npol,):
            raise ValueError("ft must have one entry per series")
        out = DeviceArray.empty((self.nout,) + tuple(x.shape[1:]), self.dtype, device=self.device)
        if self.nout == 0:
            return out
        _check(lib().pbh_dedisperse_mix(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                        a.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def dedisperse_slices(self, x, part_ptrs, part_rows, row_elems, col_offset):
        """``dedisperse_slice`` into a destination whose rows are split over several buffers: part i (device pointer
        ``part_ptrs[i]``) receives output rows ``[part_rows[i], part_rows[i+1])`` (``pbh_dedisperse_slices``)."""
        from .device import DeviceArray
        if not isinstance(x, DeviceArray):
            raise TypeError("dedisperse_slices takes a device-resident input")
        self._check_in(x)
        self._sync_stream()
        n = len(part_ptrs)
        if len(part_rows) != n + 1:
            raise ValueError("part_rows must have one more entry than part_ptrs")
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in part_ptrs])
        rows = (C.c_int64 * (n + 1))(*[int(r) for r in part_rows])
        _check(lib().pbh_dedisperse_slices(self._h, C.c_void_p(x.data_ptr()), n, ptrs, rows, int(row_elems), int(col_offset)))

    def dedisperse_slice(self, x, out_ptr, row_elems, col_offset):
        """Dedisperse device array ``x`` and write the ``(nout, nchan, npol)`` result as columns
        ``[col_offset, col_offset + nchan*npol)`` of the sample-major array at device pointer ``out_ptr`` whose rows
        are ``row_elems`` elements long -- this rank's channel slice of a full-band block, possibly on a peer GPU
        (``pbh_dedisperse_slice``).  Asynchronous on the current stream."""
        from .device import DeviceArray
        if not isinstance(x, DeviceArray):
            raise TypeError("dedisperse_slice takes a device-resident input")
        self._check_in(x)
        self._sync_stream()
        _check(lib().pbh_dedisperse_slice(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(int(out_ptr)), int(row_elems),
                                          int(col_offset)))

    def dedisperse_detect(self, x, nscrunch=1, mode="I", out=None):
        from .device import DeviceArray
        self._check_in(x)
        self._sync_stream()
        m = DETECT_MODES[mode]
        tail = {0: (self.nchan, self.npol), 1: (self.nchan,), 2: (self.nchan, 4), 3: (self.nchan, 4)}[m]
        oshape = (self.nout // int(nscrunch),) + tail
        if out is None:
            if isinstance(x, DeviceArray):
                out = DeviceArray.empty(oshape, self.real_dtype, device=self.device)
            else:
                out = np.empty(oshape, dtype=self.real_dtype)
        if isinstance(x, DeviceArray) and not x.tensor.is_contiguous():
            pitch = x.series_major_pitch()
            if pitch is None:
                raise ValueError("device input must be C-contiguous or series-major; call .contiguous()")
            rc = lib().pbh_dedisperse_detect_layout(self._h, C.c_void_p(x.raw_ptr()), 1, int(pitch),
                                                    C.c_void_p(out.data_ptr()), int(nscrunch), m)
            if rc != -2:      # PBH_ERR_UNSUPPORTED: this geometry detects from a sample-major block only
                _check(rc)
                return out
            x = x.contiguous()
        pin, lin = _ptr_loc(x)
        pout, lout = _ptr_loc(out)
        _check(lib().pbh_dedisperse_detect(self._h, pin, pout, int(nscrunch), m, lin, lout))
        return out

    def stream_detect(self, mode=None, nscrunch=1):
        """Make the streaming calls write detected rows (``pbh_plan_stream_detect``): ``mode`` one of DETECT_MODES, or None
        for voltages again.  Returns the shape of one chunk's output rows beyond the time axis."""
        if mode is None:
            _check(lib().pbh_plan_stream_detect(self._h, -1, 1))
            self._stream_tail = None
            return None
        m = DETECT_MODES[mode]
        _check(lib().pbh_plan_stream_detect(self._h, m, int(nscrunch)))
        self._stream_tail = (int(nscrunch), {0: (self.nchan, self.npol), 1: (self.nchan,), 2: (self.nchan, 4), 3: (self.nchan, 4)}[m])
        return self._stream_tail[1]

    def _stream_out(self, nchunk, sample_shape, out=None):
        """Host array a streaming call fills: voltages, or the detected rows stream_detect asked for (a caller's array is
        checked against that)."""
        tail = getattr(self, "_stream_tail", None)
        if tail is None:
            shape, dtype = (nchunk * self.nout,) + tuple(sample_shape), self.dtype
        else:
            shape, dtype = (nchunk * (self.nout // tail[0]),) + tail[1], self.real_dtype
        if out is None:
            return np.empty(shape, dtype=dtype)
        if not isinstance(out, np.ndarray) or out.dtype != dtype or not out.flags.c_contiguous or out.size != int(np.prod(shape)):
            raise ValueError(f"out must be a C-contiguous {np.dtype(dtype).name} numpy array of {int(np.prod(shape))} elements {shape}")
        return out

    def dedisperse_stream(self, x_host, out=None):
        """Overlap-save over a long host block: (total, nchan, npol) c64 -> (nchunk*hop, ...), plus ms."""
        if not isinstance(x_host, np.ndarray) or x_host.dtype != self.dtype or not x_host.flags.c_contiguous:
            raise TypeError(f"dedisperse_stream needs a C-contiguous {self.dtype} numpy array")
        if int(np.prod(x_host.shape[1:])) != self.nchan * self.npol:
            raise ValueError("sample shape does not match the plan")
        hop = self.nout
        if hop <= 0 or x_host.shape[0] < self.nsample:
            raise ValueError("empty valid region or input shorter than one chunk")
        nchunk = (x_host.shape[0] - self.nsample) // hop + 1
        out = self._stream_out(nchunk, x_host.shape[1:], out)
        self._sync_stream()
        n, ms = C.c_int64(), C.c_float()
        _check(lib().pbh_dedisperse_stream(self._h, C.c_void_p(x_host.ctypes.data), int(x_host.shape[0]),
                                           C.c_void_p(out.ctypes.data), C.byref(n), C.byref(ms)))
        assert n.value == nchunk
        return out, float(ms.value)

    def stream_stats(self):
        """Figures of this plan's last streaming call (``pbh_stream_stats``): bytes over PCIe each way, how long each
        copy stream was busy, kernel time, total time; GB/s and the overlap efficiency derived from them."""
        v = (C.c_double * 8)()
        _check(lib().pbh_stream_stats(self._h, v, 8))
        d = dict(h2d_bytes=v[0], d2h_bytes=v[1], h2d_ms=v[2], d2h_ms=v[3], kernel_ms=v[4], total_ms=v[5],
                 nchunk=int(v[6]), d2d_bytes=v[7])
        d["h2d_GBps"] = v[0] / v[2] / 1e6 if v[2] > 0 else 0.0
        d["d2h_GBps"] = v[1] / v[3] / 1e6 if v[3] > 0 else 0.0
        # 1.0 = the call took as long as its slowest stage alone (perfect overlap of upload, kernels and download)
        d["overlap_efficiency"] = max(v[2], v[3], v[4]) / v[5] if v[5] > 0 else 0.0
        return d

    def dedisperse_stream_raw(self, raw, layout, total_nsample, first=0, conj=None, scale=1.0, out=None):
        """``dedisperse_stream`` fed with raw payload bytes (uint8 numpy array + ``pbh_raw_layout_t`` fields):
        decode on the device, chunk by chunk.  Returns ((nchunk*hop, nchan, npol) complex64, ms)."""
        if not isinstance(raw, np.ndarray) or raw.dtype != np.uint8 or not raw.flags.c_contiguous:
            raise TypeError("dedisperse_stream_raw needs a C-contiguous uint8 numpy array")
        hop = self.nout
        if hop <= 0 or total_nsample < self.nsample:
            raise ValueError("empty valid region or input shorter than one chunk")
        nchunk = (int(total_nsample) - self.nsample) // hop + 1
        out = self._stream_out(nchunk, (self.nchan, self.npol), out)
        lay = _RawLayout(**{k: int(v) for k, v in layout.items()})
        mask = None
        if conj is not None:
            mask = np.ascontiguousarray(np.broadcast_to(np.asarray(conj, dtype=bool), (self.nchan, self.npol)), dtype=np.uint8)
        self._sync_stream()
        n, ms = C.c_int64(), C.c_float()
        _check(lib().pbh_dedisperse_stream_raw(self._h, C.c_void_p(raw.ctypes.data), raw.size, C.byref(lay),
                                               int(first), int(total_nsample), None if mask is None else C.c_void_p(mask.ctypes.data),
                                               float(scale), C.c_void_p(out.ctypes.data), C.byref(n), C.byref(ms)))
        assert n.value == nchunk
        return out, float(ms.value)

    def profile(self, x_dev, out_dev, iters=10):
        """Mean per-kernel milliseconds (hipEvents on the plan's stream): list of (name, ms)."""
        self._sync_stream()
        ms = (C.c_float * MAX_KERNELS)()
        names = (C.c_char_p * MAX_KERNELS)()
        nk = C.c_int()
        _check(lib().pbh_plan_profile(self._h, C.c_void_p(x_dev.data_ptr()), C.c_void_p(out_dev.data_ptr()),
                                      int(iters), ms, C.byref(nk), names))
        return [(names[i].decode(), float(ms[i])) for i in range(nk.value)]


def detect(x, mode="intensity", nscrunch=1):
    """to_intensity / to_stokes (+ optional time scrunch) of (n, nchan, npol) c64 data."""
    from .device import DeviceArray
    _require_device()
    m = DETECT_MODES[mode]
    n, nchan = x.shape[0], x.shape[1]
    npol = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    code = _dtype_code(x.dtype)
    rdt = REAL_OF[np.dtype(x.dtype)]
    if m == 0:
        oshape = (n // nscrunch,) + tuple(x.shape[1:])
    elif m == 1:
        oshape = (n // nscrunch, nchan)
    else:
        oshape = (n // nscrunch, nchan, 4)
    if isinstance(x, DeviceArray):
        x = x.contiguous()   # a strided (e.g. series-major) view costs one copy here
        out = DeviceArray.empty(oshape, rdt, device=x.device_index)
        dev, stream = x.device_index, _stream_ptr(x.device_index)
    else:
        x = np.ascontiguousarray(x)
        out = np.empty(oshape, dtype=rdt)
        import torch
        dev = torch.cuda.current_device()   # host data: the process's current GPU (one process per GPU)
        stream = _stream_ptr(dev)
    pin, lin = _ptr_loc(x)
    pout, lout = _ptr_loc(out)
    _check(lib().pbh_detect(dev, stream, code, pin, pout, int(n), int(nchan), npol, m, int(nscrunch), lin, lout))
    return out


def decode(raw, layout, first, nsample, nchan, npol, *, conj=None, scale=1.0, series_major=False, device=None):
    """Raw integer payload bytes -> DeviceArray (nsample, nchan, npol) float32 / complex64 (pbh_decode).

    ``raw``: C-contiguous uint8 numpy array (the bytes of whole blocks, headers included) or a uint8
    DeviceArray; ``layout``: dict with the fields of ``pbh_raw_layout_t``; ``conj``: per-series booleans."""
    from .device import DeviceArray
    _require_device()
    lay = _RawLayout(**{k: int(v) for k, v in layout.items()})
    dtype = np.complex64 if lay.ncomp == 2 else np.float32
    shape = (int(nsample), int(nchan), int(npol))
    if isinstance(raw, DeviceArray):
        dev = raw.device_index
        praw, loc, nbytes = C.c_void_p(raw.data_ptr()), DEVICE, int(np.prod(raw.shape))
    else:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        import torch
        dev = torch.cuda.current_device() if device is None else int(device)
        praw, loc, nbytes = C.c_void_p(raw.ctypes.data), HOST, raw.size
    if series_major and nchan * npol > 1:
        out = DeviceArray.empty_series_major(shape, dtype, device=dev)
        code, pitch = 1, out.series_major_pitch()
    else:
        out = DeviceArray.empty(shape, dtype, device=dev)
        code, pitch = 0, 0
    mask = None
    if conj is not None:
        mask = np.ascontiguousarray(np.broadcast_to(np.asarray(conj, dtype=bool), (nchan, npol)), dtype=np.uint8)
    _check(lib().pbh_decode(dev, _stream_ptr(dev), praw, nbytes, loc, C.byref(lay), int(first), int(nsample), int(nchan),
                            int(npol), None if mask is None else C.c_void_p(mask.ctypes.data), float(scale),
                            C.c_void_p(out.raw_ptr() if code else out.data_ptr()), code, int(pitch or 0)))
    return out


def fft_c2c(x, inverse=False):
    """c2c FFT along axis 0 of an (n, ...) c64 array (numpy or DeviceArray), scipy norm=None."""
    from .device import DeviceArray
    _require_device()
    code = _dtype_code(x.dtype)
    n = x.shape[0]
    batch = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    if isinstance(x, DeviceArray):
        x = x.contiguous()
        out = DeviceArray.empty(x.shape, x.dtype, device=x.device_index)
        dev, stream = x.device_index, _stream_ptr(x.device_index)
    else:
        x = np.ascontiguousarray(x)
        out = np.empty(x.shape, dtype=x.dtype)
        dev, stream = 0, C.c_void_p(0)
    pin, lin = _ptr_loc(x)
    pout, lout = _ptr_loc(out)
    _check(lib().pbh_fft_c2c(dev, stream, code, pin, pout, int(n), batch, int(bool(inverse)), lin, lout))
    return out


def mix(x_dev, ft):
    """In-place x[n, s] *= exp(2 pi i ft[s] n) on a device (n, s) array."""
    a = np.ascontiguousarray(ft, dtype=np.float64)
    n, s = x_dev.shape[0], int(np.prod(x_dev.shape[1:]))
    p = C.c_void_p(x_dev.data_ptr())
    _check(lib().pbh_mix(x_dev.device_index, _stream_ptr(x_dev.device_index), _dtype_code(x_dev.dtype), p, p, int(n), s,
                         a.ctypes.data_as(C.POINTER(C.c_double))))
    return x_dev


def zero_edges(x_dev, shift):
    """time_shift's zero fill on a device (n, s) array (in place)."""
    a = np.ascontiguousarray(shift, dtype=np.float64)
    n, s = x_dev.shape[0], int(np.prod(x_dev.shape[1:]))
    _check(lib().pbh_zero_edges(x_dev.device_index, _stream_ptr(x_dev.device_index), _dtype_code(x_dev.dtype),
                                C.c_void_p(x_dev.data_ptr()), int(n), s, a.ctypes.data_as(C.POINTER(C.c_double))))
    return x_dev


def real_to_complex_half(x_dev):
    """``utils.real_to_complex`` of a C-contiguous device (n, s) float32 array as a half-length complex transform
    (``pbh_real_to_complex``); returns the (n/2, s) complex64 DeviceArray, or None when the geometry is not covered."""
    from .device import DeviceArray
    n, s = int(x_dev.shape[0]), int(np.prod(x_dev.shape[1:]))
    m = n // 2
    if x_dev.dtype != np.float32 or n % 2 or m & (m - 1) or not (1 << 15) <= m <= (1 << 24) or s > 65535 or os.environ.get("PBH_R2C_HALF", "1") == "0":
        return None
    out = DeviceArray.empty((m,) + tuple(x_dev.shape[1:]), np.complex64, device=x_dev.device_index)
    rc = lib().pbh_real_to_complex(x_dev.device_index, _stream_ptr(x_dev.device_index), C.c_void_p(x_dev.data_ptr()),
                                   C.c_void_p(out.data_ptr()), n, s)
    if rc == -2:      # PBH_ERR_UNSUPPORTED: the full-length route
        return None
    _check(rc)
    return out


def decimate2(y_dev):
    """out[m, s] = (-1)^m y[2m, s] for a device (n, s) complex array."""
    from .device import DeviceArray
    y_dev = y_dev.contiguous()
    n, s = y_dev.shape[0], int(np.prod(y_dev.shape[1:]))
    nout = (n + 1) // 2
    out = DeviceArray.empty((nout,) + tuple(y_dev.shape[1:]), y_dev.dtype, device=y_dev.device_index)
    _check(lib().pbh_decimate2(y_dev.device_index, _stream_ptr(y_dev.device_index), _dtype_code(y_dev.dtype),
                               C.c_void_p(y_dev.data_ptr()), C.c_void_p(out.data_ptr()), int(nout), s))
    return out


def pol_basis(x_dev, to_circular):
    """to_circular / to_linear of a device (n, nchan, 2) complex array -> new DeviceArray."""
    from .device import DeviceArray
    x_dev = x_dev.contiguous()
    out = DeviceArray.empty(x_dev.shape, x_dev.dtype, device=x_dev.device_index)
    _check(lib().pbh_pol_basis(x_dev.device_index, _stream_ptr(x_dev.device_index), _dtype_code(x_dev.dtype),
                               C.c_void_p(x_dev.data_ptr()), C.c_void_p(out.data_ptr()), x_dev.size // 2,
                               int(bool(to_circular))))
    return out


def incoherent(x_dev, delays, nout):
    """out[n, c, ...] = x[n + delays[c], c, ...], n < nout, for a device array of any 4/8/16-byte dtype."""
    from .device import DeviceArray
    x_dev = x_dev.contiguous()
    nchan = x_dev.shape[1]
    inner = int(np.prod(x_dev.shape[2:])) if x_dev.ndim > 2 else 1
    unit = inner * x_dev.dtype.itemsize // 4
    d = np.ascontiguousarray(delays, dtype=np.int64)
    out = DeviceArray.empty((int(nout),) + tuple(x_dev.shape[1:]), x_dev.dtype, device=x_dev.device_index)
    _check(lib().pbh_incoherent(x_dev.device_index, _stream_ptr(x_dev.device_index), C.c_void_p(x_dev.data_ptr()),
                                C.c_void_p(out.data_ptr() if nout > 0 else x_dev.data_ptr()), int(nout), int(nchan),
                                int(unit), d.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def incoherent_series(x_dev, delays, nout):
    """The same gather on a series-major device array (time fastest): one shifted contiguous copy per series."""
    from .device import DeviceArray
    pitch = x_dev.series_major_pitch()
    if pitch is None:
        raise ValueError("incoherent_series needs a series-major device array")
    nchan = x_dev.shape[1]
    inner = int(np.prod(x_dev.shape[2:])) if x_dev.ndim > 2 else 1
    words = x_dev.tensor.element_size() // 4
    if words not in (1, 2, 4):
        raise TypeError("elements must be 4, 8 or 16 bytes")
    d = np.ascontiguousarray(delays, dtype=np.int64)
    out = DeviceArray.empty_series_major((int(nout),) + tuple(x_dev.shape[1:]), x_dev.dtype, device=x_dev.device_index)
    if nout > 0:
        _check(lib().pbh_incoherent_series(x_dev.device_index, _stream_ptr(x_dev.device_index), C.c_void_p(x_dev.raw_ptr()),
                                           int(pitch) * words, C.c_void_p(out.raw_ptr()), int(out.series_major_pitch()) * words,
                                           int(nout), int(nchan), inner, words, d.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def stft(x, nperseg, inverse=False):
    """contrib.stft / istft core on (nseg*nperseg, nchan, ...) [stft] or (nseg, nchan*nperseg, ...) [istft] data."""
    from .device import DeviceArray
    _require_device()
    code = _dtype_code(x.dtype)
    n = int(nperseg)
    inner = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    if inverse:
        if x.shape[1] % n:
            raise ValueError("channel axis is not a multiple of nperseg")
        nseg, nchan = x.shape[0], x.shape[1] // n
        oshape = (nseg * n, nchan) + tuple(x.shape[2:])
    else:
        if x.shape[0] % n:
            raise ValueError("time axis is not a multiple of nperseg")
        nseg, nchan = x.shape[0] // n, x.shape[1]
        oshape = (nseg, nchan * n) + tuple(x.shape[2:])
    if isinstance(x, DeviceArray):
        x = x.contiguous()
        out = DeviceArray.empty(oshape, x.dtype, device=x.device_index)
        dev, stream = x.device_index, _stream_ptr(x.device_index)
    else:
        x = np.ascontiguousarray(x)
        out = np.empty(oshape, dtype=x.dtype)
        dev, stream = 0, C.c_void_p(0)
    pin, lin = _ptr_loc(x)
    pout, lout = _ptr_loc(out)
    _check(lib().pbh_stft(dev, stream, code, pin, pout, int(nseg), n, int(nchan), inner, int(bool(inverse)), lin, lout))
    return out


def chirp_function(coeff_hz, nsample, dt_s, center_freq_hz, ref_freq_hz, device=0, to_device=False):
    """One channel's transfer function (nsample,) c64, generated by the HIP kernel."""
    from .device import DeviceArray
    _require_device()
    if to_device:
        out = DeviceArray.empty((int(nsample),), np.complex64, device=device)
        stream = _stream_ptr(device)
    else:
        out = np.empty((int(nsample),), dtype=np.complex64)
        stream = C.c_void_p(0)
    ptr, loc = _ptr_loc(out)
    _check(lib().pbh_chirp_function(int(device), stream, float(coeff_hz), int(nsample), float(dt_s),
                                    float(center_freq_hz), float(ref_freq_hz), ptr, loc))
    return out


def copy_bench(nbytes, iters=10, device=0):
    """Mean ms of a float4 device-to-device copy of ``nbytes`` (HBM yardstick)."""
    _require_device()
    ms = C.c_float()
    _check(lib().pbh_copy_bench(int(device), int(nbytes), int(iters), C.byref(ms)))
    return float(ms.value)


def stream_bench(nbytes, iters=10, device=0, mode="copy"):
    """Mean ms per launch of the streaming yardstick: ``copy`` (two buffers) or ``rmw`` (one buffer, read-modify-write in
    place: what the three middle passes do to the planar work buffer)."""
    _require_device()
    ms = C.c_float()
    _check(lib().pbh_stream_bench(int(device), int(nbytes), int(iters), {"copy": 0, "rmw": 1}[mode], C.byref(ms)))
    return float(ms.value)
