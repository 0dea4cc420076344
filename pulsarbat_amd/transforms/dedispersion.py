"""Dedispersion: the hot path's host side (reference pulsarbat/transforms/dedispersion.py).

Everything numeric happens in libpbhip.so (HIP kernels); this module keeps the
reference's call surface -- ``DispersionMeasure`` (``time_delay``, ``sample_delay``,
``chirp_function``, ``chirp_from_signal``) and ``coherent_dedispersion(z, DM, /, *,
ref_freq=None, chirp=None)`` -- its type checks, crop arithmetic
(dedispersion.py:127-131) and ``start_time`` bookkeeping (core.py:155-164).
The device boundary sits at dedispersion.py:125 (``ifft(fft(x) * chirp)``) and
dedispersion.py:19-23 (``_transfer_function``).  There is no CPU fallback.
"""

import math
from collections import OrderedDict
import threading

import numpy as np

from .. import _hip
from .. import units as u
from ..core import BasebandSignal, RadioSignal
from ..device import DeviceArray

__all__ = [
    "DispersionMeasure",
    "DM",
    "coherent_dedispersion",
    "dedisperse_detect",
    "coherent_dedispersion_stream",
    "incoherent_dedispersion",
]

_DM_UNIT = u.pc / u.cm ** 3
_DM_UNIT.name = "pc / cm3"


def _hz(q):
    """Frequency as float Hz; ``numpy.inf`` passes through (tests/test_dedispersion.py:17-21)."""
    if isinstance(q, (int, float)) and math.isinf(q):
        return float(q)
    return u.to_value(q, u.Hz)


class DispersionMeasure(u.Quantity):
    """Dispersion measure, default unit pc / cm^3 (dedispersion.py:26-30)."""

    __slots__ = ()
    # dedispersion.py:30: s MHz^2 cm^3 / pc / 2.41e-4
    dispersion_constant = u.Quantity(1.0 / 2.41e-4, u.s * u.MHz ** 2 * u.cm ** 3 / u.pc)

    def __init__(self, value, unit=None):
        if isinstance(value, u.Quantity) and unit is None:
            value, unit = value.to_value(_DM_UNIT), _DM_UNIT
        unit = _DM_UNIT if unit is None else unit
        if not u._as_unit(unit).is_equivalent(_DM_UNIT):
            raise u.UnitConversionError("DispersionMeasure needs units equivalent to pc / cm^3")
        super().__init__(value, unit)

    @property
    def _coeff_s_mhz2(self):
        """dispersion_constant * DM in s MHz^2 (dedispersion.py:34, 46)."""
        return self.to_value(_DM_UNIT) / 2.41e-4

    def time_delay(self, f, ref_freq):
        """Delay of ``f`` relative to ``ref_freq``: D*DM*(1/f^2 - 1/ref^2) (dedispersion.py:32-36)."""
        f_mhz = np.asarray(_hz(f), dtype=float) / 1e6
        r_mhz = np.asarray(_hz(ref_freq), dtype=float) / 1e6
        with np.errstate(divide="ignore"):
            d = self._coeff_s_mhz2 * (1.0 / f_mhz ** 2 - 1.0 / r_mhz ** 2)
        return u.Quantity(d if d.ndim else float(d), u.s)

    def sample_delay(self, f, ref_freq, sample_rate):
        """time_delay * sample_rate as plain floats (dedispersion.py:38-42)."""
        # numpy scalar / array as in the reference (callers use .round(): tests/test_dedispersion.py:187)
        return np.asarray(self.time_delay(f, ref_freq).to_value(u.s) * _hz(sample_rate))[()]

    def chirp_function(self, N, dt, center_freq, ref_freq, use_dask=False, *, device=None):
        """Transfer function of one channel, complex64 ``(N,)`` (dedispersion.py:44-57).

        Computed by the HIP chirp kernel (float64 phase).  ``use_dask`` is accepted for
        signature parity and ignored; ``device=<index>`` returns a DeviceArray instead of
        a host array.
        """
        return _hip.chirp_function(
            self._coeff_s_mhz2 * 1e12, int(N), u.to_value(dt, u.s), _hz(center_freq), _hz(ref_freq),
            device=0 if device is None else device, to_device=device is not None)

    def chirp_from_signal(self, z, /, *, ref_freq=None):
        """Chirp ``(N, nchan, 1, ...)`` that dedisperses ``z`` (dedispersion.py:59-75).

        Host-resident signals get a numpy array, device-resident ones a DeviceArray.
        """
        if not isinstance(z, BasebandSignal):
            raise TypeError("Signal must be a BasebandSignal object.")
        if ref_freq is None:
            ref_freq = z.center_freq
        plan, on_device = _plan_for(z, self, ref_freq, crop=(0, len(z)))
        shape = (len(z), z.nchan) + (1,) * (z.ndim - 2)
        if on_device:
            out = DeviceArray.empty((len(z), z.nchan), np.complex64, device=plan.device)
            plan.chirp_download(out)
            return DeviceArray(out.tensor.reshape(shape))
        return plan.chirp_download().reshape(shape)

    chirp = chirp_from_signal  # BASELINE.json north_star spells it `.chirp`


DM = DispersionMeasure

# ---- plan cache --------------------------------------------------------------------------------
_PLANS = OrderedDict()
_PLANS_LOCK = threading.Lock()
_PLAN_CACHE_SIZE = 4


def _device_of(z):
    """Device of the signal's data; host data goes to the process's current device (one process per GPU:
    torch.cuda.set_device(LOCAL_RANK) is what selects it)."""
    if isinstance(z.data, DeviceArray):
        return z.data.device_index
    import torch
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


def _geometry(z):
    nchan = z.shape[1]
    npol = int(np.prod(z.shape[2:])) if z.ndim > 2 else 1
    return len(z), nchan, npol


def _plan_for(z, dm, ref_freq, crop, chirp=None, variant="auto", per_pol=False, dtype=None, device=None):
    """Cached pbh plan for this signal geometry (+ generated chirp).  ``per_pol``: the uploaded chirp has one
    row per (channel, polarisation) series, i.e. the plan is (nsample, nchan*npol, 1).  ``crop`` need not be the
    signal's own: a rank of a channel-sharded job passes the full band's (shard.py)."""
    nsample, nchan, npol = _geometry(z)
    if per_pol:
        nchan, npol = nchan * npol, 1
    dev = _device_of(z) if device is None else int(device)
    dtype = np.dtype(z.dtype if dtype is None else dtype)
    ckey = None
    if chirp is None:
        freqs = np.asarray(u.to_value(z.channel_freqs, u.Hz), dtype=np.float64).reshape(nchan)
        coeff = dm._coeff_s_mhz2 * 1e12
        dt = u.to_value(z.dt, u.s)
        ckey = (coeff, dt, freqs.tobytes(), _hz(ref_freq))
    # (a plan is not re-entrant: the cache is per thread, so concurrent callers never share one)
    key = (nsample, nchan, npol, crop, dev, variant, dtype.str, threading.get_ident())
    with _PLANS_LOCK:
        ent = _PLANS.pop(key, None)
    if ent is None:
        plan = _hip.Plan(nsample, nchan, npol, crop[0], crop[1], device=dev, variant=variant, dtype=dtype)
        ent = [plan, object()]
    plan = ent[0]
    if chirp is not None:
        plan.chirp_upload(chirp)
        ent[1] = object()
    elif ent[1] != ckey:
        plan.chirp_generate(ckey[0], ckey[1], freqs, ckey[3])
        ent[1] = ckey
    stale = []
    with _PLANS_LOCK:
        _PLANS[key] = ent
        mine = [k for k in _PLANS if k[-1] == key[-1]]
        while len(mine) > _PLAN_CACHE_SIZE:      # the limit is per thread
            stale.append(_PLANS.pop(mine.pop(0)))
    for old in stale:
        old[0].close()
    return plan, isinstance(z.data, DeviceArray)


def clear_plan_cache():
    """Destroy the cached dedispersion plans and the library's own per-thread transform plans."""
    with _PLANS_LOCK:
        stale = list(_PLANS.values())
        _PLANS.clear()
    for ent in stale:
        ent[0].close()
    _hip.trim()


def _crop_bounds(z, dm, ref_freq):
    """start/stop of dedispersion.py:127-131."""
    delay_top = float(dm.sample_delay(z.max_freq, ref_freq, z.sample_rate))
    delay_bot = float(dm.sample_delay(z.min_freq, ref_freq, z.sample_rate))
    start = math.ceil(-min(0, delay_top, delay_bot))
    stop = len(z) - math.ceil(+max(0, delay_top, delay_bot))
    return start, stop


def _broadcast_chirp(chirp, z):
    """A user ``chirp=`` is multiplied into the spectrum unchecked: ``fft(z.data, axis=0) * chirp``
    (dedispersion.py:124-125), so anything numpy can broadcast against ``z.data`` is accepted -- the documented
    ``(N, nchan)`` / ``(N, nchan, 1...)`` forms (dedispersion.py:103-105), one row per polarisation
    ``(N, nchan, npol)``, one row for all channels ``(N,)`` / ``(N, 1)``, a 0-d array...  (A chirp with fewer axes than
    the signal gets length-1 axes appended first, as in the reference.)  Returns ``(rows, per_pol, dtype)``:
    the chirp as a C-contiguous ``(N, nseries)`` array (numpy or DeviceArray) with one column per channel, or per
    (channel, pol) series when it differs between polarisations, in the dtype of numpy's product."""
    full = tuple(z.shape)
    on_dev = isinstance(chirp, DeviceArray)
    c = chirp if on_dev else np.asarray(chirp)
    cshape = tuple(c.shape)
    if len(cshape) > len(full):
        raise ValueError(f"operands could not be broadcast together with shapes {full} {cshape}")
    padded = cshape + (1,) * (len(full) - len(cshape))   # dedispersion.py:124: length-1 axes are appended, THEN numpy broadcasts
    if any(a != 1 and a != b for a, b in zip(padded, full)):
        raise ValueError(f"operands could not be broadcast together with shapes {full} {cshape}")
    per_pol = any(a != 1 for a in padded[2:])
    target = full if per_pol else full[:2] + (1,) * (len(full) - 2)
    dtype = np.result_type(z.dtype, c.dtype)
    if dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
        raise TypeError(f"the product of {z.dtype} data and a {c.dtype} chirp is {dtype}; the HIP path takes complex64/128")
    if on_dev:
        import torch
        t = torch.broadcast_to(c.tensor.reshape(padded), target).to(_TORCH_OF[dtype]).reshape(full[0], -1).contiguous()
        return DeviceArray(t), per_pol, dtype
    rows = np.ascontiguousarray(np.broadcast_to(c.reshape(padded), target).reshape(full[0], -1), dtype=dtype)
    return rows, per_pol, dtype


class _TorchOf(dict):
    def __missing__(self, key):
        import torch
        return {np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(key)]


_TORCH_OF = _TorchOf()


def _prepare(z, DM, ref_freq, chirp, variant, allow_series=False):
    if not isinstance(z, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    if ref_freq is None:
        ref_freq = z.center_freq
    _hip._dtype_code(z.dtype)  # complex64 -> float32 kernels, complex128 -> float64 kernels (as scipy.fft)
    start, stop = _crop_bounds(z, DM, ref_freq)
    rows, per_pol, dtype = (None, False, np.dtype(z.dtype)) if chirp is None else _broadcast_chirp(chirp, z)
    plan, on_device = _plan_for(z, DM, ref_freq, (start, stop), chirp=rows, variant=variant, per_pol=per_pol, dtype=dtype)
    data = z.data
    if dtype != np.dtype(z.dtype):   # complex64 data times a complex128 chirp is complex128 in numpy: so is the result here
        data = data.astype(dtype)
    if on_device:
        # a series-major (time-fastest) device array goes through as it is when the plan has the
        # layout-aware path (multi-pass power-of-two lengths); otherwise one contiguous copy
        keep = (allow_series and not data.tensor.is_contiguous() and data.series_major_pitch() is not None
                and plan.supports_series_major)
        x = data if keep else data.contiguous()
    else:
        x = np.ascontiguousarray(data)
    return plan, x, start, stop


def _advance(z, start):
    if z.start_time is None:
        return {}
    return {"start_time": z.start_time + start / z.sample_rate}


def coherent_dedispersion(z, DM, /, *, ref_freq=None, chirp=None, variant="auto"):
    """Coherently dedisperse a baseband signal (dedispersion.py:81-133).

    ``z`` must be a BasebandSignal with complex64 data, host (numpy) or device
    (DeviceArray); the result keeps the container type.  ``ref_freq`` defaults to the
    signal's centre frequency.  A pre-computed ``chirp`` (shape ``z.shape[:2]``) is used
    unchecked, as in the reference.  The output is cropped to ``[start, stop)`` to drop
    wrap-around, and ``start_time`` advances by ``start / sample_rate``.  ``variant`` is a
    build-specific knob selecting the kernel sequence ("auto", "direct3", "planar5").
    A DeviceArray stored series-major (``DeviceArray.to_series_major``) stays that way: input and
    output skip their layout passes.
    """
    plan, x, start, stop = _prepare(z, DM, ref_freq, chirp, variant, allow_series=True)
    y = plan.dedisperse(x)
    return type(z).like(z, y, **_advance(z, start))


def dedisperse_detect(z, DM, /, *, ref_freq=None, chirp=None, mode="I", nscrunch=1, variant="auto"):
    """coherent_dedispersion followed by detection and an ``nscrunch``-fold time sum.

    Equivalent to ``coherent_dedispersion(z, DM).to_intensity()`` (mode "intensity":
    core.py:766-774) or ``.to_stokes()`` (mode "linear"/"circular": core.py:930-966; "I":
    Stokes I only), then ``a[:n*k].reshape(n, k, ...).sum(1)``.  The reference has no
    scrunch function; SURVEY.md 8a row 9 defines it.  Returns a float32 array (numpy or
    DeviceArray) and the crop start, not a Signal, since the sample rate changes.
    """
    # a series-major device array goes through as it is when a fused tail applies (nscrunch % 64 == 0, or 1; the plan falls
    # back to a sample-major copy for the geometries whose tail is not fused)
    plan, x, start, stop = _prepare(z, DM, ref_freq, chirp, variant, allow_series=int(nscrunch) % 64 == 0 or int(nscrunch) == 1)
    if plan.nchan != z.nchan:
        # a chirp that differs between polarisations runs as nchan*npol single-pol channels: detection needs the
        # (channel, pol) structure back, so it is a pass of its own here
        y = plan.dedisperse(x)   # keeps x's (nsample, nchan, npol...) trailing shape
        return _hip.detect(y, mode=mode, nscrunch=nscrunch), start
    return plan.dedisperse_detect(x, nscrunch=nscrunch, mode=mode), start


def coherent_dedispersion_stream(z, DM, /, *, chunk, ref_freq=None, variant="auto", offset=0, n=None, channels=None,
                                 detect=None, nscrunch=1):
    """Overlap-save dedispersion of a long host-resident signal in chunks of ``chunk`` samples.

    Equals ``pb.concatenate([coherent_dedispersion(z[k*hop : k*hop + chunk], DM, ref_freq=ref)
    for k in range(nchunk)])`` with ``hop`` the valid length of one chunk (the reference's own
    recipe for long series: SURVEY.md 5, transforms.py:59-148), but uploads chunk k+1 and downloads
    chunk k-1 while chunk k is on the GPU (double-buffered hipMemcpyAsync).  Returns the signal
    (start_time advanced by the crop start) and the HIP-event milliseconds of the whole stream.

    ``z`` may also be a ``pulsarbat_amd.readers.BasebandReader`` of complex voltage data (``offset`` / ``n`` select
    the samples): then the file's payload bytes are what crosses PCIe and every chunk is unpacked on the
    device in front of its transforms (``pbh_dedisperse_stream_raw``) -- the same result as streaming
    ``reader.read(offset, n)``, at a quarter of the upload for 8-bit samples.  With a reader, ``channels`` (a slice, e.g.
    ``shard.channel_slice(nchan, world, rank)``) streams a rank's share of a channel-sharded job: the result equals the
    full stream's ``[:, channels]`` -- crop and reference frequency are the FULL band's (dedispersion.py:118-131), the
    chirp the subset's -- and for channel-major files only that share of the payload is read and uploaded.

    ``detect`` ("intensity", "I", "linear", "circular") makes it a filterbank stream: every chunk ends in the fused detect
    tail of ``dedisperse_detect`` and what comes back is ``(float32 array (nchunk * hop / nscrunch, nchan[, npol | 4]),
    crop start, ms)`` -- ``to_intensity`` / ``to_stokes`` of the dedispersed stream summed over ``nscrunch`` samples, the
    voltages never leave the GPU.  A chunk's valid region is shortened to a whole number of scrunch blocks, so the chunks
    stay contiguous in time.
    """
    from ..readers import BasebandReader
    if detect is not None and (int(nscrunch) != 1 and int(nscrunch) % 64 != 0):
        raise ValueError("a detected stream needs nscrunch == 1 or a multiple of 64 (the fused detect tails)")
    if isinstance(z, BasebandReader):
        return _stream_from_reader(z, DM, chunk, ref_freq, variant, offset, n, channels, detect, int(nscrunch))
    if channels is not None:
        raise TypeError("channels= applies to streaming from a reader; slice a signal with z[:, channels]")
    if not isinstance(z, BasebandSignal):
        raise TypeError("Signal must be a BasebandSignal object.")
    if isinstance(z.data, DeviceArray):
        raise TypeError("coherent_dedispersion_stream takes a host-resident signal")
    if offset or n is not None:
        z = z[offset:None if n is None else offset + n]
    if ref_freq is None:
        ref_freq = z.center_freq
    head = z[:chunk]
    start, stop = _crop_bounds(head, DM, ref_freq)
    if detect is not None:
        stop -= (stop - start) % int(nscrunch)
    plan, _ = _plan_for(head, DM, ref_freq, (start, stop), variant=variant)
    if detect is None:
        y, ms = plan.dedisperse_stream(np.ascontiguousarray(z.data))
        return type(z).like(z, y, **_advance(z, start)), ms
    plan.stream_detect(detect, nscrunch)
    try:   # (plans are cached and shared: the tail is this call's)
        y, ms = plan.dedisperse_stream(np.ascontiguousarray(z.data))
    finally:
        plan.stream_detect(None)
    return y, start, ms


def _stream_from_reader(reader, DM, chunk, ref_freq, variant, offset, n, channels=None, detect=None, nscrunch=1):
    if not issubclass(reader._signal_type, BasebandSignal) or reader.intensity or not reader.complex_data:
        raise TypeError("streaming from a reader needs complex voltage data in a BasebandSignal type")
    n = len(reader) - offset if n is None else n
    if offset < 0 or n < 0 or offset + n > len(reader):
        raise EOFError("Cannot read beyond end of stream")
    # a zero-stride stand-in carries the metadata the plan is built from
    blank = np.broadcast_to(np.complex64(0), (chunk,) + reader.sample_shape)
    head = reader._signal_type(blank, sample_rate=reader.sample_rate, start_time=reader.time_at(offset),
                               **reader._signal_kwargs)
    if ref_freq is None:
        ref_freq = head.center_freq
    start, stop = _crop_bounds(head, DM, ref_freq)          # the FULL band's, whatever the channel subset
    if detect is not None:
        stop -= (stop - start) % nscrunch
    if channels is not None:
        if not isinstance(channels, slice) or channels.step not in (None, 1):
            raise TypeError("channels must be a contiguous slice")
        if len(reader.sample_shape) != len(reader._raw.sample_shape):
            raise ValueError("channels= needs the unsqueezed (channel, polarisation) sample axes")
        head = head[:, channels]                            # the container's own channel bookkeeping (core.py:479-498)
        if head.shape[1] == 0:
            raise ValueError("channels selects no channel")
    plan, _ = _plan_for(head, DM, ref_freq, (start, stop), variant=variant)
    shape1, lay, byte_range, mask = reader._subset_layout(channels)
    if tuple(shape1) != (plan.nchan, plan.npol):
        # unit axes dropped by squeezing: the remaining one is the channel axis
        keep = [i for i in (0, 1) if shape1[i] != 1]
        if len(keep) != 1 or shape1[keep[0]] != plan.nchan or plan.npol != 1:
            raise ValueError(f"reader samples {tuple(shape1)} do not map onto ({plan.nchan}, {plan.npol}) series")
        if keep[0] == 1:
            lay.update(stride_c=lay["stride_p"])
        lay.update(stride_p=0)
    buf, first = reader._raw.fetch(offset, n, byte_range=byte_range)
    if mask is not None:
        mask = np.asarray(mask).reshape(plan.nchan, plan.npol)
    if detect is not None:
        plan.stream_detect(detect, nscrunch)
        try:
            y, ms = plan.dedisperse_stream_raw(buf, lay, n, first=first, conj=mask, scale=reader._raw.scale)
        finally:
            plan.stream_detect(None)
        return y, start, ms
    y, ms = plan.dedisperse_stream_raw(buf, lay, n, first=first, conj=mask, scale=reader._raw.scale)
    y = y.reshape((len(y),) + tuple(head.shape[1:]))
    return type(head).like(head, y, **_advance(head, start)), ms


def incoherent_dedispersion(z, DM, /, *, ref_freq=None):
    """Incoherently dedisperse a signal: every channel is shifted by its dispersion delay rounded to
    a whole sample, and the result is cropped to the samples all channels cover
    (reference dedispersion.py:136-177).  Device-resident data is gathered by a HIP kernel
    (``pbh_incoherent``); host data keeps the reference's numpy slicing (it is a pure copy).
    """
    if not isinstance(z, RadioSignal):
        raise TypeError("Signal must be a RadioSignal object.")
    if ref_freq is None:
        ref_freq = z.center_freq
    delays = np.asarray(DM.sample_delay(z.channel_freqs, ref_freq, z.sample_rate))
    delays = delays.round().astype(np.int64)
    crop_before = -min(0, delays[0], delays[-1])
    delays = delays + crop_before
    N = len(z) - max(delays)
    if isinstance(z.data, DeviceArray) and not z.data.tensor.is_contiguous() and z.data.series_major_pitch() is not None:
        # series-major (time fastest): every channel is a contiguous run -- one shifted copy per channel instead of
        # the line-granular gather of the sample-major layout
        x = _hip.incoherent_series(z.data, delays, max(N, 0))
    elif isinstance(z.data, DeviceArray):
        x = _hip.incoherent(z.data, delays, max(N, 0))
    else:
        x = np.stack([z.data[j:j + N, i] for i, j in enumerate(delays)], axis=1)
    new_start = z.start_time
    if crop_before and z.start_time is not None:
        new_start = new_start + crop_before * z.dt
    return type(z).like(z, x, start_time=new_start)
