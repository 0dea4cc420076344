"""CPU ORACLE for the coherent-dedispersion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy/scipy.fft restatement of the reference algorithm
(theXYZT/pulsarbat @ v0.0.10-dev1).  It exists to CHECK the HIP path; it is
never the thing measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``pulsarbat_amd``) must not import anything under ``oracle/``.

Why a restatement and not the reference itself
----------------------------------------------
The reference is pure Python whose arithmetic lives in third-party wheels:
``scipy.fft`` (pocketfft), ``numpy`` ufuncs and ``astropy.units`` for unit
scaling (reference ``setup.py:8-12`` pins numpy>=1.26, scipy>=1.13,
astropy>=6.1, dask>=2024.5).  numpy 2.2.6 and scipy 1.15.3 are present here and
satisfy the pins, so this oracle calls *the same* FFT and ufunc routines the
reference calls.  astropy/dask are absent (ordinary ModuleNotFoundError, not a
permission denial), so ``import pulsarbat`` fails and astropy's unit algebra
is replaced below by explicit SI constants.

Pinning
-------
The reference ships no stored golden vectors for this path (all of its tests
are known-answer scalars or properties).  The oracle is pinned against every
known-answer / property test the reference holds for the path, in
``tests/test_oracle.py``:
  * tests/test_dedispersion.py:13-32   DM=2.41e-4 delay table (known answers)
  * tests/test_dedispersion.py:73-98   reversibility, seeds {4,8,15,16,23,42}, atol 3e-8
  * tests/test_dedispersion.py:100-139 Gabor-wavelet re-alignment (physics known answer)
  * tests/test_dedispersion.py:141-164 precomputed chirp (2-D and 3-D) equivalence
  * tests/test_dedispersion.py:36-71   crop length / start offset inequalities
  * tests/test_polarization.py:38-48   hand-computed Stokes vectors
  * tests/test_radio_signal.py:142-172 to_intensity dtype c64->f32, c128->f64

Every function cites the reference file:line it follows (paths relative to
the reference root).
"""

import math

import numpy as np
import scipy.fft

# pulsarbat/transforms/dedispersion.py:30
#   dispersion_constant = u.s * u.MHz**2 * u.cm**3 / u.pc / 2.41e-4
DISPERSION_CONSTANT = 1.0 / 2.41e-4  # s MHz^2 cm^3 / pc


def time_delay(dm, f_hz, ref_hz):
    """pulsarbat/transforms/dedispersion.py:32-36 (DispersionMeasure.time_delay).

    delay = D * DM * (1/f^2 - 1/ref^2), returned in seconds.  ``f_hz`` / ``ref_hz``
    may be numpy.inf (tests/test_dedispersion.py:17-21) or arrays.
    """
    f_mhz = np.asarray(f_hz, dtype=np.float64) / 1e6
    r_mhz = np.asarray(ref_hz, dtype=np.float64) / 1e6
    with np.errstate(divide="ignore"):
        return DISPERSION_CONSTANT * dm * (1.0 / f_mhz ** 2 - 1.0 / r_mhz ** 2)


def sample_delay(dm, f_hz, ref_hz, sample_rate_hz):
    """pulsarbat/transforms/dedispersion.py:38-42 (DispersionMeasure.sample_delay)."""
    return time_delay(dm, f_hz, ref_hz) * sample_rate_hz


def channel_freqs(center_freq_hz, chan_bw_hz, nchan, freq_align="center"):
    """pulsarbat/core.py:569-574 (RadioSignal.channel_freqs), with the odd-nchan
    rule of pulsarbat/core.py:561-567 (freq_align forced to 'center')."""
    if freq_align not in {"bottom", "center", "top"}:
        raise ValueError("Invalid freq_align.")
    if nchan % 2:
        freq_align = "center"
    _align = {"bottom": 0, "center": 0.5, "top": 1}[freq_align]
    chan_ids = np.arange(nchan) + _align - nchan / 2
    return center_freq_hz + chan_bw_hz * chan_ids


def band_edges(center_freq_hz, chan_bw_hz, nchan):
    """pulsarbat/core.py:546-554 (max_freq / min_freq): returns (min, max)."""
    bw = chan_bw_hz * nchan
    return center_freq_hz - bw / 2, center_freq_hz + bw / 2


def transfer_function(dm, N, dt_s, center_freq_hz, ref_freq_hz):
    """pulsarbat/transforms/dedispersion.py:19-23 (_transfer_function).

    f     = center_freq + fftfreq(N, dt)                      [Hz]   (:20)
    phase = coeff * f * cycle * (1/ref_freq - 1/f)**2         [cycle] (:21)
    tf    = exp(-1j * phase[rad]).astype(complex64)                    (:22-23)
    with coeff = dispersion_constant * DM (:46) = D*DM s MHz^2 = D*DM*1e12 s Hz^2.
    All float64 until the final cast, as in the reference.
    """
    coeff = DISPERSION_CONSTANT * dm * 1e12  # s Hz^2
    f = center_freq_hz + np.fft.fftfreq(N, dt_s)
    phase = coeff * f * (1.0 / ref_freq_hz - 1.0 / f) ** 2  # cycles
    tf = np.exp(-1j * (2.0 * np.pi * phase))
    return tf.astype(np.complex64)


def phase_cycles(dm, N, dt_s, center_freq_hz, ref_freq_hz, bins):
    """f64 phase (cycles) of transfer_function at selected FFT bins (for spot checks)."""
    coeff = DISPERSION_CONSTANT * dm * 1e12
    f = center_freq_hz + np.fft.fftfreq(N, dt_s)[np.asarray(bins)]
    return coeff * f * (1.0 / ref_freq_hz - 1.0 / f) ** 2


def chirp_from_signal(dm, shape, sample_rate_hz, center_freq_hz, freq_align="center",
                      ref_freq_hz=None):
    """pulsarbat/transforms/dedispersion.py:59-75 (chirp_from_signal).

    One transfer function per channel frequency (:70-73), stacked on axis 1 and
    given a length-1 axis for every signal axis >= 2 (:64, :75).
    """
    N, nchan = shape[0], shape[1]
    ndim = len(shape)
    if ref_freq_hz is None:
        ref_freq_hz = center_freq_hz  # :67-68
    dt = 1.0 / sample_rate_hz
    ix = tuple(slice(None) if i < 2 else None for i in range(ndim))  # :64
    freqs = channel_freqs(center_freq_hz, sample_rate_hz, nchan, freq_align)
    chirps = [transfer_function(dm, N, dt, f, ref_freq_hz) for f in freqs]
    return np.stack(chirps, axis=1)[ix]


def crop_bounds(dm, N, nchan, sample_rate_hz, center_freq_hz, ref_freq_hz):
    """pulsarbat/transforms/dedispersion.py:127-131: (start, stop)."""
    fmin, fmax = band_edges(center_freq_hz, sample_rate_hz, nchan)
    delay_top = float(sample_delay(dm, fmax, ref_freq_hz, sample_rate_hz))
    delay_bot = float(sample_delay(dm, fmin, ref_freq_hz, sample_rate_hz))
    start = math.ceil(-min(0, delay_top, delay_bot))
    stop = N - math.ceil(+max(0, delay_top, delay_bot))
    return start, stop


def coherent_dedispersion(x, dm, sample_rate_hz, center_freq_hz, freq_align="center",
                          ref_freq_hz=None, chirp=None, workers=None):
    """pulsarbat/transforms/dedispersion.py:81-133 (coherent_dedispersion).

    ``x`` is the BasebandSignal data, shape (nsample, nchan, ...), complex64 or
    complex128 (pulsarbat/core.py:742).  Returns (y, start, stop) where ``y`` is
    the cropped result ``x_full[start:stop]`` (:133) and the signal's start_time
    advances by start/sample_rate (pulsarbat/core.py:155-164).

    ``workers=None`` is what the reference executes (pulsarbat/fft.py:36-38 passes
    nothing, so pocketfft runs single-threaded).
    """
    x = np.asarray(x)
    if x.dtype not in (np.complex64, np.complex128):
        raise TypeError("BasebandSignal data must be complex64/complex128 (core.py:742)")
    if ref_freq_hz is None:
        ref_freq_hz = center_freq_hz  # :118-119
    if chirp is None:
        chirp = chirp_from_signal(dm, x.shape, sample_rate_hz, center_freq_hz,
                                  freq_align, ref_freq_hz)  # :121-122
    chirp = chirp[(slice(None),) * chirp.ndim + (None,) * (x.ndim - chirp.ndim)]  # :124
    y = scipy.fft.ifft(scipy.fft.fft(x, axis=0, workers=workers) * chirp, axis=0,
                       workers=workers)  # :125
    start, stop = crop_bounds(dm, x.shape[0], x.shape[1], sample_rate_hz,
                              center_freq_hz, ref_freq_hz)  # :127-131
    return y[start:stop], start, stop  # :133


def to_intensity(z):
    """pulsarbat/core.py:766-774 (BasebandSignal.to_intensity)."""
    return z.real ** 2 + z.imag ** 2


def to_stokes(z, pol_type):
    """pulsarbat/core.py:930-966 (DualPolarizationSignal.to_stokes); pol axis = 2."""
    A = np.take(z, 0, axis=2)
    B = np.take(z, 1, axis=2)
    AA = A.real ** 2 + A.imag ** 2
    BB = B.real ** 2 + B.imag ** 2
    AB = A.conj() * B
    if pol_type == "linear":  # :941-951
        i, Q, U, V = AA + BB, AA - BB, 2 * AB.real, 2 * AB.imag
    elif pol_type == "circular":  # :953-963
        i, Q, U, V = AA + BB, 2 * AB.real, 2 * AB.imag, AA - BB
    else:
        raise ValueError("pol_type must be in {'linear', 'circular'}")
    return np.stack([i, Q, U, V], axis=2)


def scrunch(a, nscrunch):
    """Time-scrunch: sum of ``nscrunch`` consecutive time samples, tail dropped.

    The reference has NO such function; the build defines it
    as ``a[:n*k].reshape(n, k, ...).sum(1)`` in the array's own dtype.
    """
    n = a.shape[0] // nscrunch
    return a[: n * nscrunch].reshape((n, nscrunch) + a.shape[1:]).sum(axis=1)


def detect_f64(z, mode, nscrunch=1, pol_type="linear"):
    """Detection (+ time scrunch) of dedispersed voltages with the POWER SUMS IN FLOAT64 -- the yardstick the detect tail's
    parity tests use ("Stokes/scrunch outputs <= 1e-5 relative").  The reference computes to_intensity /
    to_stokes in the data's own precision (core.py:766-774, 930-966) and has no scrunch; summing 1024 float32 powers one
    after the other is itself ~1e-5 noisy, so a float32 oracle cannot tell a 1e-5 error of the device path from its own.
    ``mode``: "intensity" (every series, |z|^2), "I" (Stokes I per channel) or "stokes" (I, Q, U, V on a new last axis).
    Returns (values float64, scale float64): ``scale`` is the Stokes-I (or |z|^2) sum the absolute errors are measured
    against -- Q, U and V change sign, only I is an all-positive sum."""
    z = np.asarray(z).astype(np.complex128)
    if mode == "intensity":
        v = to_intensity(z)
        sc = v
    else:
        st = to_stokes(z, pol_type)
        v = st[:, :, 0] if mode == "I" else st
        sc = st[:, :, 0] if mode == "I" else st[:, :, :1]
    if nscrunch > 1:
        v, sc = scrunch(v, nscrunch), scrunch(sc, nscrunch)
    return v, np.broadcast_to(sc, v.shape)


def synthetic_block(shape, seed):
    """SURVEY.md 8(d) synthetic input: complex standard normal / sqrt(2), complex64."""
    rng = np.random.default_rng(seed)
    re = rng.standard_normal(shape, dtype=np.float32)
    im = rng.standard_normal(shape, dtype=np.float32)
    return ((re + 1j * im) * np.float32(2 ** -0.5)).astype(np.complex64)


def stft(x, nperseg):
    """pulsarbat/contrib/misc.py:41-52 (stft) on the data array: trim, reshape, swap, fft, fftshift,
    reshape, / nperseg.  ``x`` is (nsample, nchan, ...)."""
    n = int(nperseg)
    x = x[: len(x) - len(x) % n]
    y = x.reshape((-1, n) + x.shape[1:]).swapaxes(1, 2)
    y = scipy.fft.fft(y, axis=2, n=n)
    y = np.fft.fftshift(y, axes=(2,))
    y = y.reshape((y.shape[0], -1) + y.shape[3:])
    return y / n


def istft(y, nperseg):
    """pulsarbat/contrib/misc.py:80-91 (istft) on the data array ``y`` (nseg, nchan*nperseg, ...)."""
    n = int(nperseg)
    x = y.reshape((len(y), -1, n) + y.shape[2:]) * n
    x = x.swapaxes(1, 2)
    x = np.fft.ifftshift(x, axes=(1,))
    x = scipy.fft.ifft(x, axis=1, n=n)
    x = x[:, :n]
    return x.reshape((-1,) + x.shape[2:])


def time_shift(x, shift, crop=False):
    """pulsarbat/transforms/transforms.py:248-293 (time_shift) on the data array.  ``shift`` (samples)
    broadcasts over the sample shape; returns (shifted, start, stop)."""
    shift = np.array(shift, dtype=np.float64)
    if shift.ndim > 0:
        shift = shift[(slice(None),) * shift.ndim + (None,) * (x.ndim - shift.ndim - 1)]
    f = np.fft.fftfreq(len(x), 1)[tuple(slice(None) if j == 0 else None for j in range(x.ndim))]
    ph = np.exp(-2j * np.pi * shift * f).astype(np.complex64)
    shifted = scipy.fft.ifft(scipy.fft.fft(x, axis=0) * ph, axis=0)
    shifted = shifted if np.iscomplexobj(x) else shifted.real
    shifted = np.array(shifted)
    start, stop = 0, 0
    full = np.broadcast_to(shift, x.shape[1:]) if x.ndim > 1 else shift.reshape(())
    it = np.nditer(full, flags=["multi_index"])
    for a in it:
        if a < 0:
            a = int(np.floor(a))
            shifted[(np.s_[a:],) + it.multi_index] = 0
            stop = min(stop, a)
        else:
            a = int(np.ceil(a))
            shifted[(np.s_[:a],) + it.multi_index] = 0
            start = max(start, a)
    if crop:
        shifted = shifted[start:len(shifted) + stop]
    return shifted, start, stop


def freq_shift(x, ft):
    """pulsarbat/transforms/transforms.py:337-361 (freq_shift) on the data array; ``ft`` = shift * dt
    (cycles per sample), broadcasting over the sample shape."""
    ft = np.array(ft, dtype=np.float64)
    if ft.ndim == 0:
        ft = ft[None]
    ft = ft[(slice(None),) * ft.ndim + (None,) * (x.ndim - ft.ndim - 1)]
    n = np.arange(len(x))[tuple(slice(None) if j == 0 else None for j in range(x.ndim))]
    ph = np.exp(2j * np.pi * ft * n).astype(x.dtype)
    X = np.fft.fftshift(scipy.fft.fft(x * ph, axis=0), axes=(0,))
    full = np.broadcast_to(ft * len(X), X.shape[1:])
    it = np.nditer(full, flags=["multi_index"])
    for a in it:
        if a < 0:
            X[(np.s_[int(np.floor(a)):],) + it.multi_index] = 0
        else:
            X[(np.s_[:int(np.ceil(a))],) + it.multi_index] = 0
    return scipy.fft.ifft(np.fft.ifftshift(X, axes=(0,)), axis=0)


def incoherent_dedispersion(x, dm, sample_rate_hz, center_freq_hz, chan_bw_hz, freq_align="center",
                            ref_freq_hz=None):
    """pulsarbat/transforms/dedispersion.py:136-177 on the data array; returns (y, crop_before)."""
    if ref_freq_hz is None:
        ref_freq_hz = center_freq_hz
    freqs = channel_freqs(center_freq_hz, chan_bw_hz, x.shape[1], freq_align)
    delays = np.asarray(sample_delay(dm, freqs, ref_freq_hz, sample_rate_hz)).round().astype(np.int64)
    crop_before = -min(0, delays[0], delays[-1])
    delays = delays + crop_before
    N = len(x) - max(delays)
    y = np.stack([x[j:j + N, i] for i, j in enumerate(delays)], axis=1)
    return y, int(crop_before)


def real_to_complex(z, axis=0):
    """pulsarbat/utils.py:38-65 (real_to_complex)."""
    z = np.asarray(z)
    if np.iscomplexobj(z):
        raise ValueError("Input must be real-valued.")
    out_dtype = np.complex64 if z.dtype == np.float32 else np.complex128
    N = z.shape[axis]
    if N == 0:
        return z.astype(out_dtype)
    ind = [np.newaxis] * z.ndim
    ind[axis] = slice(None)
    h = np.zeros(N, dtype=out_dtype)
    h[0] = 1
    h[1:N // 2] = 2
    if N > 1:
        h[N // 2] = 2 if N % 2 else 1
    z = scipy.fft.ifft(scipy.fft.fft(z, axis=axis) * h[tuple(ind)], axis=axis)
    z *= np.exp(-1j * np.pi / 2 * np.arange(N))[tuple(ind)]
    dec = [slice(None)] * z.ndim
    dec[axis] = slice(None, None, 2)
    return z[tuple(dec)].astype(out_dtype)
