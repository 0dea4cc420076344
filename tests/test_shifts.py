"""time_shift / freq_shift.  CPU: the oracle restatements against the reference's
known answers (tests/test_transforms.py:308-420: impulses land where shifted, sinusoids move to the
target frequency).  GPU: the HIP path against the oracle and the same reference tests."""

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc


def impulse(N, t0):
    """reference tests/test_transforms.py helper: band-limited impulse at (fractional) sample t0."""
    n = (np.arange(N) - N // 2) / N
    x = np.exp(-2j * np.pi * t0 * n)
    return np.fft.ifft(np.fft.ifftshift(x, axes=(-1,))).astype(np.complex128)


def sinusoid(N, f0):
    return np.exp(2j * np.pi * f0 * np.arange(N) / N).astype(np.complex128)


def noise(shape, seed=0):
    r = np.random.default_rng(seed)
    return (r.standard_normal(shape) + 1j * r.standard_normal(shape)).astype(np.complex128)


# ---------------- oracle pins (CPU) ----------------
def test_oracle_time_shift_int():
    x = noise((4096, 4, 2))
    for n in [-12, -3, 4, 13]:
        y, _, _ = orc.time_shift(x, n)
        if n < 0:
            assert np.allclose(y[:n], x[-n:], atol=1e-6) and np.allclose(y[n:], 0)
        else:
            assert np.allclose(y[n:], x[:-n], atol=1e-6) and np.allclose(y[:n], 0)


def test_oracle_time_shift_fractional():
    N, rng = 4096, np.random.default_rng(1)
    shift = rng.uniform(-20, 20, (4, 2))
    x = np.moveaxis(impulse(N, 100 - shift[..., None]), -1, 0)
    y1, start, stop = orc.time_shift(x, shift)
    want = np.zeros_like(y1)
    want[100] = 1.0
    assert np.allclose(y1, want, atol=1e-6)
    y2, _, _ = orc.time_shift(x, shift, crop=True)
    a, b = max(0, int(np.ceil(shift.max()))), N + min(0, int(np.floor(shift.min())))
    assert np.allclose(y2, want[a:b], atol=1e-6)


@pytest.mark.parametrize("N", [1023, 1024])
def test_oracle_freq_shift(N):
    for target in [-50, 0, 50]:
        for f0 in [-200, 0, 100]:
            y = orc.freq_shift(sinusoid(N, f0)[:, None], (target - f0) / N)
            assert np.allclose(y, sinusoid(N, target)[:, None], atol=1e-6)


# ---------------- HIP path (GPU) ----------------
@pytest.mark.gpu
class TestTimeShift:
    @pytest.mark.parametrize("use_complex", [True, False])
    @pytest.mark.parametrize("start_time", [pb.Time(56000.0, format="mjd"), None])
    @pytest.mark.parametrize("device", [False, True])
    def test_int_scalar(self, use_complex, start_time, device):
        """reference tests/test_transforms.py:310-343."""
        kw = dict(sample_rate=1 * u.kHz, start_time=start_time)
        for shape in [(4096, 4, 2), (4096, 4), (4096,)]:
            x = noise(shape).astype(np.complex64)
            z = pb.Signal(x if use_complex else np.ascontiguousarray(x.real), **kw)
            zz = z.to_device() if device else z
            for n in [-12, -7, -3, 0, 4, 8, 13]:
                for s in [n, n * u.ms]:
                    y = pb.time_shift(zz, s)
                    assert type(y.data) is type(zz.data) and y.dtype == z.dtype and y.shape == z.shape
                    if z.start_time is None:
                        assert y.start_time is None
                    else:
                        assert z.start_time.isclose(y.start_time)
                    ya, za = np.array(y), np.array(z)
                    if n < 0:
                        assert np.allclose(ya[:n], za[-n:], atol=2e-5) and np.allclose(ya[n:], 0)
                    elif n > 0:
                        assert np.allclose(ya[n:], za[:-n], atol=2e-5) and np.allclose(ya[:n], 0)
                    else:
                        assert np.allclose(ya, za)

    @pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
    def test_advanced(self, dtype):
        """reference tests/test_transforms.py:345-378 (fractional per-series shifts of impulses)."""
        N, rng = 4096, np.random.default_rng(3)
        for shape in [(4096, 4, 2), (4096, 4), (4096,)]:
            shifts = np.concatenate([rng.uniform(-20, 20, (4,) + shape[1:]), rng.uniform(0, 20, (3,) + shape[1:]),
                                     rng.uniform(-20, 0, (3,) + shape[1:])], axis=0)
            for shift in shifts:
                x = np.moveaxis(impulse(N, 100 - shift[..., None]), -1, 0).astype(dtype)
                z = pb.Signal(x, sample_rate=1 * u.kHz, start_time=pb.Time(56000.0, format="mjd"))
                y1 = pb.time_shift(z, shift, crop=False)
                y2 = pb.time_shift(z, shift, crop=True)
                want = np.zeros_like(x)
                want[100] = 1.0
                a = max(0, int(np.ceil(shift.max())))
                b = len(want) + min(0, int(np.floor(shift.min())))
                tol = 2e-5 if dtype == np.complex64 else 1e-7
                assert np.allclose(np.asarray(y1), want, atol=tol)
                assert np.allclose(np.asarray(y2), want[a:b], atol=tol)
                assert abs((y2.start_time - z.start_time).to_value(u.s) - a / 1e3) < 1e-12
                o1, _, _ = orc.time_shift(x, shift)
                assert np.abs(np.asarray(y1) - o1).max() < tol

    def test_shape_errors(self):
        """reference tests/test_transforms.py:380-392."""
        x = pb.Signal(noise((4096, 4, 2)).astype(np.complex64), sample_rate=1 * u.kHz)
        rng = np.random.default_rng(0)
        for shape in [(1,), (1, 2), (4,), (4, 1), (4, 2)]:
            _ = pb.time_shift(x, rng.uniform(-20, 20, shape))
        for shape in [(2,), (5, 2), (4, 5), (1, 4), (2, 1), (4, 2, 2)]:
            with pytest.raises(ValueError):
                _ = pb.time_shift(x, rng.uniform(-20, 20, shape))


@pytest.mark.gpu
class TestFreqShift:
    @pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
    def test_basic(self, dtype):
        """reference tests/test_transforms.py:395-425 (N = 1024; 1023 needs a non-power-of-two plan)."""
        N = 1024
        for target in [-50, 0, 50]:
            for f0 in [-200, -100, 0, 100, 200]:
                x = pb.BasebandSignal(sinusoid(N, f0)[:, None].astype(dtype), sample_rate=N * u.Hz,
                                      center_freq=1 * u.MHz)
                y = pb.freq_shift(x, (target - f0) * u.Hz)
                assert isinstance(y, pb.BasebandSignal) and isinstance(y.data, np.ndarray)
                assert x.center_freq == y.center_freq and x.freq_align == y.freq_align
                assert x.sample_rate == y.sample_rate and x.start_time == y.start_time
                tol = 3e-5 if dtype == np.complex64 else 1e-9
                assert np.allclose(np.asarray(y.data), sinusoid(N, target)[:, None], atol=tol)

    def test_out_of_band_and_oracle(self):
        x = noise((4096, 3, 2)).astype(np.complex64)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        shift = np.array([100e3, -250e3, 0.0])[:, None] * u.Hz
        y = pb.freq_shift(z, shift)
        want = orc.freq_shift(x, np.array([0.1, -0.25, 0.0])[:, None])
        assert np.abs(np.asarray(y) - want).max() < 3e-5
        assert np.allclose(np.asarray(pb.freq_shift(z, 2 * u.MHz)), 0)       # everything shifted out of band
        with pytest.raises(TypeError):
            pb.freq_shift(pb.Signal(x, sample_rate=1 * u.MHz), 1 * u.kHz)
        with pytest.raises(ValueError):
            pb.freq_shift(z, 5.0)


@pytest.mark.gpu
class TestSnippet:
    """reference tests/test_transforms.py:250-310."""

    @pytest.mark.parametrize("N", [1024, 1023])
    @pytest.mark.parametrize("with_start", [True, False])
    def test_correctness(self, N, with_start):
        from pulsarbat_amd.time import Time
        start_time = Time.now() if with_start else None
        for sr in [1 * u.Hz, 10 * u.Hz]:
            for t0 in [20.0, 10.5, 15.9]:
                x = pb.Signal(impulse(N, t0), sample_rate=sr, start_time=start_time)
                for n in [8, 16, 25]:
                    ts = [t0, t0 * x.dt]
                    if start_time is not None:
                        ts.append(start_time + t0 * x.dt)
                    for t in ts:
                        y = pb.snippet(x, t, n)
                        z = np.zeros_like(np.asarray(y.data))
                        z[0] = 1
                        assert np.allclose(np.asarray(y.data), z, atol=1e-8)
                        assert x.sample_rate == y.sample_rate and len(y) == n
                        if x.start_time is None:
                            assert y.start_time is None
                        else:
                            assert Time.isclose(y.start_time, x.start_time + t0 * x.dt)

    def test_errors(self):
        from pulsarbat_amd.time import Time
        z = pb.Signal(impulse(1024, 512), sample_rate=1 * u.Hz)
        for t, n in [(-100, 100), (-100, -100), (1000, 50), (1000, -50), (2000, 20), (2000, -20)]:
            with pytest.raises(ValueError):
                pb.snippet(z, t, n)
        with pytest.raises(TypeError):
            pb.snippet(z, 0, np.arange(4))
        with pytest.raises(ValueError):
            pb.snippet(z, np.arange(10), 10)
        with pytest.raises(ValueError):
            pb.snippet(z, Time.now(), 10)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,shift", [((1 << 19, 2, 2), 3.37), ((1 << 20, 3), np.array([-7.25, 0.5, 12.0])),
                                         ((1 << 19, 2, 2), -1000.5)])
def test_time_shift_multi_pass_lengths(shape, shift):
    """Lengths beyond one tile: the phase ramp is read as a float32 phase by the fused row pass (uniform shift: one
    shared row) or as per-series rows; noise input against the oracle."""
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
    z = pb.Signal(x, sample_rate=1 * u.kHz)
    want, start, stop = orc.time_shift(x, shift)
    for zz in (z, pb.Signal(pb.DeviceArray.from_host(x), sample_rate=1 * u.kHz)):
        y = np.asarray(pb.time_shift(zz, shift))
        assert np.linalg.norm(y - want) / np.linalg.norm(want) < 2e-6
        yc = np.asarray(pb.time_shift(zz, shift, crop=True))
        assert yc.shape[0] == shape[0] - start + stop
        assert np.linalg.norm(yc - want[start:shape[0] + stop]) / np.linalg.norm(want) < 2e-6
