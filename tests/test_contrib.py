"""contrib.stft / istft.  CPU: the oracle restatement against the reference's
own known answers (tests/test_contrib.py:22-51).  GPU: the HIP path against the oracle and the same
reference tests, for power-of-two and arbitrary nperseg, both dtypes, host and device data."""

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc


def test_oracle_single_tone():
    """reference tests/test_contrib.py:41-51."""
    x = np.exp(2j * np.pi * np.arange(1024) * 0.25)[:, None]
    for n in [32, 64, 512, 1024]:
        y = orc.stft(x, n)
        a = np.zeros_like(y)
        a[:, 3 * n // 4] = 1.0
        assert np.allclose(a, y)


@pytest.mark.parametrize("shape", [(4224, 4, 2), (4233, 3, 2)])
def test_oracle_reversibility(shape):
    """reference tests/test_contrib.py:23-39."""
    x = np.exp(1j * np.random.default_rng(0).uniform(-np.pi, np.pi, shape))
    for n in [33, 32, shape[0]]:
        y = orc.istft(orc.stft(x, n), n)
        assert np.allclose(x[: len(y)], y)


def assert_equal_radiosignals(x, y):
    assert np.allclose(np.array(x), np.array(y), atol=2e-6), f"{x.shape}, {y.shape}"
    assert x.start_time.isclose(y.start_time)
    assert u.isclose(x.sample_rate, y.sample_rate)
    assert np.allclose(x.channel_freqs.to_value(u.Hz), y.channel_freqs.to_value(u.Hz))


@pytest.mark.gpu
class TestSTFT:
    @pytest.mark.parametrize("shape", [(4224, 4, 2), (4233, 3, 2)])
    @pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
    @pytest.mark.parametrize("device", [False, True])
    def test_reversibility(self, shape, dtype, device):
        """reference tests/test_contrib.py:23-39 (dask replaced by device-resident data)."""
        kw = {"sample_rate": 4 * u.MHz, "center_freq": 400 * u.MHz, "pol_type": "linear",
              "start_time": pb.Time.now()}
        x = np.exp(1j * np.random.default_rng(1).uniform(-np.pi, +np.pi, shape)).astype(dtype)
        z = pb.DualPolarizationSignal(x, **kw)
        zz = z.to_device() if device else z
        for n in [33, 32, shape[0]]:
            s = pb.contrib.stft(zz, nperseg=n)
            assert s.shape == (shape[0] // n, shape[1] * n, shape[2]) and s.dtype == dtype
            want = orc.stft(x, n)
            tol = 3e-6 if dtype == np.complex64 else 1e-12
            assert np.abs(np.asarray(s) - want).max() < tol * max(1.0, np.abs(want).max())
            y = pb.contrib.istft(s, nperseg=n)
            assert isinstance(y, type(z))
            assert isinstance(y.data, type(zz.data))
            assert_equal_radiosignals(z[: len(y)], y)

    def test_single_tone(self):
        """reference tests/test_contrib.py:41-51."""
        x = np.exp(2j * np.pi * np.arange(1024) * 0.25)[:, None].astype(np.complex64)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        for n in [32, 64, 512, 1024]:
            y = pb.contrib.stft(z, nperseg=n)
            a = np.zeros_like(y.data)
            a[:, 3 * n // 4] = 1.0
            assert np.allclose(a, y.data, atol=2e-6)

    def test_signal_bookkeeping_and_errors(self):
        x = np.ones((1024, 2), np.complex64)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        y = pb.contrib.stft(z, nperseg=256)
        assert y.sample_rate.to_value(u.Hz) == pytest.approx(1e6 / 256) and y.freq_align == "bottom"
        assert pb.contrib.stft(z, nperseg=5).freq_align == "center"
        assert pb.contrib.stft(z, window="hann") is NotImplemented
        assert pb.contrib.istft(z, noverlap=3) is NotImplemented
        with pytest.raises(ValueError):
            pb.contrib.stft(pb.Signal(x, sample_rate=1 * u.MHz))

    @pytest.mark.parametrize("n", [64, 4096, 16384, 100, 3000])
    def test_large_blocks(self, n):
        x = orc.synthetic_block((n * 7 + 3, 8, 2), n)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        s = pb.contrib.stft(z.to_device(), nperseg=n)
        want = orc.stft(x, n)
        assert np.linalg.norm(np.asarray(s) - want) / np.linalg.norm(want) < 3e-6
        y = pb.contrib.istft(s, nperseg=n)
        assert np.linalg.norm(np.asarray(y) - x[: len(y)]) / np.linalg.norm(x) < 3e-6

    @pytest.mark.parametrize("n,tail,nseg,dtype", [
        (1 << 15, (2, 2), 5, np.complex64), (1 << 16, (1, 2), 3, np.complex64), (1 << 17, (3,), 2, np.complex64),
        (1 << 15, (1, 1), 4, np.complex64), (3 << 19, (1, 2), 2, np.complex64), (1 << 20, (2, 2), 3, np.complex64),
        (1 << 14, (2, 2), 3, np.complex128), (1 << 16, (1, 2), 2, np.complex128),
        (75 << 10, (2, 2), 3, np.complex64), (45 << 12, (1, 2), 2, np.complex64), (27 << 11, (2,), 3, np.complex128),   # 7-smooth: k_colmix
        (2025 << 10, (2,), 2, np.complex64),   # ... with both column levels (N1 = 3 x 675)
        (81000, (2, 2), 3, np.complex64), (234375, (1, 2), 2, np.complex64), (400000, (2,), 2, np.complex128),   # mixed-radix rows (k_rowmix)
    ])
    def test_segments_beyond_one_tile(self, n, tail, nseg, dtype):
        """Native segment lengths longer than a tile: the segments are a batch of multi-pass transforms and one
        pass writes fftshift / scale / the reference's layout (k_stft_out)."""
        rng = np.random.default_rng(n % 977 + nseg)
        shape = (n * nseg + 17,) + tail
        x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        s = pb.contrib.stft(z.to_device(), nperseg=n)
        want = orc.stft(x, n)
        tol = 3e-6 if dtype == np.complex64 else 1e-12
        assert s.shape == want.shape and np.linalg.norm(np.asarray(s) - want) / np.linalg.norm(want) < tol
        # every bin in its place: a tone at bin 5 of channel 0 (fftshift puts it at n/2 + 5)
        t = np.zeros(shape, dtype)
        t[:n * nseg, 0] = np.exp(2j * np.pi * 5 * np.arange(n * nseg) / n).reshape((-1,) + (1,) * (len(tail) - 1))
        st = np.asarray(pb.contrib.stft(pb.BasebandSignal(t, sample_rate=1 * u.MHz, center_freq=1 * u.GHz).to_device(), nperseg=n))
        peak = np.zeros_like(st)
        peak[:, n // 2 + 5] = 1
        assert np.abs(st - peak).max() < (1e-4 if dtype == np.complex64 else 1e-10)
        y = pb.contrib.istft(s, nperseg=n)
        assert np.linalg.norm(np.asarray(y) - x[: len(y)]) / np.linalg.norm(x[: len(y)]) < tol

    @pytest.mark.parametrize("n,dtype,tail", [(16384, np.complex64, (1, 2)), (16384, np.complex64, (3, 4)),
                                              (8192, np.complex128, (1, 2)), (8192, np.complex128, (2, 2))])
    def test_one_segment_per_tile_pairs(self, n, dtype, tail):
        """nperseg = one tile with an even number of inner elements: both series of a pair in one workgroup (k_seg_pair)."""
        rng = np.random.default_rng(3)
        shape = (n * 5 + 11,) + tail
        x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)
        z = pb.BasebandSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        s = pb.contrib.stft(z.to_device(), nperseg=n)
        want = orc.stft(x, n)
        tol = 3e-6 if dtype == np.complex64 else 1e-12
        assert np.linalg.norm(np.asarray(s) - want) / np.linalg.norm(want) < tol
        y = pb.contrib.istft(s, nperseg=n)
        assert np.linalg.norm(np.asarray(y) - x[: len(y)]) / np.linalg.norm(x[: len(y)]) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("shape,nperseg,dm", [
    ((1 << 21, 8, 2), 64, 30.0),     # 16 series per tile: whole input lines, 128-byte series-major runs
    ((1 << 23, 2, 2), 256, 60.0),    # 4 input series: the tile takes all of them
    ((1 << 22, 8, 2), 128, 80.0),    # 8 of the 16 series per tile (sibling tiles share the input's lines)
    ((1 << 21, 8, 2), 512, 40.0),    # 2 of 16 series per tile would be slower than two steps: two steps in the one call
    ((1 << 20, 3, 2), 32, 20.0),     # 6 input series: a subset size that divides them
    ((1 << 22, 2), 128, 10.0),       # single-pol baseband signal
    ((1 << 18, 4, 2), 2048, 5.0),    # beyond the fused kernel's segment lengths: two steps
    ((1 << 16, 2, 2), 64, 1.0),      # channelised block of one tile: two steps
])
def test_stft_dedisperse_fused(shape, nperseg, dm):
    """contrib.stft followed by coherent_dedispersion as one library call (pbh_stft_dedisperse) against the oracle's
    composition of the two reference functions (misc.py:41-55, dedispersion.py:118-133), and against the product's own
    two-call form: values, crop, start_time, sample rate and channel grid."""
    sr, fc = 8e6, 1.3e9
    x = orc.synthetic_block(shape, 17)
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz, start_time=pb.Time(56000.0, format="mjd"))
    z = (pb.DualPolarizationSignal(x, pol_type="linear", **kw) if len(shape) == 3 else pb.BasebandSignal(x, **kw))
    zd = z.to_device()
    y = pb.contrib.stft_dedisperse(zd, pb.DM(dm), nperseg=nperseg)
    ch = orc.stft(x, nperseg)
    want, start, stop = orc.coherent_dedispersion(ch, dm, sr / nperseg, fc, freq_align="bottom")
    assert want.shape[0] > 0
    assert isinstance(y.data, pb.DeviceArray) and type(y) is type(z) and y.shape == want.shape
    got = np.asarray(y).reshape(want.shape[0], -1)
    ref = want.reshape(want.shape[0], -1)
    err = np.linalg.norm(got - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert err.max() < 1e-5, f"per-series relative L2 {err.max():.2e}"
    two = pb.coherent_dedispersion(pb.contrib.stft(zd, nperseg=nperseg), pb.DM(dm))
    assert_equal_radiosignals(y, two)
    assert abs((y.start_time - z.start_time).to_value(u.s) - start * nperseg / sr) < 1e-12
    # host signals take the two-step route and agree
    if shape[0] <= 1 << 20:
        yh = pb.contrib.stft_dedisperse(z, pb.DM(dm), nperseg=nperseg)
        assert isinstance(yh.data, np.ndarray) and np.allclose(np.asarray(yh), np.asarray(y), atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,nperseg,dm", [
    ((1 << 21, 8, 2), 64, 30.0),     # 16 output series per tile: whole output lines
    ((1 << 23, 2, 2), 256, 60.0),    # 4 output series: the tile takes all of them
    ((1 << 22, 8, 2), 128, 80.0),    # 8 of the 16 series per tile: 64-byte runs
    ((1 << 21, 8, 2), 512, 40.0),    # 2 of 16 series per tile: two steps in the one call
    ((1 << 20, 3, 2), 32, 20.0),     # 6 output series: a subset size that divides them; the last tile of segments is short
    ((1 << 22, 2), 128, 10.0),       # single-pol baseband signal
    ((1 << 18, 4, 2), 2048, 5.0),    # beyond the fused kernel's segment lengths: two steps
    ((1 << 16, 2, 2), 64, 1.0),      # channelised block of one tile: two steps
])
def test_dedisperse_istft_fused(shape, nperseg, dm):
    """coherent_dedispersion followed by contrib.istft as one library call (pbh_dedisperse_istft) against the oracle's
    composition of the two reference functions (dedispersion.py:118-133, misc.py:58-93) and against the product's own
    two-call form.  The input is the channelised block the oracle's stft makes of a seeded time series."""
    sr, fc = 8e6, 1.3e9
    x = orc.synthetic_block(shape, 19)
    ch = np.ascontiguousarray(orc.stft(x, nperseg)).astype(np.complex64)
    kw = dict(sample_rate=sr / nperseg * u.Hz, center_freq=fc * u.Hz, freq_align="bottom",
              start_time=pb.Time(56000.0, format="mjd"))
    zc = (pb.DualPolarizationSignal(ch, pol_type="linear", **kw) if len(shape) == 3 else pb.BasebandSignal(ch, **kw))
    zd = zc.to_device()
    y = pb.contrib.dedisperse_istft(zd, pb.DM(dm), nperseg=nperseg)
    mid, start, stop = orc.coherent_dedispersion(ch, dm, sr / nperseg, fc, freq_align="bottom")
    want = orc.istft(mid, nperseg)
    assert want.shape[0] > 0
    assert isinstance(y.data, pb.DeviceArray) and type(y) is type(zc) and y.shape == want.shape
    got = np.asarray(y).reshape(want.shape[0], -1)
    ref = want.reshape(want.shape[0], -1)
    err = np.linalg.norm(got - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert err.max() < 1e-5, f"per-series relative L2 {err.max():.2e}"
    two = pb.contrib.istft(pb.coherent_dedispersion(zd, pb.DM(dm)), nperseg=nperseg)
    assert_equal_radiosignals(y, two)
    assert abs((y.start_time - zc.start_time).to_value(u.s) - start * nperseg / sr) < 1e-12
    # a series-major channelised block (what stft_dedisperse(..., out_layout="series") style pipelines keep) gives the same
    if len(shape) == 3 and zd.data.tensor.is_contiguous() and shape[0] // nperseg > 1 << 14:
        zs = type(zd).like(zd, zd.data.to_series_major())
        ys = pb.contrib.dedisperse_istft(zs, pb.DM(dm), nperseg=nperseg)
        assert np.allclose(np.asarray(ys), np.asarray(y), atol=2e-6)   # (other layout passes in front: same values to rounding)
    if shape[0] <= 1 << 20:   # host signals take the two-step route and agree
        yh = pb.contrib.dedisperse_istft(zc, pb.DM(dm), nperseg=nperseg)
        assert isinstance(yh.data, np.ndarray) and np.allclose(np.asarray(yh), np.asarray(y), atol=2e-6)
