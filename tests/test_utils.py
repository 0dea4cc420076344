"""utils: real_to_complex on the HIP pipeline; fast-length helpers on the host.
Mirrors reference tests/test_utils.py."""

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from pulsarbat_amd.utils import next_fast_len, prev_fast_len, real_to_complex
from oracle import dedisp_oracle as orc

FAST = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 18, 20, 21, 24, 25, 27, 28, 30, 32, 35, 36, 40, 42, 45,
        48, 49, 50, 54, 56, 60, 63, 64, 70, 72, 75, 80, 81, 84, 90, 96, 98, 100]


def test_fast_len():
    """reference tests/test_utils.py (7-smooth tables)."""
    for i in range(1, 101):
        assert next_fast_len(i) == min(f for f in FAST if f >= i)
        assert prev_fast_len(i) == max(f for f in FAST if f <= i)
    assert next_fast_len(1001) == 1008 and prev_fast_len(1001) == 1000
    z = pb.Signal(np.zeros((1001, 2)), sample_rate=1 * u.Hz)
    assert len(pb.fast_len(z)) == 1000


def test_native_len():
    from pulsarbat_amd.utils import next_native_len, prev_native_len
    assert next_native_len(1) == 32 and next_native_len(33) == 64 and prev_native_len(100) == 96   # 96 = 3 * 2^5
    # 7-smooth lengths with one mixed-radix column pass count as native: 625 * 2^14 and 600 * 2^14 bracket 10^7
    assert next_native_len(10_000_000) == 625 << 14 and prev_native_len(10_000_000) == 600 << 14
    assert next_native_len((3 << 20) - 5) == 3 << 20 and prev_native_len(3 << 20) == 3 << 20
    assert prev_native_len(7 << 24) == 7 << 24 and next_native_len((7 << 24) + 1) == 1 << 27
    with pytest.raises(ValueError):
        prev_native_len(8)


def gen_input(t, w, p):
    return np.cos(w * t + p)


def prediction(t, w, p):
    return np.exp(1j * ((w - (len(t) / 4)) * t[::2] + p))


def test_oracle_theoretical():
    """reference tests/test_utils.py:27-35 on the oracle restatement."""
    for N in [511, 512]:
        t = np.linspace(0, 2 * np.pi, N, endpoint=False)
        for w in [1, 2, 127, 128, 129, 254, 255]:
            for p in [-np.pi, 0, np.pi / 2]:
                assert np.allclose(prediction(t, w, p), orc.real_to_complex(gen_input(t, w, p)))


def test_bad_args_and_empty():
    with pytest.raises(ValueError):
        real_to_complex(np.ones((128, 4), dtype=complex), axis=0)
    for sample_shape in [(), (2,), (4, 4)]:
        x = np.zeros((0,) + sample_shape)
        assert np.array_equal(x, real_to_complex(x, axis=0))


@pytest.mark.gpu
class TestRealToComplex:
    def test_theoretical(self):
        """reference tests/test_utils.py:27-35 (N = 511 goes through the Bluestein plan)."""
        for N in [511, 512]:
            t = np.linspace(0, 2 * np.pi, N, endpoint=False)
            for w in [1, 2, 127, 128, 129, 254, 255]:
                for p in [-np.pi, -np.pi / 2, 0, np.pi / 2]:
                    z = real_to_complex(gen_input(t, w, p))
                    assert np.allclose(prediction(t, w, p), z, atol=1e-9)

    def test_axis(self):
        """reference tests/test_utils.py:43-56."""
        N = 128
        t = np.linspace(0, 2 * np.pi, N, endpoint=False)
        ws = [1, 2, 3, 4]
        x = np.stack([gen_input(t, w, 0) for w in ws], axis=0)
        y = np.stack([prediction(t, w, 0) for w in ws], axis=0)
        assert np.allclose(y, real_to_complex(x, axis=1), atol=1e-9)
        x = np.stack([gen_input(t, w, 0) for w in ws], axis=1)
        y = np.stack([prediction(t, w, 0) for w in ws], axis=1)
        assert np.allclose(y, real_to_complex(x, axis=0), atol=1e-9)

    def test_dtype_and_oracle(self):
        """reference tests/test_utils.py:63-68, plus parity with the oracle on noise and device data."""
        rng = np.random.default_rng(0)
        for in_type, out_type, tol in [(np.float32, np.complex64, 3e-6), (np.float64, np.complex128, 1e-12)]:
            x = rng.standard_normal((4096, 3, 2)).astype(in_type)
            y = real_to_complex(x)
            assert y.dtype == out_type and y.shape == (2048, 3, 2)
            want = orc.real_to_complex(x)
            assert np.abs(y - want).max() < tol * np.abs(want).max()
            yd = real_to_complex(pb.DeviceArray.from_host(x))
            assert isinstance(yd, pb.DeviceArray) and np.abs(np.asarray(yd) - want).max() < tol * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,axis", [((1 << 16, 3), 0), ((1 << 17,), 0), ((1 << 18, 4, 2), 0), ((5, 1 << 16), 1),
                                        ((1 << 21, 16), 0), ((1 << 25, 1), 0), ((1 << 16, 70), 0)])
def test_real_to_complex_half_length(shape, axis, monkeypatch):
    """Device float32 data whose half length is a power of two beyond one tile run ``pbh_real_to_complex``: the real series,
    time fastest, is the complex series x[2m] + i x[2m+1], and two HALF-length transforms with one mirror pass in between
    give the reference's analytic-signal / mix / decimate result (utils.py:38-65).  Against the oracle, and against the
    product's full-length route (``PBH_R2C_HALF=0``)."""
    from pulsarbat_amd import _hip
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape, dtype=np.float32)
    xd = pb.DeviceArray.from_host(x)
    calls = []
    orig = _hip.real_to_complex_half
    monkeypatch.setattr(_hip, "real_to_complex_half", lambda a: calls.append(orig(a)) or calls[-1])
    y = real_to_complex(xd, axis=axis)
    assert calls and calls[-1] is not None, "the half-length path did not run"
    assert isinstance(y, pb.DeviceArray) and y.dtype == np.complex64
    if x.size <= 1 << 23:
        want = orc.real_to_complex(x.astype(np.float64), axis=axis)
        assert y.shape == want.shape
        err = np.linalg.norm(np.asarray(y) - want) / np.linalg.norm(want)
        assert err < 1e-5, f"relative L2 {err:.2e}"
        assert np.abs(np.asarray(y) - want).max() < 2e-5 * np.abs(want).max()
    monkeypatch.setenv("PBH_R2C_HALF", "0")
    y0 = real_to_complex(xd, axis=axis)
    assert y0.shape == y.shape
    assert np.linalg.norm(np.asarray(y) - np.asarray(y0)) / np.linalg.norm(np.asarray(y0)) < 2e-6


@pytest.mark.gpu
def test_shifts_any_length():
    """time_shift / freq_shift at lengths that are not powers of two (reference uses N = 1023)."""
    from tests.test_shifts import impulse, sinusoid
    N = 1023
    for target in [-50, 0, 50]:
        for f0 in [-200, 0, 100]:
            x = pb.BasebandSignal(sinusoid(N, f0)[:, None], sample_rate=N * u.Hz, center_freq=1 * u.MHz)
            y = pb.freq_shift(x, (target - f0) * u.Hz)
            assert np.allclose(np.asarray(y.data), sinusoid(N, target)[:, None], atol=1e-8)
    N = 3000
    shift = np.array([3.25, -7.5])
    x = np.moveaxis(impulse(N, 100 - shift[..., None]), -1, 0)
    y = pb.time_shift(pb.Signal(x, sample_rate=1 * u.kHz), shift)
    want = np.zeros_like(x)
    want[100] = 1.0
    assert np.allclose(np.asarray(y), want, atol=1e-8)
