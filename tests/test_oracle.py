"""Pins the CPU oracle against every known-answer / property test the reference
holds for the hot path.  Test bodies follow the reference's own
tests (cited per test), with astropy units replaced by floats in Hz."""

import numpy as np
import pytest
import scipy.signal

from oracle import dedisp_oracle as orc

MHz = 1e6
GHz = 1e9


class TestDispersionMeasure:
    def test_basic(self):
        """reference tests/test_dedispersion.py:13-32 (known answers)."""
        dm = 2.41e-4
        for f in [0.1, 1.0, 10.0]:
            assert np.isclose(orc.time_delay(dm, f * MHz, np.inf), 1 / f / f)
            assert np.isclose(orc.time_delay(dm, np.inf, f * MHz), -(1 / f / f))
        assert np.isclose(orc.time_delay(dm, 2 * MHz, 1 * MHz), -0.75)
        for sr in [1 * MHz, 10 * MHz, 1e3]:
            assert np.isclose(orc.sample_delay(dm, 1 * MHz, np.inf, sr), sr * 1.0)
        for a in [10, 20, 100]:
            assert np.isclose(orc.time_delay(2.41e-4 * a, 1 * MHz, np.inf), a)


class TestCoherentDedispersion:
    @pytest.mark.parametrize("dm", [10.0, 50.0, 100.0])
    def test_basic(self, dm):
        """reference tests/test_dedispersion.py:36-71 (lengths, start offsets)."""
        shape = (8192, 4)
        fcen, sr = 1 * GHz, 1 * MHz
        rng = np.random.default_rng(int(dm))
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        fmin, fmax = orc.band_edges(fcen, sr, shape[1])
        for ref in [fmin, fcen, fmax]:
            y, start, stop = orc.coherent_dedispersion(x, dm, sr, fcen, ref_freq_hz=ref)
            assert len(y) == stop - start
            assert len(x) - len(y) >= orc.sample_delay(dm, fmin, fmax, sr)
            assert start >= orc.sample_delay(dm, ref, fmax, sr)

    @pytest.mark.parametrize("seed", [4, 8, 15, 16, 23, 42])
    def test_reversibility(self, seed):
        """reference tests/test_dedispersion.py:73-98 (atol 3e-8, complex128 data)."""
        ref, sr, dm = 600 * MHz, 400 * MHz, 0.01
        N, M = 2 ** 18, 2 ** 12
        R = np.random.default_rng(seed=seed)
        x = R.standard_normal(N) + 1j * R.standard_normal(N)
        x *= np.exp(-(((np.arange(N) - N // 2) / M) ** 2))
        sos = scipy.signal.butter(10, 0.45, "lowpass", fs=1.0, output="sos")
        x = scipy.signal.sosfilt(sos, x).reshape(-1, 1)

        temp, s1, _ = orc.coherent_dedispersion(x, dm, sr, ref)
        sig2, s2, _ = orc.coherent_dedispersion(temp, -dm, sr, ref)
        noffset = s1 + s2
        sig1 = x[noffset:noffset + len(sig2)]
        assert np.allclose(sig1 - sig2, 0, atol=3e-8)

    @pytest.mark.parametrize("dm", [0.01, 0.02])
    def test_correctness(self, dm):
        """reference tests/test_dedispersion.py:100-139 (Gabor wavelets re-align)."""
        ref, sr = 600 * MHz, 400 * MHz
        index, N, width = 100000, 2 ** 18, 256
        t = np.arange(N) / sr
        t0 = t[index]
        x = np.zeros(N, dtype=np.complex128)
        for df in np.linspace(-3 * sr / 8, 3 * sr / 8, 13):
            dt = orc.time_delay(dm, ref + df, ref)
            tt = t - (t0 + dt)
            x += np.exp(2j * np.pi * tt * df - (tt / (width / sr)) ** 2)
        y, noffset, _ = orc.coherent_dedispersion(x.reshape(-1, 1), dm, sr, ref)
        id1, id2 = index - noffset - 8 * width, index - noffset + 8 * width
        p1 = (np.abs(x) ** 2).sum()
        p2 = (np.abs(y[id1:id2]) ** 2).sum()
        assert np.allclose(p1, p2)
        assert np.allclose(y[id2:], 0)
        assert np.allclose(y[:id1], 0)

    @pytest.mark.parametrize("dm", [10, 20, 50])
    def test_chirp(self, dm):
        """reference tests/test_dedispersion.py:141-164 (2-D and 3-D precomputed chirp)."""
        shape = (8192, 4, 2)
        rng = np.random.default_rng(dm)
        x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        sr, fc = 1 * MHz, 1 * GHz
        chirp = orc.chirp_from_signal(dm, shape, sr, fc)
        assert chirp.shape == (8192, 4, 1) and chirp.dtype == np.complex64
        y1, _, _ = orc.coherent_dedispersion(x, dm, sr, fc)
        y2, _, _ = orc.coherent_dedispersion(x, dm, sr, fc, chirp=chirp)
        y3, _, _ = orc.coherent_dedispersion(x, dm, sr, fc, chirp=chirp.squeeze())
        assert np.allclose(y1, y2) and np.allclose(y1, y3)
        fmin, fmax = orc.band_edges(fc, sr, 4)
        for rf in [fmin, fmax]:
            chirp = orc.chirp_from_signal(dm, shape, sr, fc, ref_freq_hz=rf)
            y1, _, _ = orc.coherent_dedispersion(x, dm, sr, fc, ref_freq_hz=rf)
            y2, _, _ = orc.coherent_dedispersion(x, dm, sr, fc, ref_freq_hz=rf, chirp=chirp)
            assert np.allclose(y1, y2)


def test_config_geometry():
    """crop indices of BASELINE.json configs 2, 3 and 5."""
    N = 2 ** 24
    assert orc.crop_bounds(56.77, N, 8, 50 * MHz, 1.4 * GHz, 1.4 * GHz) == (1408404, 14607231)
    assert orc.crop_bounds(56.77, N, 64, 6.25 * MHz, 1.4 * GHz, 1.4 * GHz) == (176051, 16505967)
    assert orc.crop_bounds(1000.0, N, 64, 6.25 * MHz, 1.4 * GHz, 1.4 * GHz) == (3101118, 11999198)
    f = orc.channel_freqs(1.4 * GHz, 50 * MHz, 8)
    assert np.allclose(f, np.arange(1225, 1600, 50) * MHz)


def test_config1_identity():
    """BASELINE.json config 1: DM=0 => chirp == 1, no crop, output == input."""
    x = orc.synthetic_block((2 ** 20, 1, 1), 20260001)
    chirp = orc.chirp_from_signal(0.0, x.shape, 400 * MHz, 1.4 * GHz)
    assert np.all(chirp == 1)
    y, start, stop = orc.coherent_dedispersion(x, 0.0, 400 * MHz, 1.4 * GHz)
    assert (start, stop) == (0, 2 ** 20) and y.dtype == np.complex64
    err = np.linalg.norm(y - x) / np.linalg.norm(x)
    assert err < 1e-6


def test_stokes_known_answers():
    """reference tests/test_polarization.py:38-48 (hand-computed Stokes vectors)."""
    x = np.array([[[1 + 1j, 2 + 1j]], [[3 + 0j, 0 + 4j]], [[0 + 2j, 3 + 1j]]],
                 dtype=np.complex128)
    lin = np.array([[[7, -3, 6, -2]], [[25, -7, 0, 24]], [[14, -6, 4, -12]]])
    cir = np.array([[[7, 6, -2, -3]], [[25, 0, 24, -7]], [[14, 4, -12, -6]]])
    assert np.allclose(orc.to_stokes(x, "linear"), lin)
    assert np.allclose(orc.to_stokes(x, "circular"), cir)


def test_intensity_dtype():
    """reference tests/test_radio_signal.py:142-172: c64 -> f32, c128 -> f64."""
    for cd, fd in [(np.complex64, np.float32), (np.complex128, np.float64)]:
        z = (np.arange(6) + 1j * np.arange(6)).astype(cd).reshape(3, 2)
        i = orc.to_intensity(z)
        assert i.dtype == fd
        assert np.allclose(i, 2 * np.arange(6).reshape(3, 2) ** 2)


def test_scrunch():
    a = np.arange(10, dtype=np.float32).reshape(10, 1)
    assert np.array_equal(orc.scrunch(a, 4), np.array([[6.0], [22.0]], dtype=np.float32))
