"""CPU tests of the host layer: containers, units/time stand-ins, fft dispatch, error behaviour.
These mirror the reference's tests/test_signal.py, test_radio_signal.py, test_fft.py,
test_polarization.py and the DispersionMeasure part of test_dedispersion.py for the hot path."""

import pickle
import types

import numpy as np
import pytest
import scipy.fft

import pulsarbat_amd as pb
from pulsarbat_amd import units as u


def rnd(shape, dtype=np.complex64, seed=0):
    r = np.random.default_rng(seed)
    return (r.standard_normal(shape) + 1j * r.standard_normal(shape)).astype(dtype)


class TestUnits:
    def test_quantity_algebra(self):
        f = 1.4 * u.GHz
        assert f.to_value(u.MHz) == pytest.approx(1400.0)
        assert (1 / (50 * u.MHz)).to_value(u.ns) == pytest.approx(20.0)
        assert ((2 * u.MHz) * (3 * u.s)).to_value(u.one) == pytest.approx(6e6)
        assert u.isclose(1 * u.kHz, 1000 * u.Hz)
        assert (f + 100 * u.MHz).to_value(u.GHz) == pytest.approx(1.5)
        assert 2 * u.MHz > 1 * u.MHz and 1 * u.kHz < 1 * u.MHz
        with pytest.raises(u.UnitConversionError):
            (1 * u.MHz).to(u.s)
        with pytest.raises(u.UnitConversionError):
            (1 * u.MHz) + (1 * u.s)
        q = pickle.loads(pickle.dumps(3 * u.MHz))
        assert q.to_value(u.Hz) == 3e6

    def test_time(self):
        t0 = pb.Time(56000.0, format="mjd")
        assert t0.isot.startswith("2012-03-14T00:00:00")
        # the printed string rounds the whole time of day, with carry, not the fraction on its own
        t1 = pb.Time("2020-01-01T12:34:59.000", format="isot", precision=9)
        assert (t1 + 0.9999999996 * u.s).isot == "2020-01-01T12:35:00.000000000"
        assert (t1 + 0.5 * u.s).isot == "2020-01-01T12:34:59.500000000"
        t2 = pb.Time("2020-12-31T23:59:59.000", format="isot", precision=9)
        assert (t2 + 0.99999999999 * u.s).isot == "2021-01-01T00:00:00.000000000"
        t1 = t0 + 1.5 * u.s
        assert (t1 - t0).to_value(u.s) == pytest.approx(1.5)
        assert (t0 + 86400 * u.s).mjd == pytest.approx(56001.0)
        assert t0 < t1 and pb.Time(t1) == t1
        assert pb.Time("2012-03-14T00:00:01.5") == t1
        assert pickle.loads(pickle.dumps(t1)) == t1
        assert (pb.Time.now() - t0).to_value(u.s) > 0


class TestDispersionMeasure:
    def test_basic(self):
        """reference tests/test_dedispersion.py:13-32 verbatim (astropy units -> pulsarbat_amd.units)."""
        DM = pb.DispersionMeasure(2.41e-4)
        for f in [0.1, 1.0, 10.0]:
            dt = DM.time_delay(f * u.MHz, np.inf)
            assert u.isclose(dt, (1 / f / f) * u.s)
            dt = DM.time_delay(np.inf, f * u.MHz)
            assert u.isclose(dt, -(1 / f / f) * u.s)
        dt = DM.time_delay(2 * u.MHz, 1 * u.MHz)
        assert u.isclose(dt, -0.75 * u.s)
        for SR in [1 * u.MHz, 10 * u.MHz, 1 * u.kHz]:
            dn = DM.sample_delay(1 * u.MHz, np.inf, SR)
            assert np.isclose(dn, (SR * u.s).to_value(u.one))
        for a in [10, 20, 100]:
            DM = pb.DispersionMeasure(2.41e-4 * a)
            assert u.isclose(DM.time_delay(1 * u.MHz, np.inf), a * u.s)

    def test_type_and_sign(self):
        DM = pb.DM(56.77)
        assert isinstance(-DM, pb.DispersionMeasure) and (-DM).value == -56.77
        assert pb.DM is pb.DispersionMeasure
        with pytest.raises(u.UnitConversionError):
            pb.DispersionMeasure(1.0, u.MHz)
        f = np.array([1.0, 2.0]) * u.MHz
        d = pb.DM(2.41e-4).time_delay(f, np.inf)
        assert np.allclose(d.to_value(u.s), [1.0, 0.25])


class TestSignals:
    def test_validation(self):
        x = rnd((64, 4))
        with pytest.raises(ValueError):
            pb.Signal(x, sample_rate=1.0)
        with pytest.raises(ValueError):
            pb.Signal(x, sample_rate=-1 * u.Hz)
        with pytest.raises(ValueError):
            pb.Signal(x, sample_rate=1 * u.s)
        with pytest.raises(pb.InvalidSignalError):
            pb.RadioSignal(rnd((64,)), sample_rate=1 * u.Hz, center_freq=1 * u.Hz, chan_bw=1 * u.Hz)
        with pytest.raises(pb.InvalidSignalError):
            pb.DualPolarizationSignal(rnd((64, 4, 3)), sample_rate=1 * u.Hz, center_freq=1 * u.Hz,
                                      pol_type="linear")
        with pytest.raises(pb.InvalidSignalError):
            pb.BasebandSignal(np.zeros((64, 4), dtype=object), sample_rate=1 * u.Hz, center_freq=1 * u.Hz)
        with pytest.raises(pb.InvalidSignalError):
            pb.Signal(np.zeros((4, 0)), sample_rate=1 * u.Hz)
        with pytest.raises(ValueError):
            pb.BasebandSignal(x, sample_rate=1 * u.Hz, center_freq=1 * u.Hz, freq_align="middle")
        with pytest.raises(ValueError):
            pb.DualPolarizationSignal(rnd((8, 2, 2)), sample_rate=1 * u.Hz, center_freq=1 * u.Hz,
                                      pol_type="elliptical")
        with pytest.raises(ValueError):
            pb.Signal(x, sample_rate=1 * u.Hz, start_time="not a time")
        with pytest.raises(ValueError):
            pb.Signal(x, sample_rate=1 * u.Hz, meta=3)

    def test_dtype_rules(self):
        """core.py:78-92: real input is safe-cast to the first required dtype."""
        z = pb.BasebandSignal(np.ones((8, 2), dtype=np.float32), sample_rate=1 * u.Hz, center_freq=1 * u.Hz)
        assert z.dtype == np.complex128
        z = pb.BasebandSignal(rnd((8, 2), np.complex64), sample_rate=1 * u.Hz, center_freq=1 * u.Hz)
        assert z.dtype == np.complex64
        i = pb.IntensitySignal(np.ones((8, 2), dtype=np.int32), sample_rate=1 * u.Hz, center_freq=1 * u.Hz,
                               chan_bw=1 * u.Hz)
        assert i.dtype == np.float64

    def test_frequency_bookkeeping(self):
        """core.py:546-574 and config-2 geometry of SURVEY.md 8(a)."""
        z = pb.BasebandSignal(rnd((16, 8, 2)), sample_rate=50 * u.MHz, center_freq=1.4 * u.GHz)
        assert np.allclose(z.channel_freqs.to_value(u.MHz), np.arange(1225, 1600, 50))
        assert z.max_freq.to_value(u.GHz) == pytest.approx(1.6)
        assert z.min_freq.to_value(u.GHz) == pytest.approx(1.2)
        assert z.chan_bw.to_value(u.MHz) == 50 and z.bandwidth.to_value(u.MHz) == 400
        for align, off in (("bottom", 0.0), ("center", 0.5), ("top", 1.0)):
            z = pb.BasebandSignal(rnd((16, 4)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz, freq_align=align)
            assert np.allclose(z.channel_freqs.to_value(u.MHz), 1000 + np.arange(4) + off - 2)
        z = pb.BasebandSignal(rnd((16, 3)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz, freq_align="top")
        assert z.freq_align == "center"

    def test_slicing_and_like(self):
        t0 = pb.Time(56000.0, format="mjd")
        z = pb.DualPolarizationSignal(rnd((128, 8, 2)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz,
                                      pol_type="circular", start_time=t0, meta={"a": 1})
        y = z[10:100]
        assert type(y) is type(z) and len(y) == 90 and y.pol_type == "circular" and y.meta == {"a": 1}
        assert (y.start_time - t0).to_value(u.us) == pytest.approx(10.0)
        y = z[::2]
        assert y.sample_rate.to_value(u.kHz) == pytest.approx(500.0)
        y = z[:, 2:6]
        assert y.nchan == 4 and np.allclose(y.channel_freqs.to_value(u.Hz), z.channel_freqs.to_value(u.Hz)[2:6])
        with pytest.raises(IndexError):
            z[0]
        with pytest.raises(IndexError):
            z[:, 0]
        i = pb.IntensitySignal.like(z, np.ones((128, 8, 2), np.float32))
        assert isinstance(i, pb.IntensitySignal) and i.chan_bw.to_value(u.MHz) == 1
        with pytest.raises(ValueError):
            pb.DualPolarizationSignal.like(pb.Signal(rnd((8, 2, 2)), sample_rate=1 * u.Hz))
        assert z.time_length.to_value(u.us) == pytest.approx(128.0)
        assert (z.stop_time - t0).to_value(u.us) == pytest.approx(128.0)
        assert (t0 + 5 * u.us) in z and (t0 + 500 * u.us) not in z
        zz = pickle.loads(pickle.dumps(z))
        assert np.array_equal(np.asarray(zz), np.asarray(z)) and zz.start_time == t0

    def test_ufunc_passthrough(self):
        z = pb.BasebandSignal(rnd((16, 2)), sample_rate=1 * u.Hz, center_freq=1 * u.Hz)
        y = np.conj(z * 2)
        assert isinstance(y, pb.BasebandSignal) and np.allclose(np.asarray(y), np.conj(np.asarray(z) * 2))
        assert "BasebandSignal" in repr(z) and "Sample rate" in str(z)

    def test_intensity_dtype(self):
        """reference tests/test_radio_signal.py:142-172."""
        for cd, fd in ((np.complex64, np.float32), (np.complex128, np.float64)):
            z = pb.BasebandSignal(rnd((32, 4), cd), sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
            i = z.to_intensity()
            assert isinstance(i, pb.IntensitySignal) and i.dtype == fd
            assert np.allclose(np.asarray(i), np.abs(np.asarray(z)) ** 2)

    @pytest.mark.parametrize("pol_type", ["linear", "circular"])
    def test_pol_reversibility(self, pol_type):
        """reference tests/test_polarization.py:10-31."""
        sig = np.exp(1j * np.random.default_rng(1).uniform(-np.pi, np.pi, (256, 16, 2)))
        z = pb.DualPolarizationSignal(sig, pol_type=pol_type, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        if pol_type == "linear":
            x, y = z.to_linear(), z.to_circular().to_linear()
        else:
            x, y = z.to_circular(), z.to_linear().to_circular()
        for a in (x, y):
            assert a.pol_type == pol_type and np.allclose(np.array(z), np.array(a))

    def test_stokes(self):
        """reference tests/test_polarization.py:34-60 (hand-computed vectors)."""
        x = np.array([[[1 + 1j, 2 + 1j]], [[3 + 0j, 0 + 4j]], [[0 + 2j, 3 + 1j]]], dtype=np.complex128)
        lin = np.array([[[7, -3, 6, -2]], [[25, -7, 0, 24]], [[14, -6, 4, -12]]])
        cir = np.array([[[7, 6, -2, -3]], [[25, 0, 24, -7]], [[14, 4, -12, -6]]])
        for pol_type, stokes in zip(["linear", "circular"], [lin, cir]):
            z = pb.DualPolarizationSignal(x, pol_type=pol_type, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
            for y in (z.to_linear().to_stokes(), z.to_circular().to_stokes()):
                assert isinstance(y, pb.FullStokesSignal)
                assert np.allclose(np.array(y), stokes)
            s = z.to_stokes()
            assert np.allclose(np.asarray(s["I"]), stokes[..., 0]) and np.allclose(np.asarray(s.stokesV), stokes[..., 3])
            with pytest.raises(KeyError):
                s["X"]


class TestFFT:
    FFT_FUNCS = ["fft", "fft2", "fftn", "ifft", "ifft2", "ifftn", "hfft", "ihfft", "rfft", "rfft2",
                 "rfftn", "irfft", "irfft2", "irfftn"]

    def test_dir_funcs(self):
        """reference tests/test_fft.py:28-39."""
        for a, b in zip(dir(pb.fft), sorted(self.FFT_FUNCS)):
            assert a == b
            assert isinstance(getattr(pb.fft, a), types.FunctionType)
        with pytest.raises(AttributeError):
            _ = pb.fft.fish

    @pytest.mark.parametrize("fft_func, N", [("fft", 8), ("ifft", 8), ("irfft", 9), ("hfft", 9)])
    def test_complex_input_fft(self, fft_func, N):
        """reference tests/test_fft.py:41-54 (numpy half)."""
        for d in [np.float32, np.float64]:
            a = np.arange(N, dtype=d) + 1j * np.arange(N, dtype=d)
            x = getattr(scipy.fft, fft_func)(a)
            y = getattr(pb.fft, fft_func)(a)
            assert type(a) is type(y) and x.dtype == y.dtype and np.allclose(x, y)


class TestHotPathWithoutGpu:
    def test_fails_loudly(self):
        """No GPU here: the product path must raise, never fall back to a CPU implementation."""
        from pulsarbat_amd import _hip
        if _hip.available():
            pytest.skip("a HIP device is present")
        z = pb.BasebandSignal(rnd((1024, 2)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
        with pytest.raises(_hip.HipUnavailableError):
            pb.coherent_dedispersion(z, pb.DM(1.0))
        with pytest.raises(_hip.HipUnavailableError):
            pb.DM(1.0).chirp_from_signal(z)
        with pytest.raises(_hip.HipUnavailableError):
            pb.DM(1.0).chirp_function(1024, 1 * u.us, 1 * u.GHz, 1 * u.GHz)

    def test_type_errors_before_gpu(self):
        x = rnd((1024, 2))
        with pytest.raises(TypeError):
            pb.coherent_dedispersion(pb.Signal(x, sample_rate=1 * u.MHz), pb.DM(1.0))
        with pytest.raises(TypeError):
            pb.DM(1.0).chirp_from_signal(pb.Signal(x, sample_rate=1 * u.MHz))

    def test_detected_stream_arguments_before_gpu(self):
        """A filterbank stream needs a scrunch factor a fused detect tail exists for: said before anything touches the GPU."""
        z = pb.DualPolarizationSignal(rnd((4096, 2, 2)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear")
        for bad in (3, 48, 100):
            with pytest.raises(ValueError, match="nscrunch"):
                pb.coherent_dedispersion_stream(z, pb.DM(1.0), chunk=1024, detect="I", nscrunch=bad)

    def test_product_does_not_import_oracle(self):
        import os, re
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        # the package and the measurement scripts; (bench.py imports it inside its CPU-baseline leg only, below)
        for sub in ("pulsarbat_amd", "tools"):
            for dirpath, _, files in os.walk(os.path.join(root, sub)):
                for f in files:
                    if f.endswith(".py"):
                        src = open(os.path.join(dirpath, f)).read()
                        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
        bench = open(os.path.join(root, "bench.py")).read()
        uses = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle", bench, re.M)]
        enclosing = bench[:uses[0]].rsplit("\ndef ", 1)[-1]     # text from the last "def" before the import
        assert len(uses) == 1 and enclosing.startswith("cpu_baseline(")


class TestRound2HostLogic:
    """Host-side pieces of the round-2 entry points that need no GPU."""

    def test_user_chirp_broadcast_rules(self):
        """dedispersion.py:124-125: length-1 axes are appended to the chirp, then numpy broadcasts it against z.data."""
        from pulsarbat_amd.transforms.dedispersion import _broadcast_chirp
        x = np.zeros((64, 3, 2), np.complex64)
        z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear")
        rng = np.random.default_rng(0)
        for shape, per_pol in [((64, 3), False), ((64, 3, 1), False), ((64,), False), ((64, 1), False), ((1, 3), False),
                               ((), False), ((64, 3, 2), True), ((1, 1, 2), True), ((64, 1, 2), True)]:
            c = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)
            rows, pp, dt = _broadcast_chirp(c, z)
            assert pp is per_pol and dt == np.complex64 and rows.flags.c_contiguous
            padded = c.reshape(shape + (1,) * (3 - len(shape)))
            want = np.broadcast_to(padded, (64, 3, 2) if per_pol else (64, 3, 1)).reshape(64, -1)
            assert rows.shape == (64, 6 if per_pol else 3) and np.array_equal(rows, want)
        # numpy's promotion: complex64 data x complex128 / float64 chirp -> complex128; float32 chirp stays complex64
        assert _broadcast_chirp(np.ones((64, 3), np.complex128), z)[2] == np.complex128
        assert _broadcast_chirp(np.ones((64, 3), np.float64), z)[2] == np.complex128
        assert _broadcast_chirp(np.ones((64, 3), np.float32), z)[2] == np.complex64
        for bad in [(64, 2), (63, 3), (64, 3, 3), (64, 3, 2, 1)]:
            with pytest.raises(ValueError):
                _broadcast_chirp(np.ones(bad, np.complex64), z)

    def test_sharded_entry_point_needs_the_device(self):
        """No CPU fallback in the sharded call either: without a GPU the real plan raises HipUnavailableError."""
        from pulsarbat_amd import shard, _hip
        if _hip.available():
            pytest.skip("a GPU is present")
        x = np.zeros((4096, 4, 2), np.complex64)
        z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear")
        with pytest.raises(_hip.HipUnavailableError):
            shard.coherent_dedispersion_sharded(shard.shard_signal(z, 2, 0), pb.DM(1.0), band_min=z.min_freq,
                                                band_max=z.max_freq, ref_freq=z.center_freq, device=0)
        with pytest.raises(_hip.HipUnavailableError):
            pb.contrib.stft_dedisperse(z, pb.DM(1.0), nperseg=64)

    def test_fft_names_dispatch_numpy_to_scipy(self):
        """numpy arrays still go to scipy for every name (reference fft.py:36-38), device registration or not."""
        import scipy.fft
        x = np.random.default_rng(1).standard_normal((8, 16)).astype(np.float32)
        for name in ("fft", "ifft", "fft2", "rfft", "irfft", "hfft", "ihfft", "fftn", "rfftn"):
            got, want = getattr(pb.fft, name)(x), getattr(scipy.fft, name)(x)
            assert got.dtype == want.dtype and np.allclose(got, want)
