"""incoherent dedispersion and polarisation-basis changes on device data."""

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc


def test_oracle_incoherent_length():
    """reference tests/test_dedispersion.py:167-189 (length check) on the oracle restatement."""
    shape, sr, ref, bw = (8192, 32, 4), 1e3, 1e9, 8e6
    x = np.random.default_rng(0).standard_normal(shape)
    for dm in (50.0, 100.0, 200.0):
        y, _ = orc.incoherent_dedispersion(x, dm, sr, ref, bw)
        f = orc.channel_freqs(ref, bw, 32)
        top = round(float(orc.sample_delay(dm, ref, f[-1], sr)))
        bot = round(float(orc.sample_delay(dm, f[0], ref, sr)))
        assert len(x) - len(y) == int(top + bot)


@pytest.mark.parametrize("dm", [50.0, 100.0, 200.0])
@pytest.mark.parametrize("device", [False, pytest.param(True, marks=pytest.mark.gpu)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_incoherent_dedispersion(dm, device, dtype):
    """reference tests/test_dedispersion.py:167-189 plus element-wise parity with the oracle."""
    SR, ref_freq, shape = 1 * u.kHz, 1 * u.GHz, (8192, 32, 4)
    t0 = pb.Time(56000.0, format="mjd")
    DM = pb.DispersionMeasure(dm)
    x = np.random.default_rng(1).standard_normal(shape).astype(dtype)
    z1 = pb.FullStokesSignal(x, sample_rate=SR, start_time=t0, center_freq=ref_freq, chan_bw=8 * u.MHz)
    with pytest.raises(TypeError):
        pb.incoherent_dedispersion(pb.Signal(x, sample_rate=SR), DM)
    z2 = pb.incoherent_dedispersion(z1.to_device() if device else z1, DM)
    delay_top = DM.sample_delay(ref_freq, z1.channel_freqs[-1], SR).round()
    delay_bot = DM.sample_delay(z1.channel_freqs[0], ref_freq, SR).round()
    assert len(z1) - len(z2) == int(delay_top + delay_bot)
    want, crop_before = orc.incoherent_dedispersion(x, dm, 1e3, 1e9, 8e6)
    assert z2.dtype == dtype and type(z2) is type(z1)
    assert np.array_equal(np.asarray(z2), want)                     # a gather: bit exact
    assert abs((z2.start_time - t0).to_value(u.s) - crop_before / 1e3) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_incoherent_complex_device(dtype):
    x = (np.random.default_rng(2).standard_normal((4096, 16, 2)) * (1 + 1j)).astype(dtype)
    z = pb.DualPolarizationSignal(x, sample_rate=1 * u.kHz, center_freq=1 * u.GHz, pol_type="linear").to_device()
    y = pb.incoherent_dedispersion(z, pb.DM(30.0))
    want, _ = orc.incoherent_dedispersion(x, 30.0, 1e3, 1e9, 1e3)
    assert isinstance(y.data, pb.DeviceArray) and np.array_equal(np.asarray(y), want)


@pytest.mark.gpu
@pytest.mark.parametrize("pol_type", ["linear", "circular"])
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_pol_reversibility_device(pol_type, dtype):
    """reference tests/test_polarization.py:10-31 on device-resident data."""
    sig = np.exp(1j * np.random.default_rng(3).uniform(-np.pi, np.pi, (4096, 16, 2))).astype(dtype)
    z = pb.DualPolarizationSignal(sig, pol_type=pol_type, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    zd = z.to_device()
    if pol_type == "linear":
        x, y, mid, host_mid = zd.to_linear(), zd.to_circular().to_linear(), zd.to_circular(), z.to_circular()
    else:
        x, y, mid, host_mid = zd.to_circular(), zd.to_linear().to_circular(), zd.to_linear(), z.to_linear()
    tol = 1e-6 if dtype == np.complex64 else 1e-13
    assert np.allclose(np.asarray(mid), np.asarray(host_mid), atol=tol)
    for a in (x, y):
        assert isinstance(a.data, pb.DeviceArray) and a.pol_type == pol_type
        assert np.allclose(np.array(z), np.array(a), atol=tol)
    # Stokes from either basis agree (reference tests/test_polarization.py:50-60)
    s1, s2 = zd.to_linear().to_stokes(), zd.to_circular().to_stokes()
    assert np.allclose(np.asarray(s1), np.asarray(s2), atol=10 * tol)


@pytest.mark.gpu
def test_incoherent_series_major_arrays():
    """Series-major device arrays: one shifted contiguous copy per channel; same values, layout kept."""
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((1 << 16, 6, 2)) + 1j * rng.standard_normal((1 << 16, 6, 2))).astype(np.complex64)
    z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=400 * u.MHz, pol_type="linear",
                                  start_time=pb.Time(56000.0, format="mjd"))
    a = pb.incoherent_dedispersion(z.to_device(), pb.DM(3.0))
    zs = type(z).like(z, z.to_device().data.to_series_major())
    b = pb.incoherent_dedispersion(zs, pb.DM(3.0))
    assert b.data.series_major_pitch() is not None and a.shape == b.shape
    assert np.array_equal(np.asarray(a), np.asarray(b)) and a.start_time.isclose(b.start_time)


@pytest.mark.gpu
def test_incoherent_series_major_more_series_than_a_grid_dimension():
    """70 000 series (what a 4096-point channelisation of a few streams, held series-major, looks like): the shifted-copy
    kernel is launched in slabs of at most 65 535 series (grid.y's limit) -- bit exact, the last series included."""
    nchan, n = 35000, 96
    rng = np.random.default_rng(12)
    x = (rng.standard_normal((n, nchan, 2)) + 1j * rng.standard_normal((n, nchan, 2))).astype(np.complex64)
    z = pb.DualPolarizationSignal(x, sample_rate=1 * u.kHz, center_freq=400 * u.MHz, pol_type="linear")
    zs = type(z).like(z, z.to_device().data.to_series_major())
    y = pb.incoherent_dedispersion(zs, pb.DM(5.0))
    want, _ = orc.incoherent_dedispersion(x, 5.0, 1e3, 400e6, 1e3)
    assert y.shape == want.shape and 0 < want.shape[0] < n and np.array_equal(np.asarray(y), want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((200003, 8, 2), np.complex64), ((150001, 8, 4), np.float32),
                                         ((100000, 4, 1), np.complex64), ((131072, 6, 2), np.complex64)])
def test_incoherent_large_blocks_two_pass(shape, dtype):
    """Long sample-major blocks with a power-of-two number of 8-byte series go through the two-pass form of the gather
    (series-major scratch copy + re-interleave with per-series offsets); odd lengths, odd delays; still bit exact.
    The (6, 2) case has 12 series and takes the direct gather."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal(shape).astype(np.float32)
    if dtype == np.complex64:
        x = (x + 1j * rng.standard_normal(shape).astype(np.float32)).astype(np.complex64)
    kw = dict(sample_rate=1 * u.MHz, center_freq=400 * u.MHz, start_time=pb.Time(56000.0, format="mjd"))
    if dtype == np.float32:
        z = pb.FullStokesSignal(x, chan_bw=1 * u.MHz, **kw)
    elif shape[2] == 2:
        z = pb.DualPolarizationSignal(x, pol_type="linear", **kw)
    else:
        z = pb.BasebandSignal(x, **kw)
    y = pb.incoherent_dedispersion(z.to_device(), pb.DM(5.0))
    want, crop_before = orc.incoherent_dedispersion(x, 5.0, 1e6, 400e6, 1e6)
    assert isinstance(y.data, pb.DeviceArray) and y.shape == want.shape and want.shape[0] > 65536
    assert np.array_equal(np.asarray(y), want)
    assert abs((y.start_time - z.start_time).to_value(u.s) - crop_before / 1e6) < 1e-12
