"""GPU parity tests: the HIP path (through the C ABI, via the reference-shaped Python API)
against the CPU oracle on the same seeded inputs.  Tolerance: BASELINE.json north_star asks for
<= 1e-5 relative for complex64; the metric is per-series relative L2 error plus max-abs error
over rms.  Test bodies mirror the reference's tests/test_dedispersion.py."""

import os

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from oracle import dedisp_oracle as orc
from tests._detect_check import assert_detect_close

pytestmark = pytest.mark.gpu

RTOL_L2 = 1e-5      # relative L2 per series (north_star)
RTOL_MAX = 4e-5     # max |err| / rms(ref) per series


def series_errors(got, ref):
    got = np.asarray(got).reshape(ref.shape[0], -1)
    ref = ref.reshape(ref.shape[0], -1)
    num = np.linalg.norm(got - ref, axis=0)
    den = np.linalg.norm(ref, axis=0)
    rms = den / np.sqrt(ref.shape[0])
    return (num / den).max(), (np.abs(got - ref).max(axis=0) / rms).max()


def same_kernels_or_close(a, b, rel=5e-7):
    """Two routes through the library that run the same butterflies.  Where they also run the same KERNELS the results are
    bit-identical; since round 4 the sample-major route of quad-series blocks is the four-pass schedule
    (csrc/fd4_kernels.hpp), whose row kernel hipcc contracts into multiply-adds differently: equal to the last bits then,
    not always bit for bit (tests/test_gpu_fd4.py holds both schedules against each other and against the oracle)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if np.array_equal(a, b):
        return True
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    num = np.linalg.norm((a2 - b2).astype(np.complex128 if np.iscomplexobj(a2) else np.float64), axis=0)
    den = np.linalg.norm(b2.astype(np.complex128 if np.iscomplexobj(b2) else np.float64), axis=0)
    return bool((num <= rel * den).all())


def make_signal(x, sr, fc, **kw):
    if x.ndim == 3 and x.shape[2] == 2:
        return pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz,
                                         pol_type="linear", **kw)
    return pb.BasebandSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, **kw)


def check(shape, dm, sr, fc, ref=None, seed=1, variant="auto", device=False, freq_align="center"):
    x = orc.synthetic_block(shape, seed)
    z = make_signal(x, sr, fc, freq_align=freq_align, start_time=pb.Time(56000.0, format="mjd"))
    zz = z.to_device() if device else z
    y = pb.coherent_dedispersion(zz, pb.DM(dm), ref_freq=None if ref is None else ref * u.Hz,
                                 variant=variant)
    yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc, freq_align=freq_align, ref_freq_hz=ref)
    assert type(y) is type(z)
    assert type(y.data) is type(zz.data)
    assert y.dtype == np.complex64
    assert y.shape == yr.shape
    dt = (y.start_time - z.start_time).to_value(u.s)
    assert abs(dt - start / sr) < 1e-12
    if yr.size:
        l2, mx = series_errors(y, yr)
        assert l2 < RTOL_L2, f"relative L2 {l2:.2e}"
        assert mx < RTOL_MAX, f"max abs / rms {mx:.2e}"
    return y


class TestCoherentDedispersion:
    @pytest.mark.parametrize("dm", [10.0, 50.0, 100.0])
    def test_basic(self, dm):
        """reference tests/test_dedispersion.py:36-71: type errors, lengths, start offsets,
        ref_freq in {min, center, max}; plus oracle parity."""
        shape = (8192, 4)
        fcen, sr = 1e9, 1e6
        x = orc.synthetic_block(shape, 3)
        with pytest.raises(TypeError):
            pb.coherent_dedispersion(pb.Signal(x, sample_rate=sr * u.Hz), pb.DM(dm))
        with pytest.raises(TypeError):
            pb.DM(dm).chirp_from_signal(pb.Signal(x, sample_rate=sr * u.Hz))
        z = make_signal(x, sr, fcen, start_time=pb.Time.now())
        DM = pb.DM(dm)
        for ref in [z.min_freq, z.center_freq, z.max_freq]:
            y = pb.coherent_dedispersion(z, DM, ref_freq=ref)
            assert len(z) - len(y) >= DM.sample_delay(z.min_freq, z.max_freq, z.sample_rate)
            d_st = y.start_time - z.start_time
            toffset = int(np.rint((d_st * z.sample_rate).to_value(u.one)))
            assert toffset >= DM.sample_delay(ref, z.max_freq, z.sample_rate)
            yr, _, _ = orc.coherent_dedispersion(x, dm, sr, fcen, ref_freq_hz=ref.to_value(u.Hz))
            assert series_errors(y, yr)[0] < RTOL_L2

    @pytest.mark.parametrize("seed", [4, 8, 15, 16, 23, 42])
    def test_reversibility(self, seed):
        """reference tests/test_dedispersion.py:73-98 with complex64 data: +DM then -DM returns
        the input.  The reference's atol 3e-8 needs complex128 data; for complex64 the oracle
        itself reaches ~1.5e-6, so the bound here is 1e-5 of the peak."""
        import scipy.signal
        ref, sr, dm = 600e6, 400e6, 0.01
        N, M = 2 ** 18, 2 ** 12
        R = np.random.default_rng(seed=seed)
        x = R.standard_normal(N) + 1j * R.standard_normal(N)
        x *= np.exp(-(((np.arange(N) - N // 2) / M) ** 2))
        sos = scipy.signal.butter(10, 0.45, "lowpass", fs=1.0, output="sos")
        x = scipy.signal.sosfilt(sos, x).astype(np.complex64).reshape(-1, 1)
        sig = make_signal(x, sr, ref, start_time=pb.Time(56000.0, format="mjd"))
        temp = pb.coherent_dedispersion(sig, pb.DM(dm))
        # the second call needs a power-of-two length (the first crop removed samples): take the
        # 2^17 window centred on the pulse (Gaussian envelope, sigma 2^12 samples)
        off1 = int(np.rint(((temp.start_time - sig.start_time) * sig.sample_rate).to_value(u.one)))
        a = N // 2 - off1 - 2 ** 16
        sig2 = pb.coherent_dedispersion(temp[a:a + 2 ** 17], -pb.DM(dm))
        toffset = sig2.start_time - sig.start_time
        noffset = int(np.rint((toffset * sig.sample_rate).to_value(u.one)))
        sig1 = sig[noffset:noffset + len(sig2)]
        res = np.array(sig1) - np.array(sig2)
        assert np.abs(res).max() < 1e-5 * np.abs(x).max()

    @pytest.mark.parametrize("dm", [0.01, 0.02])
    def test_correctness(self, dm):
        """reference tests/test_dedispersion.py:100-139: pre-dispersed Gabor wavelets re-align."""
        ref, sr = 600e6, 400e6
        index, N, width = 100000, 2 ** 18, 256
        t = np.arange(N) / sr
        x = np.zeros(N, dtype=np.complex128)
        for df in np.linspace(-3 * sr / 8, 3 * sr / 8, 13):
            tt = t - (t[index] + orc.time_delay(dm, ref + df, ref))
            x += np.exp(2j * np.pi * tt * df - (tt / (width / sr)) ** 2)
        x = x.astype(np.complex64).reshape(-1, 1)
        sig1 = make_signal(x, sr, ref, start_time=pb.Time.now())
        sig2 = pb.coherent_dedispersion(sig1, pb.DM(dm))
        toffset = sig2.start_time - sig1.start_time
        noffset = int(np.rint((toffset * sig2.sample_rate).to_value(u.one)))
        a1, a2 = np.array(sig1), np.array(sig2)
        id1, id2 = index - noffset - 8 * width, index - noffset + 8 * width
        p1 = (np.abs(a1) ** 2).sum()
        p2 = (np.abs(a2[id1:id2]) ** 2).sum()
        assert np.allclose(p1, p2, rtol=1e-5)
        assert np.allclose(a2[id2:], 0, atol=1e-5)
        assert np.allclose(a2[:id1], 0, atol=1e-5)

    @pytest.mark.parametrize("dm", [10, 20, 50])
    def test_chirp(self, dm):
        """reference tests/test_dedispersion.py:141-164: precomputed chirp (3-D and 2-D)."""
        shape = (8192, 4, 2)
        x = orc.synthetic_block(shape, dm)
        z = make_signal(x, 1e6, 1e9)
        DM = pb.DM(dm)
        chirp = DM.chirp_from_signal(z)
        assert chirp.shape == (8192, 4, 1) and chirp.dtype == np.complex64
        cr = orc.chirp_from_signal(dm, shape, 1e6, 1e9)
        assert np.abs(chirp - cr).max() < 2.5e-7
        y1 = pb.coherent_dedispersion(z, DM)
        y2 = pb.coherent_dedispersion(z, DM, chirp=chirp)
        y3 = pb.coherent_dedispersion(z, DM, chirp=chirp.squeeze())
        assert np.allclose(y1, y2) and np.allclose(y1, y3)
        for rf in [z.min_freq, z.max_freq]:
            chirp = DM.chirp_from_signal(z, ref_freq=rf)
            y1 = pb.coherent_dedispersion(z, DM, ref_freq=rf)
            y2 = pb.coherent_dedispersion(z, DM, ref_freq=rf, chirp=chirp)
            assert np.allclose(y1, y2)
        # an oracle-made chirp drives the HIP path to the oracle's answer
        yo, _, _ = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
        y4 = pb.coherent_dedispersion(z, DM, chirp=cr)
        assert series_errors(y4, yo)[0] < RTOL_L2


@pytest.mark.parametrize("log2n", [5, 6, 9, 12, 14])
def test_single_tile_lengths(log2n):
    check((1 << log2n, 4, 2), 0.02 if log2n < 12 else 5.0, 1e6, 1e9, seed=log2n)


@pytest.mark.parametrize("log2n", [15, 16, 17, 19, 20, 21])
@pytest.mark.parametrize("variant", ["direct3", "planar5"])
def test_multi_pass_lengths(log2n, variant):
    check((1 << log2n, 2, 2), 20.0, 1e6, 1e9, seed=log2n, variant=variant)


@pytest.mark.parametrize("shape", [(1 << 16, 1), (1 << 16, 3, 2), (1 << 16, 5), (1 << 15, 7, 2),
                                   (1 << 16, 2, 2, 3), (4096, 33), (1 << 17, 16, 2)])
def test_ragged_series_counts(shape):
    """nchan / npol combinations that do not fill a 16-column tile (and extra trailing axes)."""
    check(shape, 5.0, 1e6, 1e9, seed=sum(shape))


@pytest.mark.parametrize("freq_align", ["bottom", "center", "top"])
def test_freq_align(freq_align):
    check((1 << 15, 4, 2), 30.0, 1e6, 1e9, freq_align=freq_align)


def test_config1_identity():
    """BASELINE.json configs[0]: 2^20 x 1 x 1, DM = 0: chirp == 1, no crop, output == input."""
    x = orc.synthetic_block((1 << 20, 1, 1), 20260001)
    z = make_signal(x, 400e6, 1.4e9)
    y = pb.coherent_dedispersion(z, pb.DM(0.0))
    assert y.shape == x.shape
    assert series_errors(y, x)[0] < 2e-6
    c = pb.DM(0.0).chirp_from_signal(z)
    assert np.all(c == 1)


def test_device_resident_roundtrip():
    """persist() semantics: DeviceArray in -> DeviceArray out, then detection on device."""
    y = check((1 << 16, 4, 2), 10.0, 1e6, 1e9, device=True)
    assert isinstance(y.data, pb.DeviceArray)
    yh = np.asarray(y)
    i = y.to_intensity()
    assert isinstance(i, pb.IntensitySignal) and i.dtype == np.float32
    assert np.allclose(np.asarray(i.data), orc.to_intensity(yh), rtol=1e-5, atol=1e-7)
    for pol in ("linear", "circular"):
        zz = pb.DualPolarizationSignal(y.data, sample_rate=y.sample_rate, center_freq=y.center_freq,
                                       pol_type=pol)
        s = zz.to_stokes()
        assert isinstance(s, pb.FullStokesSignal) and s.shape == yh.shape[:2] + (4,)
        assert np.allclose(np.asarray(s.data), orc.to_stokes(yh, pol), rtol=1e-4, atol=1e-5)


def test_stokes_known_answers_device():
    """reference tests/test_polarization.py:38-48 on device-resident data."""
    x = np.array([[[1 + 1j, 2 + 1j]], [[3 + 0j, 0 + 4j]], [[0 + 2j, 3 + 1j]]], dtype=np.complex64)
    lin = np.array([[[7, -3, 6, -2]], [[25, -7, 0, 24]], [[14, -6, 4, -12]]])
    cir = np.array([[[7, 6, -2, -3]], [[25, 0, 24, -7]], [[14, 4, -12, -6]]])
    for pol, want in (("linear", lin), ("circular", cir)):
        z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type=pol)
        got = np.asarray(z.to_device().to_stokes().data)
        assert np.allclose(got, want)


def test_detect_scrunch():
    """dedisperse + Stokes-I + time scrunch (BASELINE.json configs[4] shape of work, small)."""
    shape, dm, sr, fc, k = (1 << 17, 4, 2), 20.0, 1e6, 1e9, 64
    x = orc.synthetic_block(shape, 11)
    z = make_signal(x, sr, fc)
    got, start = pb.dedisperse_detect(z, pb.DM(dm), mode="I", nscrunch=k)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, sr, fc)
    want = orc.scrunch(orc.to_stokes(yr, "linear")[:, :, 0], k)
    assert start == s0 and got.shape == want.shape and got.dtype == np.float32
    assert_detect_close(got, yr, "I", k)


def test_fft_dispatch_device():
    """reference tests/test_fft.py:41-61: same container type, same dtype, allclose to scipy."""
    import scipy.fft
    x = orc.synthetic_block((4096, 6), 5)
    d = pb.DeviceArray.from_host(x)
    for name in ("fft", "ifft"):
        got = getattr(pb.fft, name)(d, axis=0)
        want = getattr(scipy.fft, name)(x, axis=0)
        assert type(got) is type(d) and got.dtype == want.dtype
        assert np.linalg.norm(np.asarray(got) - want) / np.linalg.norm(want) < 1e-6
    with pytest.raises(TypeError):
        pb.fft.rfft(d)   # scipy: "x must be a real sequence"


def test_fft_all_names_axes_n_norm_on_device():
    """The reference's pb.fft exposes fourteen scipy.fft names and dispatches each on the array type (fft.py:8-48; its
    channeliser calls pb.fft.fft(x, axis=2, n=nfft), contrib/misc.py:47): every name on a DeviceArray must return a
    DeviceArray equal to scipy's result for scipy's n / s / axis / axes / norm arguments."""
    import scipy.fft
    rng = np.random.default_rng(3)
    xc = (rng.standard_normal((6, 40, 48)) + 1j * rng.standard_normal((6, 40, 48))).astype(np.complex64)
    xr = rng.standard_normal((6, 40, 48)).astype(np.float32)
    dc, dr = pb.DeviceArray.from_host(xc), pb.DeviceArray.from_host(xr)

    def same(name, dev, host, tol=2e-6, **kw):
        got = getattr(pb.fft, name)(dev, **kw)
        want = getattr(scipy.fft, name)(host, **kw)
        assert isinstance(got, pb.DeviceArray), name
        g = np.asarray(got)
        assert g.shape == want.shape and g.dtype == want.dtype, (name, kw, g.shape, want.shape, g.dtype, want.dtype)
        assert np.linalg.norm(g - want) / np.linalg.norm(want) < tol, (name, kw)

    for name in ("fft", "ifft"):
        same(name, dc, xc)                                  # last axis by default
        same(name, dc, xc, axis=2, n=64)                    # the channeliser's call: zero padding
        same(name, dc, xc, axis=1, n=33)                    # truncation to an odd length
        same(name, dc, xc, axis=0, norm="ortho")
        same(name, dc, xc, axis=-2, norm="forward")
    for name in ("fft2", "ifft2", "fftn", "ifftn"):
        same(name, dc, xc)
        same(name, dc, xc, s=(32, 50), axes=(1, 2), norm="ortho")
    same("rfft", dr, xr)
    same("rfft", dr, xr, n=50, axis=1)
    same("rfft2", dr, xr)
    same("rfftn", dr, xr, s=(8, 40, 30))
    same("ihfft", dr, xr, axis=1)
    for n in (None, 94, 95, 60):
        same("irfft", dc, xc, n=n)
        same("hfft", dc, xc, n=n, tol=4e-6)
    same("irfft2", dc, xc)
    same("irfftn", dc, xc, s=(6, 40, 77))
    with pytest.raises(ValueError):
        pb.fft.fft(dc, norm="bogus")


@pytest.mark.parametrize("shape,dm", [((1000, 2), 1.0), ((16, 3), 0.001), ((6, 1), 0.0), ((8190, 4, 2), 10.0),
                                      ((100000, 2, 2), 20.0), ((98304, 2), 20.0), ((65537, 1), 5.0),
                                      ((250047, 3, 2), 30.0), (((1 << 20) + 1, 1, 2), 40.0)])
def test_arbitrary_lengths(shape, dm):
    """nsample that is not a power of two (the reference takes any length through pocketfft):
    Bluestein over the power-of-two pipeline."""
    check(tuple(shape), dm, 1e6, 1e9, seed=7)


def test_arbitrary_length_stream_and_detect():
    shape, dm = (50000, 2, 2), 10.0
    x = orc.synthetic_block(shape, 3)
    z = make_signal(x, 1e6, 1e9)
    got, start = pb.dedisperse_detect(z, pb.DM(dm), mode="linear", nscrunch=10)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    want = orc.scrunch(orc.to_stokes(yr, "linear"), 10)
    assert start == s0 and got.shape == want.shape
    assert_detect_close(got, yr, "linear", 10)


def test_errors():
    zz = make_signal(orc.synthetic_block((1024, 2), 1), 1e6, 1e9)
    with pytest.raises(ValueError):   # does not broadcast against z.data
        pb.coherent_dedispersion(zz, pb.DM(1.0), chirp=np.ones((1024, 3), np.complex64))


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (2^24 x 8 x 2, DM 56.77): size-independent checks.
    (a) crop geometry; (b) linearity: D(a x1 + b x2) = a D(x1) + b D(x2); (c) a DM = 0 plan on the
    same block is the identity; (d) sparse samples of one series against the oracle's 1-D result."""
    import torch
    n, nchan, npol = 1 << 24, 8, 2
    sr, fc, dm = 50e6, 1.4e9, 56.77
    g = torch.Generator(device="cuda").manual_seed(99)
    def rnd():
        t = torch.randn((n, nchan, npol, 2), generator=g, device="cuda", dtype=torch.float32)
        return torch.view_as_complex(t)
    x1, x2 = rnd(), rnd()
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
    z1 = pb.DualPolarizationSignal(pb.DeviceArray(x1), **kw)
    y1 = pb.coherent_dedispersion(z1, pb.DM(dm))
    assert y1.shape == (14607231 - 1408404, nchan, npol)
    a, b = 0.75 - 0.5j, -1.25 + 0.25j
    y2 = pb.coherent_dedispersion(pb.DualPolarizationSignal(pb.DeviceArray(x2), **kw), pb.DM(dm))
    y12 = pb.coherent_dedispersion(pb.DualPolarizationSignal(pb.DeviceArray(a * x1 + b * x2), **kw), pb.DM(dm))
    lin = a * y1.data.tensor + b * y2.data.tensor
    err = (torch.linalg.vector_norm(y12.data.tensor - lin) / torch.linalg.vector_norm(lin)).item()
    assert err < 2e-6, f"linearity residual {err:.2e}"
    del y2, y12, lin, x2
    y0 = pb.coherent_dedispersion(z1, pb.DM(0.0))
    err0 = (torch.linalg.vector_norm(y0.data.tensor - x1) / torch.linalg.vector_norm(x1)).item()
    assert y0.shape == (n, nchan, npol) and err0 < 2e-6, f"identity residual {err0:.2e}"
    del y0
    # the series-major (time-fastest) device layout gives the same bits with two kernels fewer
    zs = pb.DualPolarizationSignal(pb.DeviceArray(x1).to_series_major(), **kw)
    ys = pb.coherent_dedispersion(zs, pb.DM(dm))
    assert ys.data.series_major_pitch() is not None
    assert same_kernels_or_close(ys.data.tensor.cpu().numpy(), y1.data.tensor.cpu().numpy())
    del zs, ys
    # one series against the oracle (1-D, 2^24: a few seconds of CPU)
    c, p = 5, 1
    xs = x1[:, c, p].cpu().numpy().reshape(-1, 1)
    f = orc.channel_freqs(fc, sr, nchan)[c]
    chirp = orc.transfer_function(dm, n, 1 / sr, f, fc).reshape(-1, 1)
    import scipy.fft
    ref = scipy.fft.ifft(scipy.fft.fft(xs, axis=0) * chirp, axis=0)[1408404:14607231, 0]
    got = y1.data.tensor[:, c, p].cpu().numpy()
    e = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert e < RTOL_L2, f"series ({c},{p}) relative L2 {e:.2e}"


def test_detect_scrunch_one_tile_many_series():
    """One-tile plans with many series (layout pass + planar rows + layout pass): the scrunching detect tail reads the planar
    copy, the voltages are not stored."""
    shape, dm, sr, fc = (1 << 14, 128, 2), 2.0, 1e6, 1e9
    x = orc.synthetic_block(shape, 15)
    z = make_signal(x, sr, fc)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, sr, fc)
    for mode, k in (("I", 64), ("linear", 128)):
        got, start = pb.dedisperse_detect(z.to_device(), pb.DM(dm), mode=mode, nscrunch=k)
        want = orc.scrunch(orc.to_stokes(yr, "linear")[:, :, 0] if mode == "I" else orc.to_stokes(yr, mode), k)
        got = np.asarray(got)
        assert start == s0 and got.shape == want.shape
        assert_detect_close(got, yr, mode, k)


@pytest.mark.parametrize("mode,k", [("I", 1024), ("linear", 256), ("circular", 64), ("intensity", 128),
                                    ("I", 48), ("linear", 1)])
def test_detect_scrunch_modes(mode, k):
    """Fused (k % 64 == 0, planar5) and unfused detect tails against the oracle."""
    shape, dm, sr, fc = (1 << 17, 4, 2), 20.0, 1e6, 1e9
    x = orc.synthetic_block(shape, 12)
    z = make_signal(x, sr, fc)
    got, start = pb.dedisperse_detect(z, pb.DM(dm), mode=mode, nscrunch=k)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, sr, fc)
    if mode == "intensity":
        want = orc.scrunch(orc.to_intensity(yr), k)
    elif mode == "I":
        want = orc.scrunch(orc.to_stokes(yr, "linear")[:, :, 0], k)
    else:
        want = orc.scrunch(orc.to_stokes(yr, mode), k)
    assert start == s0 and got.shape == want.shape and got.dtype == np.float32
    assert_detect_close(got, yr, mode, k)


@pytest.mark.parametrize("shape,chunk,dm,mode,ns", [((1 << 21, 2, 2), 1 << 18, 30.0, "I", 64), ((1 << 21, 2, 2), 1 << 18, 30.0, "linear", 1),
                                                    ((3 << 19, 4), 1 << 17, 10.0, "intensity", 128),
                                                    ((1 << 22, 2, 2), 1 << 20, 80.0, "I", 1024),     # the detecting column pass
                                                    ((1 << 21, 2, 2), 1 << 18, 30.0, "circular", 256)])
def test_stream_detected(shape, chunk, dm, mode, ns):
    """A filterbank stream (pbh_plan_stream_detect): the detected, scrunched valid regions of the chunks, contiguous in time --
    detect + scrunch of what the voltage stream returns for the same (shortened) valid region, chunk by chunk, and the
    download is the detected rows only."""
    from pulsarbat_amd.transforms.dedispersion import _crop_bounds, _plan_for
    sr, fc = 1e6, 1e9
    x = orc.synthetic_block(shape, 23)
    z = make_signal(x, sr, fc)
    got, start, ms = pb.coherent_dedispersion_stream(z, pb.DM(dm), chunk=chunk, detect=mode, nscrunch=ns)
    head = z[:chunk]
    s0, s1 = _crop_bounds(head, pb.DM(dm), head.center_freq)
    s1 -= (s1 - s0) % ns
    hop = s1 - s0
    nchunk = (shape[0] - chunk) // hop + 1
    assert start == s0 and ms > 0 and len(got) == nchunk * (hop // ns) and got.dtype == np.float32
    # the same chunks through the oracle (one reference call per chunk, cropped to the shortened valid region)
    want, volts = [], []
    for k in range(nchunk):
        yk = orc.coherent_dedispersion(x[k * hop:k * hop + chunk], dm, sr, fc)[0][:hop]
        d = orc.to_intensity(yk) if mode == "intensity" else (orc.to_stokes(yk, "linear")[:, :, 0] if mode == "I" else orc.to_stokes(yk, mode))
        want.append(orc.scrunch(d, ns))
        volts.append(yk)
    want = np.concatenate(want, axis=0)
    assert got.reshape(want.shape).shape == want.shape
    assert_detect_close(got, np.concatenate(volts, axis=0), mode, ns)      # (hop is a multiple of ns: the sums do not straddle chunks)
    plan, _ = _plan_for(head, pb.DM(dm), head.center_freq, (s0, s1))
    st = plan.stream_stats()
    assert st["d2h_bytes"] == got.nbytes and st["h2d_bytes"] == x.nbytes - (shape[0] - (chunk + (nchunk - 1) * hop)) * x[0].nbytes
    # ... and the plan is back to voltages for the next caller
    y, _ = pb.coherent_dedispersion_stream(z[:chunk + hop], pb.DM(dm), chunk=chunk)
    assert np.asarray(y).dtype == np.complex64


@pytest.mark.parametrize("shape,chunk,dm", [((1 << 18, 4, 2), 1 << 15, 20.0), ((300000, 2, 2), 1 << 16, 50.0),
                                            ((1 << 16, 3), 1 << 14, 5.0),
                                            ((200000, 4, 2), 20000, 5.0), ((300000, 2, 2), 64800, 8.0)])   # 7-smooth chunks (k_colmix, one and two levels)
def test_stream_overlap_save(shape, chunk, dm):
    """BASELINE configs[3] (overlap-save streaming) at small size: equals the concatenation of
    per-chunk reference calls, and is time-contiguous with the single-call result."""
    sr, fc = 1e6, 1e9
    x = orc.synthetic_block(shape, 21)
    z = make_signal(x, sr, fc, start_time=pb.Time(56000.0, format="mjd"))
    y, ms = pb.coherent_dedispersion_stream(z, pb.DM(dm), chunk=chunk)
    first, start, stop = orc.coherent_dedispersion(x[:chunk], dm, sr, fc)
    hop = stop - start
    nchunk = (shape[0] - chunk) // hop + 1
    assert len(y) == nchunk * hop and ms > 0
    want = np.concatenate([orc.coherent_dedispersion(x[k * hop:k * hop + chunk], dm, sr, fc)[0]
                           for k in range(nchunk)], axis=0)
    assert series_errors(y, want)[0] < RTOL_L2
    assert abs((y.start_time - z.start_time).to_value(u.s) - start / sr) < 1e-12


@pytest.mark.parametrize("epoch", [1, 2, 3, 5])
@pytest.mark.parametrize("shape,chunk,dm", [((1 << 18, 4, 2), 1 << 15, 20.0), ((1 << 16, 3), 1 << 13, 5.0),
                                            ((200000, 1, 1), 1 << 14, 30.0)])
def test_stream_uploads_every_row_once(shape, chunk, dm, epoch, monkeypatch):
    """The overlap of consecutive chunks stays in HBM (two device windows, `epoch` chunks each): the host-to-device volume
    is the input once (transforms.py:101-110: a stream's samples exist once), whatever the epoch length -- short epochs
    exercise the window hand-over, odd row sizes the one-chunk-per-epoch fall-back -- and the result does not change."""
    from pulsarbat_amd.transforms.dedispersion import _crop_bounds, _plan_for
    monkeypatch.setenv("PBH_STREAM_EPOCH", str(epoch))
    sr, fc = 1e6, 1e9
    x = orc.synthetic_block(shape, 22)
    z = make_signal(x, sr, fc)
    head = z[:chunk]
    bounds = _crop_bounds(head, pb.DM(dm), head.center_freq)
    plan, _ = _plan_for(head, pb.DM(dm), head.center_freq, bounds)
    y, ms = plan.dedisperse_stream(x)
    hop = bounds[1] - bounds[0]
    nchunk = (shape[0] - chunk) // hop + 1
    assert nchunk >= 6 and len(y) == nchunk * hop
    want = np.concatenate([orc.coherent_dedispersion(x[k * hop:k * hop + chunk], dm, sr, fc)[0] for k in range(nchunk)], axis=0)
    assert series_errors(y, want)[0] < RTOL_L2
    st = plan.stream_stats()
    row = 8 * int(np.prod(shape[1:]))
    assert st["nchunk"] == nchunk and st["total_ms"] == pytest.approx(ms)
    assert st["h2d_bytes"] == (chunk + (nchunk - 1) * hop) * row            # every row once
    assert st["d2h_bytes"] == nchunk * hop * row
    per_epoch = epoch if (hop * row) % 16 == 0 else 1
    assert st["d2d_bytes"] == ((nchunk - 1) // per_epoch) * (chunk - hop) * row
    assert 0 < st["overlap_efficiency"] <= 1.0 and st["kernel_ms"] > 0


# ---- complex128 (float64 kernels): the reference accepts both dtypes (core.py:742) and keeps
# dtype in = dtype out (tests/test_fft.py:53-54); its own tests feed complex128 ----------------------
# With the HIP-generated chirp the bound is set by complex64 rounding of the chirp (the reference
# rounds it too, dedispersion.py:23): the float64 phase differs from numpy's by ~1 ulp, which flips
# the last float32 bit of a few chirp samples (6e-8 each).  With the oracle's own chirp uploaded the
# float64 kernels agree with pocketfft to ~1e-15.
RTOL_F64 = 1e-9
RTOL_F64_SAME_CHIRP = 2e-14


def check128(shape, dm, sr=1e6, fc=1e9, seed=2, variant="auto", device=False):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5
    z = make_signal(x, sr, fc, start_time=pb.Time(56000.0, format="mjd"))
    zz = z.to_device() if device else z
    y = pb.coherent_dedispersion(zz, pb.DM(dm), variant=variant)
    yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    assert y.dtype == np.complex128 and type(y.data) is type(zz.data)
    assert y.shape == yr.shape
    l2, mx = series_errors(y, yr)
    assert l2 < RTOL_F64, f"relative L2 {l2:.2e}"
    chirp = orc.chirp_from_signal(dm, x.shape, sr, fc)
    y2 = pb.coherent_dedispersion(zz, pb.DM(dm), variant=variant, chirp=chirp)
    l2, mx = series_errors(y2, yr)
    assert l2 < RTOL_F64_SAME_CHIRP, f"relative L2 with the oracle's chirp {l2:.2e}"
    return y


@pytest.mark.parametrize("shape,dm", [((16, 2), 0.001), ((4096, 4, 2), 5.0), ((8192, 4), 50.0), ((1 << 13, 3, 2), 10.0),
                                      ((1 << 14, 2, 2), 10.0), ((1 << 16, 2, 2), 20.0), ((1 << 18, 2), 20.0),
                                      ((1 << 20, 1, 2), 40.0),
                                      # two row tiles per workgroup: the case that exposed the gfx950 wide-store hazard
                                      ((1 << 20, 2, 2), 40.0), ((1 << 19, 4, 2), 25.0)])
def test_c128_parity(shape, dm):
    check128(shape, dm)


@pytest.mark.parametrize("variant", ["direct3", "planar5", "block3"])
def test_c128_variants(variant):
    check128((1 << 17, 4, 2), 20.0, variant=variant, device=True)


@pytest.mark.parametrize("shape,dm", [((1000, 2), 1.0), ((30000, 2, 2), 10.0), ((65537, 1), 5.0)])
def test_c128_arbitrary_lengths(shape, dm):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    z = make_signal(x, 1e6, 1e9)
    y = pb.coherent_dedispersion(z, pb.DM(dm))
    yr, _, _ = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    assert y.dtype == np.complex128 and series_errors(y, yr)[0] < RTOL_F64


@pytest.mark.parametrize("seed", [4, 8, 15, 16, 23, 42])
def test_c128_reversibility_reference_tolerance(seed):
    """reference tests/test_dedispersion.py:73-98 at its own tolerance (atol 3e-8, complex128 data)."""
    import scipy.signal
    ref, sr, dm = 600e6, 400e6, 0.01
    N, M = 2 ** 18, 2 ** 12
    R = np.random.default_rng(seed=seed)
    x = R.standard_normal(N) + 1j * R.standard_normal(N)
    x *= np.exp(-(((np.arange(N) - N // 2) / M) ** 2))
    sos = scipy.signal.butter(10, 0.45, "lowpass", fs=1.0, output="sos")
    x = scipy.signal.sosfilt(sos, x).reshape(-1, 1)
    sig = make_signal(x, sr, ref, start_time=pb.Time(56000.0, format="mjd"))
    temp = pb.coherent_dedispersion(sig, pb.DM(dm))          # length 2^18 - crop: not a power of two
    sig2 = pb.coherent_dedispersion(temp, -pb.DM(dm))        # -> Bluestein, exactly as the reference chains them
    toffset = sig2.start_time - sig.start_time
    noffset = int(np.rint((toffset * sig.sample_rate).to_value(u.one)))
    sig1 = sig[noffset:noffset + len(sig2)]
    res = np.array(sig1) - np.array(sig2)
    assert np.allclose(res, 0, atol=3e-8)


def test_c128_detect_and_fft():
    import scipy.fft
    rng = np.random.default_rng(9)
    shape = (1 << 15, 2, 2)
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    z = make_signal(x, 1e6, 1e9)
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(10.0))
    i = y.to_intensity()
    assert i.dtype == np.float64                      # reference tests/test_radio_signal.py:142-172
    assert np.allclose(np.asarray(i.data), orc.to_intensity(np.asarray(y)), rtol=1e-12)
    s = y.to_stokes()
    assert s.dtype == np.float64
    assert np.allclose(np.asarray(s.data), orc.to_stokes(np.asarray(y), "linear"), rtol=1e-10, atol=1e-12)
    got, start = pb.dedisperse_detect(z, pb.DM(10.0), mode="I", nscrunch=64)
    yr, s0, _ = orc.coherent_dedispersion(x, 10.0, 1e6, 1e9)
    assert got.dtype == np.float64 and start == s0
    assert np.allclose(got, orc.scrunch(orc.to_stokes(yr, "linear")[:, :, 0], 64), rtol=1e-11)
    d = pb.DeviceArray.from_host(x[:4096].reshape(4096, -1).copy())
    for name in ("fft", "ifft"):
        g = getattr(pb.fft, name)(d, axis=0)
        w = getattr(scipy.fft, name)(x[:4096].reshape(4096, -1), axis=0)
        assert g.dtype == np.complex128 and np.linalg.norm(np.asarray(g) - w) / np.linalg.norm(w) < 1e-13


@pytest.mark.parametrize("n,batch", [(1 << 15, 6), (1 << 18, 2), (1000, 5), (12345, 3), (65537, 1), (20, 4)])
@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_fft_dispatch_any_length(n, batch, dtype):
    """pb.fft.fft / ifft on device arrays for lengths beyond one tile and non powers of two
    (reference tests/test_fft.py:41-61: same container type, same dtype, allclose to scipy)."""
    import scipy.fft
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((n, batch)) + 1j * rng.standard_normal((n, batch))).astype(dtype)
    d = pb.DeviceArray.from_host(x)
    tol = 2e-6 if dtype == np.complex64 else 1e-12
    for name in ("fft", "ifft"):
        got = getattr(pb.fft, name)(d, axis=0)
        want = getattr(scipy.fft, name)(x, axis=0)
        assert type(got) is type(d) and got.dtype == want.dtype
        assert np.linalg.norm(np.asarray(got) - want) / np.linalg.norm(want) < tol
    back = pb.fft.ifft(pb.fft.fft(d, axis=0), axis=0)
    assert np.linalg.norm(np.asarray(back) - x) / np.linalg.norm(x) < 2 * tol


# ---- series-major (time-fastest) device arrays: layout passes skipped (pbh_dedisperse_layout) --------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,dm,dtype", [((1 << 16, 4, 2), 20.0, np.complex64), ((1 << 18, 3, 2), 30.0, np.complex64),
                                            ((1 << 17, 5), 10.0, np.complex64), ((1 << 16, 2, 2), 20.0, np.complex128)])
def test_series_major_io(shape, dm, dtype):
    rng = np.random.default_rng(11)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    z = make_signal(x, 1e6, 1e9, start_time=pb.Time(56000.0, format="mjd"))
    yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    tol = RTOL_L2 if dtype == np.complex64 else RTOL_F64
    zd = z.to_device()
    zs = z.to_device(series_major=True)
    assert zs.data.series_major_pitch() is not None and not zs.data.tensor.is_contiguous()
    assert np.array_equal(np.asarray(zs.data), x)
    # series-major in -> series-major out
    y = pb.coherent_dedispersion(zs, pb.DM(dm))
    assert y.shape == yr.shape and y.data.series_major_pitch() is not None
    assert series_errors(y, yr)[0] < tol
    assert abs((y.start_time - z.start_time).to_value(u.s) - start / 1e6) < 1e-12
    # mixed ends through the plan
    from pulsarbat_amd.transforms.dedispersion import _prepare
    plan, xin, _, _ = _prepare(zs, pb.DM(dm), None, None, "auto", allow_series=True)
    y2 = plan.dedisperse(xin, out_layout="sample")
    assert y2.tensor.is_contiguous() and series_errors(y2, yr)[0] < tol
    y3 = plan.dedisperse(zd.data, out_layout="series")
    assert y3.series_major_pitch() is not None and series_errors(y3, yr)[0] < tol
    # same kernels in the middle, same arithmetic: bit-identical to the sample-major path whenever that path
    # uses the persistent column kernel too (column transforms of 64 points and more)
    y0 = plan.dedisperse(zd.data)
    if plan.info["n1"] >= 64:
        assert same_kernels_or_close(y0, y3)
        import os
        os.environ["PBH_FD4"] = "0"      # the five-pass schedule: the same kernels as the series-major route, bit for bit
        try:
            assert np.array_equal(np.asarray(plan.dedisperse(zd.data)), np.asarray(y3))
        finally:
            os.environ.pop("PBH_FD4", None)
    else:
        assert series_errors(y0, np.asarray(y3))[0] < (2e-6 if dtype == np.complex64 else 1e-13)


@pytest.mark.gpu
def test_series_major_fallback_small_and_bluestein():
    """Lengths without the layout-aware path (one tile, Bluestein) take one contiguous copy instead."""
    rng = np.random.default_rng(12)
    for n in (4096, 30000):
        x = (rng.standard_normal((n, 2, 2)) + 1j * rng.standard_normal((n, 2, 2))).astype(np.complex64)
        z = make_signal(x, 1e6, 1e9)
        zs = type(z).like(z, z.to_device().data.to_series_major())
        y = pb.coherent_dedispersion(zs, pb.DM(5.0))
        yr, _, _ = orc.coherent_dedispersion(x, 5.0, 1e6, 1e9)
        assert series_errors(y, yr)[0] < RTOL_L2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,shape,dm", [(np.complex128, (1 << 20, 2, 2), 40.0), (np.complex64, (1 << 21, 4, 2), 30.0)])
def test_repeatable_bit_for_bit(dtype, shape, dm):
    """The same call gives the same bits every time (tile hand-out order, prefetch timing and pinned stores
    must not leak into the results); persistent loops run several tiles per workgroup at these sizes."""
    rng = np.random.default_rng(2)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    z = make_signal(x, 1e6, 1e9).to_device()
    ref = {}
    for it in range(6):
        for variant in ("planar5", "direct3"):
            y = np.asarray(pb.coherent_dedispersion(z, pb.DM(dm), variant=variant))
            if variant in ref:
                assert np.array_equal(y, ref[variant]), f"{variant}: repeat {it} differs from the first run"
            else:
                ref[variant] = y
    yr, _, _ = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    for variant, y in ref.items():
        assert series_errors(y, yr)[0] < (RTOL_L2 if dtype == np.complex64 else RTOL_F64)


_ORACLE_2P24 = {}


@pytest.mark.gpu
@pytest.mark.parametrize("mode,nscrunch,dm,ref", [("I", 1024, 56.77, None), ("I", 64, 56.77, None), ("linear", 1024, 56.77, None),
                                                  ("intensity", 16384, 30.0, None),
                                                  ("I", 1024, 20.0, "top"), ("intensity", 256, 100.0, "bottom"),
                                                  ("circular", 256, 20.0, "top"),
                                                  ("I", 4096, 3.0, None), ("linear", 16384, 101.0, None)])
def test_detect_inside_the_column_pass(mode, nscrunch, dm, ref):
    """N = 2^24 (1024-row column tiles): |z|^2 and Stokes I are summed inside the inverse column pass and the dedispersed
    voltages are never stored (k_colq<.., DET> + k_detect_reduce; the four-parameter modes keep the read pass over the stored
    voltages, k_detect_planar, and are checked here at the same size).  Same numbers as detecting the stored voltages, for
    crop starts that are and are not multiples of the 16-column tile width (ref_freq at the band edges: start or stop at
    the end of the block), scrunch factors from one tile-row fraction to a whole row, sample-major and series-major input."""
    n, nchan, npol, sr, fc = 1 << 24, 2, 2, 50e6, 1.4e9
    rng = np.random.default_rng(61)
    x = rng.standard_normal((n, nchan, npol, 2), dtype=np.float32).view(np.complex64)[..., 0]
    x *= (1 + np.arange(nchan * npol, dtype=np.float32).reshape(nchan, npol))   # every series its own power
    x[..., 1] += np.complex64(0.5 + 0.3j) * x[..., 0]                           # ... and the pols correlated: U, V != 0
    z = make_signal(x, sr, fc).to_device()
    rf = None if ref is None else (fc + sr * nchan / 2 if ref == "top" else fc - sr * nchan / 2) * u.Hz
    # the ORACLE's voltages (one pocketfft run per (dm, ref), shared by the cases; float64 power sums: tests/_detect_check.py)
    key = (dm, ref)
    if key not in _ORACLE_2P24:
        _ORACLE_2P24.clear()               # one 1-GB result at a time
        _ORACLE_2P24[key] = orc.coherent_dedispersion(x, dm, sr, fc, ref_freq_hz=None if rf is None else rf.to_value(u.Hz),
                                                      workers=min(16, os.cpu_count() or 1))
    yr, s0, _ = _ORACLE_2P24[key]
    got, start = pb.dedisperse_detect(z, pb.DM(dm), ref_freq=rf, mode=mode, nscrunch=nscrunch)
    got = np.asarray(got)
    assert start == s0 and got.dtype == np.float32
    if ref is None:
        assert start % 16 != 0, "pick a DM whose crop start is not tile-aligned"
    assert_detect_close(got, yr, mode, nscrunch, what=f"start {start}")
    zs = type(z).like(z, z.data.to_series_major())
    got_s, start_s = pb.dedisperse_detect(zs, pb.DM(dm), ref_freq=rf, mode=mode, nscrunch=nscrunch)
    assert start_s == start and same_kernels_or_close(got_s, got, rel=2e-6)
    if mode == "I" and nscrunch == 1024 and ref is None:
        # a chirp given by the caller (complex64 rows, k_row instead of the phase-row kernel in front of the detecting pass)
        c = pb.DM(dm).chirp_from_signal(z)   # device-resident, (n, nchan, 1)
        got_c, start_c = pb.dedisperse_detect(z, pb.DM(dm), chirp=c, mode=mode, nscrunch=nscrunch)
        assert start_c == start and np.allclose(np.asarray(got_c), got, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,mode", [((1 << 18, 4, 2), "I"), ((1 << 18, 4, 2), "linear"), ((1 << 18, 8, 2), "circular"),
                                        ((1 << 18, 1, 2), "intensity"), ((1 << 19, 16), "intensity"), ((1 << 18, 64, 2), "I"),
                                        ((1 << 18, 3, 2), "I"),
                                        ((3 << 17, 2, 2), "linear"), ((400000, 2, 2), "I"), ((64800, 4, 2), "intensity"),   # m 2^k, 7-smooth
                                        ((20000, 8, 2), "circular"), ((52488, 2, 2), "I"),                                 # ... one level; mixed rows
                                        ((1 << 18, 16, 2), "I"), ((1 << 17, 32, 2), "circular"), ((1 << 18, 32), "intensity"),
                                        ((1 << 16, 128, 2), "I"), ((1 << 15, 256, 2), "linear"), ((1 << 17, 12, 2), "circular"),   # two-axis tiles
                                        ((1 << 17, 100), "intensity"),
                                        ((1 << 14, 128, 2), "I"), ((1 << 14, 128), "intensity")])   # one-tile plans with many series
def test_detect_in_the_last_layout_pass(shape, mode):
    """nscrunch = 1: to_intensity / to_stokes of the dedispersed voltages at full time resolution, computed by the last layout
    pass from the planar workspace (k_reinterleave_p2<.., DET>; the rows beyond the last whole tile by k_detect_planar) -- the
    voltages are never stored.  Many series and even counts that are no power of two: the two-axis tile kernel (k_reint_blk<..,
    DET>)."""
    dm, sr, fc = 12.0, 1e6, 1e9
    x = orc.synthetic_block(shape, 33)
    z = make_signal(x, sr, fc)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, sr, fc)
    want = orc.to_intensity(yr) if mode == "intensity" else (orc.to_stokes(yr, "linear")[:, :, 0] if mode == "I" else orc.to_stokes(yr, mode))
    for dev in (False, True):
        got, start = pb.dedisperse_detect(z.to_device() if dev else z, pb.DM(dm), mode=mode, nscrunch=1)
        got = np.asarray(got)
        if len(shape) == 2:
            got = got[..., 0]   # (dedisperse_detect reports an explicit polarisation axis of one)
        assert start == s0 and got.shape == want.shape and got.dtype == np.float32
        assert_detect_close(got, yr, mode, 1)
    # a series-major device array: read by the first column pass as it is (the non-power-of-two series count falls back to a copy)
    zd = z.to_device()
    zs = type(zd).like(zd, zd.data.to_series_major())
    got_s, start_s = pb.dedisperse_detect(zs, pb.DM(dm), mode=mode, nscrunch=1)
    got_s = np.asarray(got_s)
    assert start_s == s0
    assert_detect_close(got_s[..., 0] if len(shape) == 2 else got_s, yr, mode, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,mode,nscrunch,dm", [(22, "I", 1024, 30.0), (22, "intensity", 64, 5.0), (23, "I", 256, 56.77),
                                                    (23, "intensity", 16384, 10.0), (22, "I", 4096, 0.7),
                                                    (21, "I", 1024, 3.0), (20, "intensity", 256, 1.5), (20, "I", 2048, 0.4)])
def test_detect_inside_the_column_pass_wider_tiles(log2n, mode, nscrunch, dm):
    """The same for 2^20 ... 2^23 samples (64- to 512-row column tiles of 256 to 32 columns: sixteen to two 16-column groups per
    tile, the scrunch boundary inside any of them), against the oracle."""
    n, nchan, npol, sr, fc = 1 << log2n, 3, 2, 10e6, 1.2e9
    x = orc.synthetic_block((n, nchan, npol), 71)
    z = make_signal(x, sr, fc)
    got, start = pb.dedisperse_detect(z.to_device(), pb.DM(dm), mode=mode, nscrunch=nscrunch)
    yr, s0, _ = orc.coherent_dedispersion(x, dm, sr, fc)
    want = orc.scrunch(orc.to_intensity(yr) if mode == "intensity" else orc.to_stokes(yr, "linear")[:, :, 0], nscrunch)
    got = np.asarray(got)
    assert start == s0 and got.shape == want.shape
    assert_detect_close(got, yr, mode, nscrunch)
    zs = z.to_device()
    zs = type(zs).like(zs, zs.data.to_series_major())
    got_s, _ = pb.dedisperse_detect(zs, pb.DM(dm), mode=mode, nscrunch=nscrunch)
    assert same_kernels_or_close(got_s, got, rel=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,nscrunch", [("I", 64), ("linear", 128), ("intensity", 64), ("I", 16)])
def test_series_major_dedisperse_detect(mode, nscrunch):
    """dedisperse_detect on a series-major device array: same numbers as on the contiguous array."""
    rng = np.random.default_rng(13)
    shape = (1 << 17, 4, 2)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(np.complex64)
    z = make_signal(x, 1e6, 1e9).to_device()
    zs = type(z).like(z, z.data.to_series_major())
    a, s0 = pb.dedisperse_detect(z, pb.DM(15.0), mode=mode, nscrunch=nscrunch)
    b, s1 = pb.dedisperse_detect(zs, pb.DM(15.0), mode=mode, nscrunch=nscrunch)
    assert s0 == s1 and a.shape == b.shape
    # (N1 = 8 here: the sample-major call uses the one-tile column kernel, the series-major one the persistent kernel)
    assert np.allclose(np.asarray(a), np.asarray(b), rtol=2e-5, atol=2e-5 * np.abs(np.asarray(a)).max())


@pytest.mark.gpu
def test_series_major_arrays_in_other_ops():
    """A series-major device array is an ordinary (strided) array for everything else: detection, FFT,
    slicing and host copies give the same numbers as the contiguous array."""
    rng = np.random.default_rng(14)
    x = (rng.standard_normal((4096, 3, 2)) + 1j * rng.standard_normal((4096, 3, 2))).astype(np.complex64)
    z = make_signal(x, 1e6, 1e9).to_device()
    zs = type(z).like(z, z.data.to_series_major())
    assert np.array_equal(np.asarray(zs.to_intensity().data), np.asarray(z.to_intensity().data))
    assert np.array_equal(np.asarray(zs.to_stokes().data), np.asarray(z.to_stokes().data))
    assert np.array_equal(np.asarray(pb.fft.fft(zs.data, axis=0)), np.asarray(pb.fft.fft(z.data, axis=0)))
    part = zs[100:2148]
    assert part.data.series_major_pitch() == zs.data.series_major_pitch()
    assert np.array_equal(np.asarray(part.data), x[100:2148])


@pytest.mark.gpu
@pytest.mark.parametrize("nchan,npol,dtype", [(32, 2, np.complex64), (64, 2, np.complex64), (128, 1, np.complex64),
                                              (48, 2, np.complex64), (256, 2, np.complex64), (96, 2, np.complex64),
                                              (331, 2, np.complex64), (257, 1, np.complex64), (150, 1, np.complex128),
                                              (129, 2, np.complex128)])
def test_many_series(nchan, npol, dtype):
    """Wide blocks: S = 64, 128 (row transposes), S > 128 (two-axis tiles: 512, 192, 662 series; complex128 any S),
    odd S > 128 in complex64 and S = 96 (generic kernels)."""
    rng = np.random.default_rng(15)
    shape = (1 << 15, nchan, npol) if npol > 1 else (1 << 15, nchan)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    z = make_signal(x, 1e6, 1e9)
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(3.0))
    yr, _, _ = orc.coherent_dedispersion(x, 3.0, 1e6, 1e9)
    # complex128: a 150-300 MHz wide band puts large float64 phases at the band edges -> a few more last-bit
    # flips of the complex64-rounded chirp than RTOL_F64's narrow-band cases (see the comment above RTOL_F64)
    assert y.shape == yr.shape and series_errors(y, yr)[0] < (RTOL_L2 if dtype == np.complex64 else 1e-8)


# ---- long blocks: column transform split into a radix-P stage and P row blocks (k_radix_p + k_colq) ----------
@pytest.fixture
def small_qmax(monkeypatch):
    """Force the split at small sizes: at most 32 (complex64) / 16 (complex128) rows per column tile."""
    from pulsarbat_amd.transforms.dedispersion import clear_plan_cache
    clear_plan_cache()
    yield monkeypatch
    monkeypatch.delenv("PBH_QMAX", raising=False)
    clear_plan_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dm,dtype,qmax,nk", [
    ((1 << 20, 2, 2), 20.0, np.complex64, 32, 7),    # N1 = 64  -> P = 2
    ((1 << 21, 3, 2), 30.0, np.complex64, 32, 7),    # N1 = 128 -> P = 4
    ((1 << 22, 1, 2), 30.0, np.complex64, 32, 7),    # N1 = 256 -> P = 8, S = 2 (would be direct3)
    ((1 << 23, 1), 30.0, np.complex64, 32, 7),       # N1 = 512 -> P = 16, one series
    ((1 << 19, 2, 2), 10.0, np.complex128, 16, 7),   # float64: N2 = 2^13, N1 = 64 -> P = 4
    ((1 << 20, 4), 10.0, np.complex128, 32, 7),      # N1 = 128 -> P = 4
    ((1 << 19, 1, 2), 10.0, np.complex128, 32, 7),   # N1 = 64 -> P = 2: series-major / detect callers run it unsplit
])
def test_split_column_transform(small_qmax, shape, dm, dtype, qmax, nk):
    small_qmax.setenv("PBH_QMAX", str(qmax))
    rng = np.random.default_rng(21)
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    z = make_signal(x, 1e6, 1e9, start_time=pb.Time(56000.0, format="mjd"))
    yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    tol = RTOL_L2 if dtype == np.complex64 else RTOL_F64
    from pulsarbat_amd.transforms.dedispersion import _prepare
    plan, xin, _, _ = _prepare(z.to_device(), pb.DM(dm), None, None, "auto")
    assert plan.info["nkernel"] in (nk, nk - 2)   # 5 when the radix stage is folded into the layout passes
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(dm))
    assert y.shape == yr.shape and series_errors(y, yr)[0] < tol
    # user chirp (uploaded in the reference's order), chirp download, and the fused detect tail use the same row order
    chirp = orc.chirp_from_signal(dm, x.shape, 1e6, 1e9)
    y2 = pb.coherent_dedispersion(z.to_device(), pb.DM(dm), chirp=chirp)
    assert series_errors(y2, yr)[0] < tol
    got = pb.DM(dm).chirp_from_signal(z.to_device())
    assert np.abs(np.asarray(got).reshape(chirp.shape) - chirp).max() < 2.5e-7
    if x.ndim == 3:
        a, s0 = pb.dedisperse_detect(z.to_device(), pb.DM(dm), mode="I", nscrunch=64)
        assert s0 == start
        assert_detect_close(a, yr, "I", 64)
    # series-major arrays: the radix stage reads / writes the caller's arrays (5 kernels), same bits
    zs = type(z).like(z, z.to_device().data.to_series_major())
    ys = pb.coherent_dedispersion(zs, pb.DM(dm))
    assert series_errors(ys, yr)[0] < tol
    assert int(np.prod(x.shape[1:])) == 1 or ys.data.series_major_pitch() is not None   # one series: both layouts coincide
    if x.ndim == 3:
        b2, _ = pb.dedisperse_detect(zs, pb.DM(dm), mode="I", nscrunch=64)
        # (the folded and the stand-alone radix stage contract their multiply-adds differently: last-bit differences)
        assert np.allclose(np.asarray(a), np.asarray(b2), rtol=2e-5)
        assert_detect_close(b2, yr, "I", 64)


@pytest.mark.gpu
def test_split_column_transform_nonpow2(small_qmax):
    """A non-power-of-two length whose convolution plan (L = 2^21) uses the split: filter rows permuted."""
    small_qmax.setenv("PBH_QMAX", "32")
    rng = np.random.default_rng(22)
    x = (rng.standard_normal((700001, 2, 2)) + 1j * rng.standard_normal((700001, 2, 2))).astype(np.complex64)
    z = make_signal(x, 1e6, 1e9)
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(7.0))
    yr, _, _ = orc.coherent_dedispersion(x, 7.0, 1e6, 1e9)
    assert series_errors(y, yr)[0] < RTOL_L2
    f = pb.fft.fft(z.to_device().data, axis=0)      # plain ring transform (its sub-plan is split too)
    assert np.linalg.norm(np.asarray(f) - np.fft.fft(x, axis=0)) / np.linalg.norm(np.fft.fft(x, axis=0)) < 2e-6


# ---- lengths m * 2^k (m = 3, 5, 7) run natively: the odd factor is the radix-P stage of the split column pass ------
@pytest.mark.gpu
@pytest.mark.parametrize("n,shape_tail,dtype", [
    (3 << 19, (2, 2), np.complex64), (5 << 19, (2, 2), np.complex64), (7 << 19, (1, 2), np.complex64),
    (3 << 20, (3,), np.complex64),            # 3 series: stand-alone radix stage + two-axis layout tiles
    (5 << 17, (2, 2), np.complex128), (3 << 18, (2,), np.complex128),
])
def test_odd_factor_lengths(n, shape_tail, dtype):
    from pulsarbat_amd.transforms.dedispersion import _prepare, clear_plan_cache
    clear_plan_cache()
    rng = np.random.default_rng(31)
    shape = (n,) + shape_tail
    x = ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 2 ** -0.5).astype(dtype)
    z = make_signal(x, 1e6, 1e9, start_time=pb.Time(56000.0, format="mjd"))
    dm = 30.0
    yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    tol = RTOL_L2 if dtype == np.complex64 else RTOL_F64
    plan, _, _, _ = _prepare(z.to_device(), pb.DM(dm), None, None, "auto")
    assert plan.info["nkernel"] in (5, 7)            # a native multi-pass plan, not the convolution detour
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(dm))
    assert y.shape == yr.shape and series_errors(y, yr)[0] < tol
    assert abs((y.start_time - z.start_time).to_value(u.s) - start / 1e6) < 1e-12
    chirp = orc.chirp_from_signal(dm, x.shape, 1e6, 1e9)
    y2 = pb.coherent_dedispersion(z.to_device(), pb.DM(dm), chirp=chirp)
    assert series_errors(y2, yr)[0] < tol
    got = pb.DM(dm).chirp_from_signal(z.to_device())
    assert np.abs(np.asarray(got).reshape(chirp.shape) - chirp).max() < 2.5e-7


@pytest.mark.gpu
@pytest.mark.parametrize("n", [700001, 1200007, 1700003])
def test_arbitrary_length_uses_short_convolution(n):
    """2N - 1 = 1.4 M -> convolution length 3 * 2^19 (not 2^21); 2.4 M -> 5 * 2^19 (not 2^22); 3.4 M -> 7 * 2^19."""
    rng = np.random.default_rng(32)
    x = (rng.standard_normal((n, 1, 2)) + 1j * rng.standard_normal((n, 1, 2))).astype(np.complex64)
    z = make_signal(x, 1e6, 1e9)
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(7.0))
    yr, _, _ = orc.coherent_dedispersion(x, 7.0, 1e6, 1e9)
    assert series_errors(y, yr)[0] < RTOL_L2


# ---- pb.fft.fft / ifft of native lengths beyond one tile: multi-pass transform + natural-order output pass ---------
def _fft_check(x, tol):
    import scipy.fft
    d = pb.DeviceArray.from_host(x)
    for name in ("fft", "ifft"):
        got = np.asarray(getattr(pb.fft, name)(d, axis=0))
        want = getattr(scipy.fft, name)(x.astype(np.complex128), axis=0)
        assert got.dtype == x.dtype and got.shape == x.shape
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < tol
    # a delta at n0 -> exp(-2 pi i k n0 / N): catches any bin permutation exactly
    n0 = x.shape[0] // 3 + 1
    e = np.zeros_like(x)
    e[n0] = 1
    got = np.asarray(pb.fft.fft(pb.DeviceArray.from_host(e), axis=0))
    k = np.arange(x.shape[0])
    want = np.exp(-2j * np.pi * ((k * n0) % x.shape[0]) / x.shape[0])
    assert np.abs(got - want.reshape((-1,) + (1,) * (x.ndim - 1))).max() < (1e-5 if x.dtype == np.complex64 else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,tail,dtype", [
    (1 << 15, (2,), np.complex64), (1 << 19, (2, 2), np.complex64), (1 << 22, (4,), np.complex64),
    (3 << 19, (2,), np.complex64), (5 << 19, (2, 2), np.complex64), (7 << 20, (2,), np.complex64),
    (1 << 20, (3, 2), np.complex64),          # 6 series: two-axis layout tiles in front
    (1 << 19, (3,), np.complex64), (1 << 17, (1,), np.complex64),   # odd batch: 8-byte stores in the output pass
    (1 << 14, (3,), np.complex128), (1 << 18, (2, 2), np.complex128), (3 << 18, (1,), np.complex128),
])
def test_fft_native_lengths(n, tail, dtype):
    rng = np.random.default_rng(n % 1000 + len(tail))
    x = (rng.standard_normal((n,) + tail) + 1j * rng.standard_normal((n,) + tail)).astype(dtype)
    _fft_check(x, 2e-6 if dtype == np.complex64 else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,tail", [(1 << 20, (2,)), (1 << 22, (8,)), (1 << 21, (6,)), (1 << 22, (1,)), (1 << 21, (3,))])
def test_fft_native_split_column(n, tail, small_qmax):
    """Same with the split column transform (P = 2 .. 8 at these sizes with PBH_QMAX=32): rows permuted."""
    small_qmax.setenv("PBH_QMAX", "32")
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n,) + tail) + 1j * rng.standard_normal((n,) + tail)).astype(np.complex64)
    _fft_check(x, 2e-6)


@pytest.mark.gpu
def test_default_radix2_split():
    """2^25 complex64 samples: N1 = 2048 is split 2 x 1024 by default because the radix-2 stage rides in the layout
    passes (5 kernels).  Series-major arrays and the detect tail have no layout pass to carry it: the same plan
    (chirp rows in split order) runs the unsplit column passes for them."""
    from pulsarbat_amd.transforms.dedispersion import _prepare, clear_plan_cache
    clear_plan_cache()
    shape, dm = (1 << 25, 1, 2), 40.0
    rng = np.random.default_rng(77)
    x = ((rng.standard_normal(shape, dtype=np.float32) + 1j * rng.standard_normal(shape, dtype=np.float32)) * np.float32(2 ** -0.5))
    z = make_signal(x, 2e6, 1e9, start_time=pb.Time(56000.0, format="mjd"))
    yr, start, stop = orc.coherent_dedispersion(x, dm, 2e6, 1e9)
    plan, _, _, _ = _prepare(z.to_device(), pb.DM(dm), None, None, "auto")
    assert plan.info["n1"] == 2048 and plan.info["nkernel"] == 5
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(dm))
    assert y.shape == yr.shape and series_errors(y, yr)[0] < RTOL_L2
    zs = type(z).like(z, z.to_device().data.to_series_major())
    ys = pb.coherent_dedispersion(zs, pb.DM(dm))
    assert ys.data.series_major_pitch() is not None and series_errors(ys, yr)[0] < RTOL_L2
    a, s0 = pb.dedisperse_detect(z.to_device(), pb.DM(dm), mode="I", nscrunch=1024)
    assert s0 == start
    assert_detect_close(a, yr, "I", 1024)
    clear_plan_cache()


@pytest.mark.gpu
def test_trim_releases_cached_transform_plans():
    """pbh_fft_c2c keeps per-thread plans (workspace = data size); pbh_trim frees them and the next call rebuilds."""
    import torch
    from pulsarbat_amd import _hip
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((1 << 20, 8)) + 1j * rng.standard_normal((1 << 20, 8))).astype(np.complex64)
    d = pb.DeviceArray.from_host(x)
    _hip.trim()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    a = np.asarray(pb.fft.fft(d, axis=0))
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 >= x.nbytes          # the cached plan's workspace
    _hip.trim()
    assert torch.cuda.mem_get_info()[0] - free1 >= x.nbytes
    b = np.asarray(pb.fft.fft(d, axis=0))
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_concurrent_threads():
    """Distinct threads dedisperse concurrently (ctypes releases the GIL; a dask-threads-like caller): plans are
    cached per thread, never shared, and every thread gets the oracle's result -- same geometry in all threads,
    which is the case a shared plan would break."""
    import threading
    shape, dm, sr, fc = (1 << 17, 4, 2), 15.0, 1e6, 1e9
    xs = [orc.synthetic_block(shape, 100 + i) for i in range(4)]
    wants = [orc.coherent_dedispersion(x, dm, sr, fc)[0] for x in xs]
    errs, fails = [None] * 4, []

    def work(i):
        try:
            z = make_signal(xs[i], sr, fc)
            worst = 0.0
            for rep in range(6):
                y = pb.coherent_dedispersion(z.to_device() if rep % 2 else z, pb.DM(dm))
                worst = max(worst, series_errors(y, wants[i])[0])
                f = np.asarray(pb.fft.fft(pb.DeviceArray.from_host(xs[i][:1 << 16, 0]), axis=0))
                ref = np.fft.fft(xs[i][:1 << 16, 0], axis=0)
                worst = max(worst, np.linalg.norm(f - ref) / np.linalg.norm(ref))
            errs[i] = worst
        except Exception as exc:   # surfaced below: an exception in a thread must fail the test
            fails.append(repr(exc))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not fails, fails
    assert all(e is not None and e < 2e-6 for e in errs), errs


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(1 << 20) + 1, (5 << 19) + 3])
def test_odd_length_single_series_padding(n):
    """One series of odd length whose convolution plan is a split one (L = 3 * 2^k): the zero padding happens in the
    de-interleave pass, where a 16-byte vector holds two time samples -- the last one must not read the element
    after the input.  The input is a view of a larger buffer whose next element is huge."""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal(n + 1) + 1j * rng.standard_normal(n + 1)).astype(np.complex64)
    x[n] = 1e6 + 1e6j
    big = pb.DeviceArray.from_host(x)
    view = big[:n]
    assert view.data_ptr() == big.data_ptr()
    z = pb.BasebandSignal(pb.DeviceArray(view.tensor.reshape(n, 1)), sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    y = pb.coherent_dedispersion(z, pb.DM(3.0))
    yr, _, _ = orc.coherent_dedispersion(x[:n].reshape(n, 1), 3.0, 1e6, 1e9)
    assert y.shape == yr.shape and np.linalg.norm(np.asarray(y) - yr) / np.linalg.norm(yr) < RTOL_L2
    s = pb.Signal(view, sample_rate=1 * u.kHz)
    got = np.asarray(pb.time_shift(s, 29.09))
    ref, _, _ = orc.time_shift(x[:n], 29.09)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 4e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,tail,dtype,g0", [
    (4233, (3, 2), np.complex64, 1), ((1 << 15) + 1, (1,), np.complex64, 3), (1 << 16, (4, 2), np.complex64, 1),
    (3 << 19, (2,), np.complex64, 2), (1 << 20, (1, 2), np.complex64, 0), (100003, (2, 3), np.complex64, 1),
    ((1 << 14) + 3, (1,), np.complex128, 1), (1 << 17, (2, 2), np.complex128, 3), (2999, (5,), np.complex128, 2),
])
def test_no_access_outside_the_arrays(n, tail, dtype, g0):
    """The input is a view into a buffer of NaNs (at a sample offset that breaks 16-byte alignment for odd series
    counts), the output a view into a buffer of sentinels: a read outside the input poisons the result, a write
    outside the output shows (tests/tools/fuzz_guard.py runs this over random shapes)."""
    from pulsarbat_amd.transforms.dedispersion import _prepare
    G = 64
    rng = np.random.default_rng(n % 1013)
    x = (rng.standard_normal((n,) + tail) + 1j * rng.standard_normal((n,) + tail)).astype(dtype)
    buf = np.full((G + g0 + n + G,) + tail, np.nan + 1j * np.nan, dtype=dtype)
    buf[G + g0:G + g0 + n] = x
    view = pb.DeviceArray.from_host(buf)[G + g0:G + g0 + n]
    dm = 0.2 * min(1.0, n / 4096)
    z = pb.BasebandSignal(view, sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    yr, start, stop = orc.coherent_dedispersion(x, dm, 1e6, 1e9)
    plan, xin, s0, s1 = _prepare(z, pb.DM(dm), None, None, "auto")
    assert (s0, s1) == (start, stop)
    nout = stop - start
    obuf = pb.DeviceArray.from_host(np.full((G + nout + G,) + tail, 777.0 + 0j, dtype=dtype))
    plan.dedisperse(xin, out=obuf[G:G + nout])
    res = np.asarray(obuf)
    assert np.all(res[:G] == 777.0) and np.all(res[G + nout:] == 777.0)
    got = res[G:G + nout]
    assert np.all(np.isfinite(got))
    assert np.linalg.norm(got - yr) / np.linalg.norm(yr) < (5e-6 if dtype == np.complex64 else 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((1 << 16, 4, 2), np.complex64), ((100003, 3), np.complex64), ((4097, 5, 2), np.complex64),
                                         ((1 << 15, 2, 2), np.complex128), ((777, 7), np.complex128), ((1 << 12, 200), np.complex64)])
def test_layout_conversion(shape, dtype):
    """DeviceArray.to_series_major runs the pipeline's de-interleave kernel (pbh_relayout); values and strides."""
    from pulsarbat_amd import _hip
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)
    d = pb.DeviceArray.from_host(x)
    for align in (0, 5):
        s = d.to_series_major(align_start=align)
        assert s.series_major_pitch() is not None and tuple(s.shape) == shape
        assert np.array_equal(np.asarray(s), x)
        back = pb.DeviceArray.empty(shape, dtype)
        _hip.relayout(s, back)
        assert np.array_equal(np.asarray(back), x)
        s2 = pb.DeviceArray.empty_series_major(shape, dtype, align_start=3)
        _hip.relayout(s, s2)            # series-major -> series-major with another pitch / offset
        assert np.array_equal(np.asarray(s2), x)


@pytest.mark.parametrize("shape,dm,dtype", [((1 << 14, 256, 2), 20.0, np.complex64), ((4096, 1024, 2), 5.0, np.complex64),
                                            ((8192, 600), 10.0, np.complex64), ((2048, 512, 4), 2.0, np.complex64),
                                            ((4096, 256, 2), 5.0, np.complex128), ((1 << 14, 96, 1), 20.0, np.complex64)])
def test_one_tile_blocks_with_many_series(shape, dm, dtype):
    """Channelised blocks no longer than a tile (what stft with long segments hands to coherent_dedispersion): layout
    pass + planar row pass + layout pass; rows of a tile share chirp rows polarisation by polarisation."""
    sr, fc = 1e6 / 64, 1e9
    x = orc.synthetic_block(shape, 23).astype(dtype)
    z = make_signal(x, sr, fc, start_time=pb.Time(56000.0, format="mjd"))
    for zz in (z, z.to_device()):
        y = pb.coherent_dedispersion(zz, pb.DM(dm))
        yr, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
        assert y.shape == yr.shape and yr.shape[0] > 0 and y.dtype == dtype
        l2, mx = series_errors(y, yr)
        assert l2 < (RTOL_L2 if dtype == np.complex64 else 1e-8), f"relative L2 {l2:.2e}"   # (complex128: the chirp is complex64-rounded on both sides; last-bit flips of that rounding)


def test_channelised_block_beyond_2gib():
    """2^14 samples x 8192 channels x 2 pol (2 GiB): the shape contrib.stft(nperseg=1024) makes of the headline block."""
    import torch
    n, nchan, npol, sr, fc, dm = 1 << 14, 8192, 2, 50e6 / 1024, 1.4e9, 56.77
    g = torch.Generator(device="cuda").manual_seed(5)
    xt = torch.view_as_complex(torch.randn((n, nchan, npol, 2), generator=g, device="cuda", dtype=torch.float32))
    z = pb.DualPolarizationSignal(pb.DeviceArray(xt), sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear",
                                  freq_align="bottom")
    y = pb.coherent_dedispersion(z, pb.DM(dm))
    start, stop = orc.crop_bounds(dm, n, nchan, sr, fc, fc)
    assert y.shape == (stop - start, nchan, npol)
    import scipy.fft
    freqs = orc.channel_freqs(fc, sr, nchan, "bottom")
    for c in (0, 1, 4095, 8191):
        chirp = orc.transfer_function(dm, n, 1 / sr, freqs[c], fc)[:, None]
        xs = xt[:, c, :].cpu().numpy()
        ref = scipy.fft.ifft(scipy.fft.fft(xs, axis=0) * chirp, axis=0)[start:stop]
        got = y.data.tensor[:, c, :].cpu().numpy()
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < RTOL_L2


# ---- 7-smooth lengths: what the reference's fast_len / next_fast_len / prev_fast_len hand out --------------------------
# (reference pulsarbat/utils.py:68-130, transforms.py:364-382).  N = P * Q * 2^k runs the planar pipeline with mixed-radix
# column passes (k_colmix); pbh_plan_info reports the geometry, so a silent detour through the padded convolution fails.
SMOOTH = [
    (288, 9, 32),             # 2^5 3^2: Q = 9, rows of 32 points (512 rows per tile, 9 per series: ragged row tiles)
    (20000, 625, 32),         # 2^5 5^4
    (64800, 2025, 32),        # 2^5 3^4 5^2: P = 3, Q = 675
    (400000, 3125, 128),      # 2^7 5^5: P = 5, Q = 625
    (107520, 105, 1024),      # 2^10 3 5 7
    (294912, 18, 16384),      # 2^15 3^2: Q = 18 (radices 3, 3, 2)
    (1000000, 15625, 64),     # 10^6 = 2^6 5^6: P = 25, Q = 625
    (735 * 2048, 735, 2048),  # 2^11 3 5 7^2
    (27 << 16, 108, 16384),   # 2^16 3^3: N1 = 108 keeps a factor 4 (radices 3, 3, 3, 4)
]


@pytest.fixture
def mixed_all(monkeypatch):
    """PBH_MIXED is read once per process: the body runs in a child process with PBH_MIXED=2 (the default since the
    two-level route beats the convolution plan; the explicit setting keeps this test meaningful if the default moves)."""
    import os
    import subprocess
    import sys

    def run(code, **more):
        env = dict(os.environ, PBH_MIXED="2", **more)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout
    return run


def test_7smooth_two_level_lengths(mixed_all):
    """N1 = P * Q beyond one column tile (P > 1): both column roles of k_colmix, complex64 and complex128; the P-point stage of
    the first two lengths (P = 3, 5) runs the elementwise k_radix_p by default and the tile kernel with PBH_MIX_RADIXP=0."""
    code = """
import numpy as np, sys
sys.path.insert(0, "tests")
import test_gpu_parity as t
from pulsarbat_amd import _hip
for n, n1, n2 in [(64800, 2025, 32), (400000, 3125, 128), (1000000, 15625, 64)]:
    info = _hip.Plan(n, 1, 1, 0, n, device=0).info
    assert (info["n1"], info["n2"], info["nkernel"]) == (n1, n2, 7), info
    for tail in [(1,), (3, 2), (8, 2)]:
        for device in (False, True):
            t.check((n,) + tail, 3.0, 1e6, 1e9, seed=n % 97, device=device)
t.check128((400000, 2, 2), 3.0)
print("two-level ok")
"""
    assert "two-level ok" in mixed_all(code)
    assert "two-level ok" in mixed_all(code, PBH_MIX_RADIXP="0")


@pytest.mark.parametrize("n,n1,n2", SMOOTH)
@pytest.mark.parametrize("tail", [(1,), (3, 2), (8, 2)])
def test_7smooth_lengths(n, n1, n2, tail):
    from pulsarbat_amd import _hip
    info = _hip.Plan(n, 1, 1, 0, n, device=0).info
    assert (info["n1"], info["n2"]) == (n1, n2), info
    if n * int(np.prod(tail)) > 1 << 25:
        pytest.skip("oracle time")
    for device in (False, True):
        check((n,) + tail, 3.0, 1e6, 1e9, seed=n % 97, device=device)


def test_7smooth_detect_series_major_and_user_chirp():
    n, tail = 107520, (4, 2)
    x = orc.synthetic_block((n,) + tail, 5)
    z = make_signal(x, 1e6, 1e9).to_device()
    # fused detection tail
    d, s0 = pb.dedisperse_detect(z, pb.DM(4.0), mode="I", nscrunch=64)
    yr, start, stop = orc.coherent_dedispersion(x, 4.0, 1e6, 1e9)
    want = orc.scrunch(orc.to_stokes(yr, "linear")[:, :, 0], 64)
    assert s0 == start and np.asarray(d).shape == want.shape
    assert_detect_close(d, yr, "I", 64)
    zs0 = type(z).like(z, z.data.to_series_major())
    d2, _ = pb.dedisperse_detect(zs0, pb.DM(4.0), mode="I", nscrunch=64)
    assert_detect_close(d2, yr, "I", 64)
    # series-major arrays at both ends
    zs = type(z).like(z, z.data.to_series_major())
    ys = pb.coherent_dedispersion(zs, pb.DM(4.0))
    assert series_errors(np.asarray(ys), yr)[0] < RTOL_L2
    # a user chirp goes through the plan-order upload
    c = np.exp(2j * np.pi * np.random.default_rng(0).random((n, tail[0]))).astype(np.complex64)
    yc = pb.coherent_dedispersion(z, pb.DM(4.0), chirp=c)
    import scipy.fft
    ref = scipy.fft.ifft(scipy.fft.fft(x, axis=0) * c[:, :, None], axis=0)[start:stop]
    assert series_errors(np.asarray(yc), ref)[0] < RTOL_L2


@pytest.mark.parametrize("n", [20000, 400000])
def test_7smooth_c128(n):
    check128((n, 2, 2), 3.0)


def test_native_length_list_matches_the_plans():
    """Every length pulsarbat_amd.utils calls native gets a plan without the convolution detour (a real N1 x N2 split
    beyond one tile), and 7-smooth neighbours outside the list are not in it."""
    from pulsarbat_amd import _hip
    from pulsarbat_amd.utils import _native_lens
    lens = _native_lens(1 << 25)
    rng = np.random.default_rng(3)
    for n in sorted(set(rng.choice(lens, 60).tolist() + [96, 625 << 14, 600 << 14, 3 << 20])):
        info = _hip.Plan(int(n), 1, 1, 0, int(n), device=0).info
        assert info["n1"] * info["n2"] == n and info["n2"] <= 1 << 14, (n, info)   # (a convolution plan reports n1 = 1, n2 = n)
    for n in (10_000_000, 16_000_000, 3 ** 15, 1000):
        assert n not in lens


@pytest.mark.parametrize("n,tail,dtype", [
    (75 << 14, (2,), np.complex64), (81 << 12, (3,), np.complex64), (35 << 11, (2, 2), np.complex64),
    (625 << 10, (1,), np.complex64), (45 << 13, (2,), np.complex128),
    (2025 << 10, (2,), np.complex64), (3125 << 10, (1,), np.complex128),   # N1 = P x Q: both column levels
    (81000, (2,), np.complex64), (437400, (3,), np.complex64), (234375, (2,), np.complex64), (99225, (2,), np.complex128),
    (8505000, (1,), np.complex64),   # few factors of two / odd: mixed-radix rows, undone by the output pass
    (400000, (2,), np.complex64), (1000000, (3,), np.complex64),   # 2^7 5^5, 2^6 5^6: short power-of-two rows -> mixed-radix rows
])
def test_fft_7smooth_lengths(n, tail, dtype):
    """pb.fft.fft / ifft of 7-smooth lengths q * 2^k (q <= 1024, 2^k >= 1024): mixed-radix column pass + the engine's row
    transform, no convolution ring (same numbers either way; tools/bench_fft.py shows the rate)."""
    rng = np.random.default_rng(n % 1000 + len(tail))
    x = (rng.standard_normal((n,) + tail) + 1j * rng.standard_normal((n,) + tail)).astype(dtype)
    _fft_check(x, 2e-6 if dtype == np.complex64 else 1e-12)


# 7-smooth lengths with only three or four factors of two: no power-of-two rows, the rows are mixed-radix as well (k_rowmix)
ROWMIX = [81000,        # 2^3 3^4 5^3
          176400,       # 2^4 3^2 5^2 7^2
          437400,       # 2^3 3^7 5^2
          9000,         # 2^3 3^2 5^3: short everything
          8505000,      # 2^3 3^5 5^4 7: two column levels
          91125 * 4,    # 2^2 3^6 5^3: rows start at 32-byte offsets
          2 * 3 ** 5 * 5 ** 3 * 7,   # 2 3^5 5^3 7 = 425250: one factor of two
          3 * 5 ** 7,   # 234375, odd: planar rows at odd element offsets
          3 ** 4 * 5 ** 2 * 7 ** 2]  # 99225, odd


@pytest.mark.parametrize("n", ROWMIX)
@pytest.mark.parametrize("tail", [(1,), (3, 2), (4, 2)])
def test_7smooth_few_factors_of_two(n, tail):
    from pulsarbat_amd import _hip
    info = _hip.Plan(n, 1, 1, 0, n, device=0).info
    n1, n2 = info["n1"], info["n2"]
    # a real split with mixed-radix rows: N2 = 1..16 times an odd factor, not the (1, n) of a convolution plan
    assert n1 * n2 == n and n1 > 1 and n2 & (n2 - 1) and n2 <= 1024, info
    if n * int(np.prod(tail)) > 1 << 25:
        pytest.skip("oracle time")
    for device in (False, True):
        check((n,) + tail, 3.0, 1e6, 1e9, seed=n % 97, device=device)
    if tail == (3, 2):
        check128((n,) + tail, 3.0)


def test_7smooth_few_factors_user_chirp_and_shift():
    """user chirps go through the plan-order upload (rows in k_rowmix's digit-reversed order), time_shift through
    pbh_chirp_special"""
    n, tail = 81000, (3, 2)
    x = orc.synthetic_block((n,) + tail, 6)
    z = make_signal(x, 1e6, 1e9).to_device()
    c = np.exp(2j * np.pi * np.random.default_rng(1).random((n, tail[0]))).astype(np.complex64)
    yc = pb.coherent_dedispersion(z, pb.DM(4.0), chirp=c)
    _, start, stop = orc.coherent_dedispersion(x, 4.0, 1e6, 1e9)
    import scipy.fft
    ref = scipy.fft.ifft(scipy.fft.fft(x, axis=0) * c[:, :, None], axis=0)[start:stop]
    assert series_errors(np.asarray(yc), ref)[0] < RTOL_L2
    ch = pb.DM(4.0).chirp_from_signal(z)
    want = orc.chirp_from_signal(4.0, (n,) + tail, 1e6, 1e9)
    assert np.abs(np.asarray(ch) - want).max() < 2e-6
    ys = pb.time_shift(z, 2.5)
    assert series_errors(np.asarray(ys), orc.time_shift(x, 2.5)[0])[0] < RTOL_L2
    # odd length: bins and the band mask of freq_shift in natural order through the same plans
    n = 99225
    x = orc.synthetic_block((n,) + tail, 7)
    z = make_signal(x, 1e6, 1e9).to_device()
    assert series_errors(np.asarray(pb.time_shift(z, -3.25)), orc.time_shift(x, -3.25)[0])[0] < RTOL_L2
    ft = 0.1371
    assert series_errors(np.asarray(pb.freq_shift(z, ft * 1e6 * u.Hz)), orc.freq_shift(x, ft))[0] < 2e-5


@pytest.mark.parametrize("nchan,npol", [(9, 1), (7, 2)])
def test_single_stage_rows_tile_handout(nchan, npol):
    """7-smooth plan with 32-point rows (1 372 000 = 2^5 5^3 7^3): the phase row kernel's rows are ONE radix-32 stage with no
    LDS exchange, so nothing ordered thread 0's write of the next-tile slot against the other waves' read of it -- every
    (channel, row-group) pair beyond the first two per workgroup could be transformed twice or not at all, i.e. channels
    >= 6 here were garbage.  Found by tests/tools/fuzz_parity.py (seed 7) in round 3; present since round 2."""
    n, sr, fc, dm = 1372000, 25e6, 4e8, 1.0
    shape = (n, nchan, npol) if npol > 1 else (n, nchan)
    x = orc.synthetic_block(shape, 77)
    z = make_signal(x, sr, fc)
    y = pb.coherent_dedispersion(z.to_device(), pb.DM(dm))
    want, start, stop = orc.coherent_dedispersion(x, dm, sr, fc)
    got = np.asarray(y).reshape(stop - start, -1)
    ref = want.reshape(stop - start, -1)
    err = np.linalg.norm(got - ref, axis=0) / np.linalg.norm(ref, axis=0)
    assert err.max() < RTOL_L2, f"per-series relative L2 {err}"
