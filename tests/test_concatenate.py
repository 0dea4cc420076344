"""Host-side transforms next to the path: concatenate (defines the streaming driver's expected
result) and signal_transform.  Cases follow reference tests/test_transforms.py:44-215."""

import itertools

import numpy as np
import pytest

import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from pulsarbat_amd.time import Time


def same_signal(x, y):
    assert type(x) is type(y) and x.shape == y.shape
    assert np.array_equal(np.asarray(x.data), np.asarray(y.data))
    assert u.isclose(x.sample_rate, y.sample_rate)
    assert Time.isclose(x.start_time, y.start_time)


def same_radiosignal(x, y):
    same_signal(x, y)
    assert u.isclose(x.center_freq, y.center_freq) and u.isclose(x.chan_bw, y.chan_bw)
    assert x.freq_align == y.freq_align


def test_basic():
    rng = np.random.default_rng(0)
    z = pb.Signal(rng.standard_normal((16, 16)), sample_rate=1 * u.Hz, start_time=Time.now())
    x, y = z[:10], z[10:]
    for axis in [0, "time"]:
        same_signal(z, pb.concatenate([x, y], axis=axis))
        with pytest.raises(ValueError):
            pb.concatenate([y, x], axis=axis)
    y.sample_rate = x.sample_rate * 2
    with pytest.raises(ValueError):
        pb.concatenate([x, y], axis=0)
    x, y = z[:, :10], z[:, 10:]
    same_signal(z, pb.concatenate([x, y], axis=1))
    with pytest.raises(TypeError):
        pb.concatenate([x, y], axis="freq")
    with pytest.raises(ValueError):
        pb.concatenate([])
    with pytest.raises(TypeError):
        pb.concatenate([np.zeros(4)])

    z = pb.RadioSignal(rng.standard_normal((16, 16)), sample_rate=1 * u.Hz, start_time=Time.now(),
                       chan_bw=1 * u.MHz, center_freq=1 * u.GHz)
    x, y = z[:, :10], z[:, 10:]
    for axis in [1, "freq"]:
        same_radiosignal(z, pb.concatenate([x, y], axis=axis))
        with pytest.raises(ValueError):
            pb.concatenate([y, x], axis=axis)
    y.chan_bw = x.chan_bw * 2
    with pytest.raises(ValueError):
        pb.concatenate([x, y], axis=1)


def test_quadrants_signal():
    """A B / C D tiles of a plain Signal: valid along t = AB AD CB CD; along x: any pair with equal start."""
    z = pb.Signal(np.random.default_rng(1).standard_normal((16, 16)), sample_rate=1 * u.Hz, start_time=Time.now())
    A, B, C, D = z[:8, :8], z[8:, :8], z[:8, 8:], z[8:, 8:]
    same_signal(z, pb.concatenate([pb.concatenate([A, B], axis=0), pb.concatenate([C, D], axis=0)], axis=1))
    same_signal(z, pb.concatenate([pb.concatenate([A, C], axis=1), pb.concatenate([B, D], axis=1)], axis=0))
    time_ok = [(A, B), (A, D), (C, B), (C, D)]
    other_ok = [(A, A), (A, C), (B, B), (B, D), (C, C), (C, A), (D, D), (D, B)]
    for X, Y in itertools.product([A, B, C, D], repeat=2):
        if any(X is m and Y is n for m, n in time_ok):
            pb.concatenate([X, Y], axis=0)
        else:
            with pytest.raises(ValueError):
                pb.concatenate([X, Y], axis=0)
        if any(X is m and Y is n for m, n in other_ok):
            pb.concatenate([X, Y], axis=1)
        else:
            with pytest.raises(ValueError):
                pb.concatenate([X, Y], axis=1)


def test_quadrants_radiosignal():
    """With frequencies attached: valid along t = AB CD, along f = AC BD."""
    z = pb.RadioSignal(np.random.default_rng(2).standard_normal((16, 16)), sample_rate=1 * u.Hz,
                       start_time=Time.now(), chan_bw=1 * u.MHz, center_freq=1 * u.GHz)
    A, B, C, D = z[:8, :8], z[8:, :8], z[:8, 8:], z[8:, 8:]
    same_radiosignal(z, pb.concatenate([pb.concatenate([A, B], axis="time"), pb.concatenate([C, D], axis="time")],
                                       axis="freq"))
    same_radiosignal(z, pb.concatenate([pb.concatenate([A, C], axis="freq"), pb.concatenate([B, D], axis="freq")],
                                       axis="time"))
    for X, Y in itertools.product([A, B, C, D], repeat=2):
        if any(X is m and Y is n for m, n in [(A, B), (C, D)]):
            pb.concatenate([X, Y], axis=0)
        else:
            with pytest.raises(ValueError):
                pb.concatenate([X, Y], axis=0)
        if any(X is m and Y is n for m, n in [(A, C), (B, D)]):
            pb.concatenate([X, Y], axis=1)
        else:
            with pytest.raises(ValueError):
                pb.concatenate([X, Y], axis=1)


def test_signal_transform():
    z = pb.Signal(np.arange(12.0).reshape(6, 2), sample_rate=2 * u.Hz)

    @pb.signal_transform
    def scale(x, k=1.0):
        return x * k

    y = scale(z, k=3.0)
    assert type(y) is pb.Signal and np.array_equal(y.data, z.data * 3) and u.isclose(y.sample_rate, z.sample_rate)
    y = scale(z, k=2.0, signal_kwargs={"sample_rate": 4 * u.Hz})
    assert u.isclose(y.sample_rate, 4 * u.Hz)
    with pytest.raises(TypeError):
        scale(z, signal_type=int)
    assert scale.__name__ == "scale"


@pytest.mark.gpu
def test_stream_equals_concatenate_of_chunks():
    """BASELINE configs[3] semantics: the streaming driver's output is pb.concatenate of per-chunk calls."""
    from oracle import dedisp_oracle as orc
    n, chunk, dm = 1 << 17, 1 << 15, 4.0
    x = orc.synthetic_block((n, 2, 2), 3)
    z = pb.DualPolarizationSignal(x, sample_rate=1e6 * u.Hz, center_freq=1e9 * u.Hz, pol_type="linear",
                                  start_time=pb.Time(56000.0, format="mjd"))
    y, ms = pb.coherent_dedispersion_stream(z, pb.DM(dm), chunk=chunk)
    first = pb.coherent_dedispersion(z[:chunk], pb.DM(dm))
    hop = len(first)
    parts = [pb.coherent_dedispersion(z[k * hop:k * hop + chunk], pb.DM(dm)) for k in range((n - chunk) // hop + 1)]
    want = pb.concatenate(parts)
    assert y.shape == want.shape and np.allclose(np.asarray(y.data), np.asarray(want.data), atol=1e-6)
    assert Time.isclose(y.start_time, want.start_time)
