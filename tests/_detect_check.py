"""Shared check of detected / scrunched output against the oracle in float64 ("Stokes/scrunch outputs
<= 1e-5 relative (all-positive sums)").

``want`` comes from the ORACLE's dedispersed voltages (``orc.coherent_dedispersion``) with the detection and the time sums in
float64 (``orc.detect_f64``): a float32 reference that adds 1024 powers one after the other is itself ~1e-5 off, which is
why rounds 1-3 accepted 3e-5 ... 2.4e-4.  Scale of the error: the Stokes-I (or |z|^2) sum of the same output sample -- Q, U
and V change sign, only I is an all-positive sum.  Below 16 samples per sum an output is (nearly) a single power: its error
is set by the voltage contract (max |dz| <= 4e-5 rms, tests/test_gpu_parity.py RTOL_MAX), not by summation, and the scale is
the series' MEAN power times the scrunch factor with 4e-5 as the bound."""

import numpy as np

from oracle import dedisp_oracle as orc

DETECT_RTOL = 1e-5
DETECT_RTOL_SINGLE = 4e-5


def detect_errors(got, yr, mode, nscrunch, pol_type="linear"):
    """(max error, bound): ``got`` detected output of the device path, ``yr`` the oracle's dedispersed voltages
    (nout, nchan[, npol]), ``mode`` in {"intensity", "I", "linear", "circular"}."""
    yr = np.asarray(yr)
    if mode == "intensity":
        want, scale = orc.detect_f64(yr, "intensity", nscrunch)
    elif mode == "I":
        want, scale = orc.detect_f64(yr, "I", nscrunch, pol_type)
    else:
        want, scale = orc.detect_f64(yr, "stokes", nscrunch, mode)
    got = np.asarray(got).astype(np.float64)
    assert got.shape == want.shape or got.size == want.size, (got.shape, want.shape)
    got = got.reshape(want.shape)
    if nscrunch >= 16:
        err = np.abs(got - want) / scale
        return float(err.max()), DETECT_RTOL
    mean = scale.mean(axis=0, keepdims=True)            # per channel (and series): mean power x nscrunch
    err = np.abs(got - want) / mean
    return float(err.max()), DETECT_RTOL_SINGLE


def assert_detect_close(got, yr, mode, nscrunch, pol_type="linear", what=""):
    err, bound = detect_errors(got, yr, mode, nscrunch, pol_type)
    assert err <= bound, f"{what} mode {mode} x{nscrunch}: max error {err:.2e} of the float64 sums (bound {bound:.0e})"
    return err
