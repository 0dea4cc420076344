"""The four-pass schedule (csrc/fd4_kernels.hpp: forward column pass straight from the caller's sample-major block, Q4-order
intermediate, gang-scheduled row pass) against the oracle, and against the five-pass schedule it replaces (same
butterflies, same twiddles, same order of operations: only the addresses and the pacing of the memory instructions differ).

Reference expression: pulsarbat/transforms/dedispersion.py:125-133."""

import hashlib
import os

import numpy as np
import pytest

from oracle import dedisp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SR, FC = 1e6, 1e9


def _run(log2n, nchan, npol, dm, seed):
    """One dedispersion through the C ABI; returns (output, kernel names, crop)."""
    from pulsarbat_amd import _hip
    from pulsarbat_amd.device import DeviceArray
    n = 1 << log2n
    x = orc.synthetic_block((n, nchan, npol), seed)
    start, stop = orc.crop_bounds(dm, n, nchan, SR, FC, FC)
    freqs = FC + SR * (np.arange(nchan) + 0.5 - nchan / 2)
    with _hip.Plan(n, nchan, npol, start, stop) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / SR, freqs, FC)
        xd = DeviceArray.from_host(x)
        y = plan.dedisperse(xd)
        names = [k for k, _ in plan.profile(xd, y, iters=1)]
        y2 = plan.dedisperse(xd)          # profile() re-ran the kernels into y: take a fresh result
        got = np.asarray(y2)
    return x, got, names, (start, stop)


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


CASES = [(20, 8, 2, 30.0), (20, 2, 2, 10.0), (20, 4, 1, 30.0), (20, 6, 2, 50.0), (21, 4, 2, 30.0), (22, 8, 2, 60.0), (23, 2, 2, 100.0),
         # more than one 128-byte line of series per sample: gangs of 5, 6, 8 and 16 quads
         (20, 10, 2, 30.0), (20, 12, 2, 30.0), (20, 16, 2, 30.0), (20, 32, 2, 30.0), (21, 32, 1, 20.0)]


@pytest.mark.parametrize("log2n,nchan,npol,dm", CASES)
def test_four_pass_parity(log2n, nchan, npol, dm):
    if os.environ.get("PBH_FD4", "1") == "0":
        pytest.skip("PBH_FD4=0 in the environment")
    seed = 100 + log2n + nchan
    x, got, names, (start, stop) = _run(log2n, nchan, npol, dm, seed)
    assert names == ["k_col_fwd", "k_row_fused", "k_col_inv", "k_reinterleave"], names
    want, s0, s1 = orc.coherent_dedispersion(x, dm, SR, FC)
    assert (s0, s1) == (start, stop) and got.shape == want.shape
    g, w = got.reshape(len(got), -1), want.reshape(len(want), -1)
    err = np.linalg.norm(g - w, axis=0) / np.linalg.norm(w, axis=0)
    assert err.max() < 1e-5, err
    # the five-pass schedule on the same input (the library reads PBH_FD4 at every call)
    os.environ["PBH_FD4"] = "0"
    try:
        _, got5, names5, _ = _run(log2n, nchan, npol, dm, seed)
    finally:
        os.environ.pop("PBH_FD4", None)
    assert names5[0] == "k_deinterleave" and len(names5) == 5, names5
    # same butterflies in the same order; hipcc contracts multiply-adds differently in differently paced instantiations,
    # so the two agree to the last bits, not always bit for bit
    d = np.linalg.norm((got5 - got).reshape(len(got), -1), axis=0) / np.linalg.norm(w, axis=0)
    assert d.max() < 5e-7, d


def test_not_taken_where_the_geometry_does_not_fit():
    """odd series counts, series-major ends and long blocks keep the five-pass schedule."""
    _, got, names, _ = _run(20, 3, 1, 10.0, 5)          # S = 3
    assert names[0] == "k_deinterleave"
    _, got, names, _ = _run(20, 5, 2, 10.0, 5)          # S = 10
    assert names[0] == "k_deinterleave"


_SEARCH = r"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, {root!r})
from oracle import dedisp_oracle as orc
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
n, nchan, npol, dm, sr, fc = 1 << 23, 8, 2, 30.0, 1e6, 1e9
x = orc.synthetic_block((n, nchan, npol), 11)
start, stop = orc.crop_bounds(dm, n, nchan, sr, fc, fc)
freqs = fc + sr * (np.arange(nchan) + 0.5 - nchan / 2)
with _hip.Plan(n, nchan, npol, start, stop) as plan:
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, fc)
    xd = DeviceArray.from_host(x)
    y = np.asarray(plan.dedisperse(xd))
    y2 = np.asarray(plan.dedisperse(xd))
    assert np.array_equal(y, y2)
    cls = plan.buffer_class(xd)
    assert cls in (-1, 0, 1), cls
    assert np.array_equal(np.asarray(xd), x)          # the search only reads the caller's arrays
print("digest", hashlib.sha256(y.tobytes()).hexdigest())
"""


def test_second_work_buffer_search():
    """The first call of a plan with blocks of 1 GiB and more looks for a second work buffer of another allocation class than
    the first one, the input and the output (csrc/pbhip.hip: ensure_work2; DESIGN.md 6d d).  Placement is a speed matter only:
    the result is bit-identical with the search switched off, the caller's input is left as it was, the trace names the pick."""
    import subprocess
    import sys
    outs = {}
    for cls in ("1", "0"):
        env = dict(os.environ, PBH_CLASS=cls, PBH_TRACE_ALLOC="1")
        r = subprocess.run([sys.executable, "-c", _SEARCH.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[cls] = ([ln for ln in r.stdout.splitlines() if ln.startswith("digest")][0], r.stderr)
    assert outs["1"][0] == outs["0"][0]
    picked = [ln for ln in outs["1"][1].splitlines() if "work2: candidate" in ln]
    assert len(picked) == 1 and "no probing" not in picked[0], outs["1"][1][-1500:]
    timed = [ln for ln in outs["1"][1].splitlines() if "work2 candidate" in ln]
    assert 1 <= len(timed) <= 16 and all("from the input" in ln for ln in timed)
    assert "(no probing)" in [ln for ln in outs["0"][1].splitlines() if "work2: candidate" in ln][0]
