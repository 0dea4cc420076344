"""Host <-> device transfers of caller memory (pbh_transfer, the streaming entry points).

The mechanism behind round 1's intermittent "Memory access fault ... Write access to a read-only page" on a HOST
address: a large host-to-device copy from pageable memory makes the HIP runtime pin that range on the fly and cache
the pin; after the caller's allocator has recycled the addresses, a later device-to-host copy INTO them faults.  The
library never hands pageable caller memory to the runtime (pinned bounce buffers; the streaming entry points pin
explicitly or fail).  These tests drive that exact sequence once, deterministically -- they are not repeated runs.
"""

import ctypes as C

import numpy as np
import pytest

from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray

pytestmark = pytest.mark.gpu


def test_recycled_heap_range_h2d_then_d2h():
    """malloc -> large H2D from it -> free -> malloc again (glibc hands the same range back) -> large D2H into it."""
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    nbytes = 96 << 20    # well above the bounce threshold and glibc's mmap threshold alike is fine: brk or mmap
    same = 0
    dev = DeviceArray.empty((nbytes // 8,), np.complex64, device=0)
    for rnd in range(3):
        p1 = libc.malloc(nbytes)
        a = np.ctypeslib.as_array(C.cast(p1, C.POINTER(C.c_uint8)), shape=(nbytes,))
        a[:] = (np.arange(nbytes, dtype=np.uint32) * (rnd + 3) >> 3).astype(np.uint8)
        want = a.copy()
        _hip.transfer(0, dev.data_ptr(), p1, nbytes, to_host=False)
        del a
        libc.free(p1)
        p2 = libc.malloc(nbytes)          # the recycled range (same address in practice)
        same += int(p2 == p1)
        b = np.ctypeslib.as_array(C.cast(p2, C.POINTER(C.c_uint8)), shape=(nbytes,))
        b[:] = 0
        _hip.transfer(0, p2, dev.data_ptr(), nbytes, to_host=True)
        assert np.array_equal(b, want)
        del b
        libc.free(p2)
    assert same >= 1, "the allocator never handed the range back: the sequence under test did not happen"


def test_stream_entry_points_never_take_pageable_memory():
    """pbh_dedisperse_stream on (a) ordinary numpy memory -- pinned by the call for its duration -- and (b) input and
    output carved from ONE allocation so that they share a page: the call either pins both or fails with a HIP error;
    it never falls back to asynchronous copies from pageable memory."""
    import pulsarbat_amd as pb
    from pulsarbat_amd import units as u
    from oracle import dedisp_oracle as orc
    shape, dm, sr, fc, chunk = (1 << 17, 2, 2), 20.0, 1e6, 1e9, 1 << 15
    x = orc.synthetic_block(shape, 3)
    z = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=fc * u.Hz, pol_type="linear")
    y, _ = pb.coherent_dedispersion_stream(z, pb.DM(dm), chunk=chunk)
    first, start, stop = orc.coherent_dedispersion(x[:chunk], dm, sr, fc)
    hop = stop - start
    nchunk = (shape[0] - chunk) // hop + 1
    want = np.concatenate([orc.coherent_dedispersion(x[k * hop:k * hop + chunk], dm, sr, fc)[0] for k in range(nchunk)])
    assert np.linalg.norm(np.asarray(y) - want) / np.linalg.norm(want) < 1e-5
    # (b) adjacent input and output inside one buffer
    with _hip.Plan(chunk, 2, 2, start, stop, device=0) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, orc.channel_freqs(fc, sr, 2), fc)
        nin, nout = x.size, nchunk * hop * 4
        buf = np.empty(nin + nout, dtype=np.complex64)
        xin = buf[:nin].reshape(shape)
        xin[...] = x
        out = buf[nin:].reshape(nchunk * hop, 2, 2)
        try:
            got, _ = plan.dedisperse_stream(xin, out=out)
        except _hip.HipError as exc:
            assert "page-locked" in str(exc)
        else:
            assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-5
