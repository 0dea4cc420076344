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
    """pbh_dedisperse_stream / _raw on (a) ordinary numpy memory -- pinned by the call for its duration -- and (b) a
    buffer the call CANNOT pin: a page in its middle is unmapped, so hipHostRegister of the range fails (and its ends are
    not pinned memory).  Both entry points must then return PBH_ERR_HIP ("page-locked"); there is no fall-back to
    asynchronous copies from pageable memory -- one would die on the hole.  (hipHostRegister of a range that overlaps an
    existing registration SUCCEEDS on this ROCm, so that cannot serve as the obstacle.)"""
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

    import mmap
    libc = C.CDLL(None, use_errno=True)
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]

    def holed(nbytes):
        """Anonymous mapping of >= nbytes with ONE PAGE UNMAPPED in its middle: it cannot be page-locked (and a copy that
        touched the hole would be a segmentation fault, not a wrong number -- the call has to fail before any copy)."""
        size = (nbytes + 3 * mmap.PAGESIZE) & ~(mmap.PAGESIZE - 1)
        mm = mmap.mmap(-1, size)
        addr = C.addressof(C.c_char.from_buffer(mm))
        hole = (addr + nbytes // 2) & ~(mmap.PAGESIZE - 1)
        return mm, addr, hole

    with _hip.Plan(chunk, 2, 2, start, stop, device=0) as plan:
        plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, orc.channel_freqs(fc, sr, 2), fc)
        good_in = x.copy()
        good_out = np.empty((nchunk * hop, 2, 2), np.complex64)
        keep = []
        for victim in ("in", "out"):
            nbytes = good_in.nbytes if victim == "in" else good_out.nbytes
            mm, addr, hole = holed(nbytes)
            arr = np.frombuffer(mm, dtype=np.complex64, count=nbytes // 8).reshape((-1, 2, 2))
            if victim == "in":
                arr[...] = good_in
            assert libc.munmap(hole, mmap.PAGESIZE) == 0
            keep.append((mm, arr))
            with pytest.raises(_hip.HipError, match="page-locked"):
                if victim == "in":
                    plan.dedisperse_stream(arr, out=good_out)
                else:
                    plan.dedisperse_stream(good_in, out=arr)
        # the raw entry point: same rule (8-bit complex payload, one block)
        raw = np.random.default_rng(1).integers(0, 256, shape[0] * 4 * 2, dtype=np.uint8)
        lay = dict(nbits=8, ncomp=2, code=0, blk_samples=shape[0], blk_stride=raw.size, hdr_bytes=0, elem0=0,
                   stride_t=4, stride_c=2, stride_p=1)
        mm, addr, hole = holed(raw.nbytes)
        bad_raw = np.frombuffer(mm, dtype=np.uint8, count=raw.nbytes)
        bad_raw[...] = raw
        assert libc.munmap(hole, mmap.PAGESIZE) == 0
        keep.append((mm, bad_raw))
        with pytest.raises(_hip.HipError, match="page-locked"):
            plan.dedisperse_stream_raw(bad_raw, lay, shape[0])
        # nothing was left registered or broken: both entry points work on ordinary buffers afterwards
        got, _ = plan.dedisperse_stream(good_in, out=good_out)
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-5
        plan.dedisperse_stream_raw(raw, lay, shape[0])
        del arr, bad_raw
