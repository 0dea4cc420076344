"""Node-level pieces of the channel-sharded path: device buffers that the ranks of one node (one process
per GPU) share through the C ABI's ``pbh_node_*`` entry points, and the gather built on them.

The reference gathers chunked results in-process (``Signal.compute()``, pulsarbat/core.py:298-309).  With
one process per GPU the full-band ``(nout, nchan, npol)`` block lives on a destination rank; the other
ranks map it over xGMI and the LAST KERNEL of their dedispersion writes their channel slice into it
(``pbh_dedisperse_slice``) -- no transpose / all-gather / concatenate passes after the transform.
torch.distributed is used for what it is here for: exchanging the 64-byte handles and the closing barrier.
"""

import ctypes as C

import numpy as np

from . import _hip
from .device import DeviceArray

__all__ = ["NodeBuffer", "PeerBuffer", "ChannelGather"]


class _Cai:
    """Minimal ``__cuda_array_interface__`` carrier so torch can view library-owned device memory."""

    def __init__(self, ptr, shape, dtype, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": np.dtype(dtype).str,
                                         "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner   # keeps the allocation alive as long as any tensor view exists


class NodeBuffer:
    """A whole device allocation owned by this rank (``pbh_node_alloc``): exportable to the node's other ranks."""

    def __init__(self, shape, dtype, device):
        _hip._require_device()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        ptr = C.c_void_p()
        _hip._check(_hip.lib().pbh_node_alloc(self.device, max(self.nbytes, 16), C.byref(ptr)))
        self.ptr = ptr.value
        self._array = None

    def handle(self):
        """The 64-byte handle a peer passes to :class:`PeerBuffer`."""
        h = (C.c_ubyte * 64)()
        _hip._check(_hip.lib().pbh_node_export(self.device, C.c_void_p(self.ptr), C.cast(h, C.c_void_p)))
        return bytes(h)

    @property
    def array(self):
        """DeviceArray view of the buffer (the buffer lives as long as the view does)."""
        if self._array is None:
            import torch
            t = torch.as_tensor(_Cai(self.ptr, self.shape, self.dtype, self), device=torch.device("cuda", self.device))
            if t.data_ptr() != self.ptr:
                raise _hip.HipError("torch copied the node buffer instead of viewing it")
            self._array = DeviceArray(t)
        return self._array

    def close(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _hip.lib().pbh_node_free(self.device, C.c_void_p(ptr))
            except Exception:   # interpreter shutdown
                pass

    __del__ = close


class PeerBuffer:
    """Another rank's :class:`NodeBuffer`, mapped into this process (``pbh_node_import``)."""

    def __init__(self, handle, device):
        if len(handle) != 64:
            raise ValueError("a node handle is 64 bytes")
        self.device = int(device)
        h = (C.c_ubyte * 64).from_buffer_copy(handle)
        ptr = C.c_void_p()
        _hip._check(_hip.lib().pbh_node_import(self.device, C.cast(h, C.c_void_p), C.byref(ptr)))
        self.ptr = ptr.value

    def close(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _hip.lib().pbh_node_release(self.device, C.c_void_p(ptr))
            except Exception:
                pass

    __del__ = close


class ChannelGather:
    """Gather of channel-sharded results by direct writes into the destination ranks' full-band blocks.

    ``mode="all"``: every rank ends up with the full ``(nout, nchan_total, npol)`` block (what ``gather=True`` returns
    on every rank); ``mode="root"``: only ``root`` does.  A destination's block is a run of ROW-CHUNKS, each its own
    allocation of at most ``chunk_bytes`` (1 GiB): mapping an allocation larger than 2 GiB into a peer process hangs in
    this ROCm stack's IPC (measured: 2040 MiB maps in 0.1 ms, 2056 MiB never returns), and the blocks in question are
    2 - 17 GB.  Every rank runs its plan once: the pipeline's last kernel writes the rank's channel slice into the chunks
    of the first destination (``pbh_dedisperse_slices``; its own chunks when it is a destination), and ``pbh_place``
    pushes the slice from there to the other destinations' chunks.  After the closing barrier a destination joins its
    chunks into one contiguous array (a local pass at HBM speed, small beside the xGMI transfer it follows).
    Reusable for repeated calls of one geometry (a stream of blocks): chunks and mappings are set up once.
    """

    def __init__(self, nout, nchan_local, npol, dtype, device, group=None, mode="all", root=0, chunk_bytes=None):
        import os
        import torch.distributed as dist
        if chunk_bytes is None:   # PBH_GATHER_CHUNK_BYTES: tests force many chunks at small sizes
            chunk_bytes = int(os.environ.get("PBH_GATHER_CHUNK_BYTES", 1 << 30))
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        self.group, self.mode, self.root = group, mode, int(root)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.nout, self.npol, self.dtype, self.device = int(nout), int(npol), np.dtype(dtype), int(device)
        counts = [None] * self.world
        dist.all_gather_object(counts, int(nchan_local), group=group)
        self.counts = [int(c) for c in counts]
        self.nchan_total = sum(self.counts)
        self.chan_lo = sum(self.counts[:self.rank])
        self.nchan_local = int(nchan_local)
        self.is_dest = mode == "all" or self.rank == self.root
        row_bytes = self.row_elems * self.dtype.itemsize
        rows_per = max(1, min(int(chunk_bytes), (2 << 30) - 1) // max(row_bytes, 1))
        self.part_rows = list(range(0, self.nout, rows_per)) + [self.nout] if self.nout > 0 else [0, 0]
        nparts = len(self.part_rows) - 1
        self.own = None
        if self.is_dest:
            self.own = [NodeBuffer((self.part_rows[i + 1] - self.part_rows[i], self.nchan_total, self.npol), self.dtype,
                                   self.device) for i in range(nparts)]
        handles = [None] * self.world
        dist.all_gather_object(handles, [b.handle() for b in self.own] if self.own is not None else None, group=group)
        self.peers = {}
        for r, hs in enumerate(handles):
            if hs is not None and r != self.rank:
                self.peers[r] = [PeerBuffer(h, self.device) for h in hs]

    @property
    def row_elems(self):
        return self.nchan_total * self.npol

    def run(self, plan, x):
        """Dedisperse this rank's shard ``x`` with ``plan`` and deliver the slice to every destination.
        Returns the full-band DeviceArray on destination ranks, None elsewhere.  Collective: ends with a barrier."""
        import torch
        import torch.distributed as dist
        lib = _hip.lib()
        if plan.nchan != self.nchan_local or plan.npol != self.npol or plan.nout != self.nout:
            raise ValueError("plan geometry does not match the gather")
        off = self.chan_lo * self.npol
        ncol = self.nchan_local * self.npol
        dests = ([[b.ptr for b in self.own]] if self.own is not None else []) + \
                [[p.ptr for p in ps] for _, ps in sorted(self.peers.items())]
        if self.nout > 0 and ncol > 0:
            plan.dedisperse_slices(x, dests[0], self.part_rows, self.row_elems, off)
            esz = self.dtype.itemsize
            stream = _hip._stream_ptr(self.device)
            for d in dests[1:]:
                for i, (src, dst) in enumerate(zip(dests[0], d)):
                    rows = self.part_rows[i + 1] - self.part_rows[i]
                    _hip._check(lib.pbh_place(self.device, stream, _hip._dtype_code(self.dtype),
                                              C.c_void_p(src + off * esz), self.row_elems,
                                              C.c_void_p(dst + off * esz), self.row_elems, rows, ncol))
        torch.cuda.synchronize(self.device)   # this rank's writes (local and peer) have landed
        dist.barrier(group=self.group)        # ... and so have everybody else's
        if self.own is None:
            return None
        if len(self.own) == 1:
            return DeviceArray(self.own[0].array.tensor.clone())   # (a copy: the chunks are re-used by the next run)
        full = DeviceArray.empty((self.nout, self.nchan_total, self.npol), self.dtype, device=self.device)
        for i, b in enumerate(self.own):
            full.tensor[self.part_rows[i]:self.part_rows[i + 1]].copy_(b.array.tensor)
        return full

    def close(self):
        import torch.distributed as dist
        for ps in getattr(self, "peers", {}).values():
            for p in ps:
                p.close()
        self.peers = {}
        try:
            dist.barrier(group=self.group)   # nobody still has this rank's chunks mapped when they may be freed
        except Exception:
            pass
        for b in (getattr(self, "own", None) or []):
            b.close()
        self.own = None
