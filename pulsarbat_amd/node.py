"""Node-level pieces of the channel-sharded path: device buffers that the ranks of one node (one process
per GPU) share through the C ABI's ``pbh_node_*`` entry points, and the gather built on them.

The reference gathers chunked results in-process (``Signal.compute()``, pulsarbat/core.py:298-309).  With
one process per GPU the full-band ``(nout, nchan, npol)`` block lives on a destination rank; the other
ranks map it over xGMI and the LAST KERNEL of their dedispersion writes their channel slice into it
(``pbh_dedisperse_slice``) -- no transpose / all-gather / concatenate passes after the transform.
torch.distributed is used for what it is here for: exchanging the 64-byte handles and the closing barrier.
"""

import ctypes as C
import os

import numpy as np

from . import _hip
from .device import DeviceArray

__all__ = ["NodeBuffer", "PeerBuffer", "SharedBuffer", "SharedPeer", "ChannelGather", "GatherError", "MAX_NODE_BYTES"]

# The largest buffer a peer has been SEEN to map (tools/ipc_probe.py: 2040 MiB maps in 0.1 ms, 2056 MiB never returns from
# hipIpcOpenMemHandle).  pbh_node_alloc refuses anything larger; ChannelGather uses 1-GiB row-chunks.
MAX_NODE_BYTES = 2040 << 20


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
            if self.nbytes and t.data_ptr() != self.ptr:   # (an empty tensor has no storage to point anywhere)
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


class SharedBuffer:
    """A device buffer of ANY size that the node's other ranks can map (``pbh_node_share_alloc``: hipMemCreate + a POSIX file
    descriptor).  ``fd`` is the descriptor to hand to the peers (``send_fds``); ``close_fd()`` once they have it."""

    def __init__(self, shape, dtype, device):
        _hip._require_device()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self.nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 16)
        ptr, fd = C.c_void_p(), C.c_int(-1)
        _hip._check(_hip.lib().pbh_node_share_alloc(self.device, self.nbytes, C.byref(ptr), C.byref(fd)))
        self.ptr, self.fd = ptr.value, fd.value
        self._array = None

    def close_fd(self):
        fd, self.fd = getattr(self, "fd", -1), -1
        if fd >= 0:
            try:
                os.close(fd)
            except OSError:
                pass

    @property
    def array(self):
        if self._array is None:
            import torch
            t = torch.as_tensor(_Cai(self.ptr, self.shape, self.dtype, self), device=torch.device("cuda", self.device))
            if int(np.prod(self.shape)) and t.data_ptr() != self.ptr:
                raise _hip.HipError("torch copied the shared buffer instead of viewing it")
            self._array = DeviceArray(t)
        return self._array

    def close(self):
        self.close_fd()
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _hip.lib().pbh_node_share_free(self.device, C.c_void_p(ptr))
            except Exception:   # interpreter shutdown
                pass

    __del__ = close


class SharedPeer:
    """Another rank's :class:`SharedBuffer`, mapped from its file descriptor (``pbh_node_share_import``)."""

    def __init__(self, fd, nbytes, device):
        self.device = int(device)
        ptr = C.c_void_p()
        _hip._check(_hip.lib().pbh_node_share_import(self.device, int(fd), int(nbytes), C.byref(ptr)))
        self.ptr = ptr.value

    def close(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            try:
                _hip.lib().pbh_node_share_free(self.device, C.c_void_p(ptr))
            except Exception:
                pass

    __del__ = close


def _exchange_fds(group, my_fds, timeout=60.0):
    """Every rank that has file descriptors to share (``my_fds``, may be empty) serves them on a Unix socket; every rank
    fetches the descriptors of every other serving rank.  Returns ``{rank: [fd, ...]}`` (the caller owns and closes them).
    Descriptors cannot travel through torch.distributed: SCM_RIGHTS over a Unix-domain socket is the kernel's way."""
    import os
    import socket
    import tempfile
    import threading
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    path, server, private, setup_err = None, None, None, None
    if my_fds:
        # local, fallible work (a TMPDIR beyond sun_path's 108 bytes, an unwritable directory): a failure here must not
        # keep this rank out of the collective below -- it travels in it, and every rank raises after it
        try:
            if os.environ.get("PBH_TEST_FDS_BIND_FAIL") == str(rank):
                raise OSError("injected bind failure (PBH_TEST_FDS_BIND_FAIL)")
            private = tempfile.mkdtemp(prefix="pbh_gather_")   # mode 0700: only this user's processes can reach the socket
            path = os.path.join(private, "fds.sock")
            server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            server.bind(path)
            server.listen(world)
            server.settimeout(timeout)
        except Exception as exc:
            setup_err = f"rank {rank}: {exc!r}"
            if server is not None:
                server.close()
                server = None
            path = None
    paths = [None] * world
    dist.all_gather_object(paths, (path, len(my_fds), setup_err), group=group)
    bad = [e for _, _, e in paths if e]
    if bad:
        if private is not None:
            for undo, what in ((os.unlink, os.path.join(private, "fds.sock")), (os.rmdir, private)):
                try:
                    undo(what)
                except OSError:
                    pass
        raise _hip.HipError("descriptor exchange could not be set up: " + "; ".join(bad))
    paths = [(pth, nfd) for pth, nfd, _ in paths]
    errors = []

    def serve():
        try:
            for _ in range(world - 1):
                conn, _ = server.accept()
                with conn:
                    conn.settimeout(timeout)
                    conn.recv(1)
                    socket.send_fds(conn, [b"f"], list(my_fds))
        except Exception as exc:
            errors.append(exc)

    th = None
    if server is not None and world > 1:
        th = threading.Thread(target=serve, daemon=True)
        th.start()
    got = {}
    try:
        for r, (pth, nfd) in enumerate(paths):
            if r == rank or pth is None:
                continue
            with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as c:
                c.settimeout(timeout)
                c.connect(pth)
                c.send(b"r")
                _, fds, _, _ = socket.recv_fds(c, 16, nfd)
                if len(fds) != nfd:
                    raise _hip.HipError(f"rank {r} sent {len(fds)} descriptors, {nfd} expected")
                got[r] = list(fds)
    finally:
        if th is not None:
            th.join(timeout)
        if server is not None:
            server.close()
            for undo, what in ((os.unlink, path), (os.rmdir, private)):
                try:
                    undo(what)
                except OSError:
                    pass
    if errors:
        raise errors[0]
    return got


class GatherError(_hip.HipError):
    """A step of the collective gather failed on some rank; raised on EVERY rank of the group."""


class ChannelGather:
    """Gather of channel-sharded results by direct writes into the destination ranks' full-band blocks.

    ``mode="all"``: every rank ends up with the full ``(nout, nchan_total, npol)`` block (what ``gather=True`` returns
    on every rank); ``mode="root"``: only ``root`` does.  A destination's block is ONE contiguous buffer shared through a
    file descriptor (:class:`SharedBuffer`: hipMemCreate + SCM_RIGHTS; any size -- tools/micro/vmmprobe.hip shares a 3-GiB
    allocation between two processes).  Where that cannot be set up, all ranks fall back together to the older form: a run
    of ROW-CHUNKS, each its own hipMalloc of at most ``chunk_bytes`` (1 GiB) shared through hipIpc handles -- mapping an
    allocation larger than 2 GiB that way hangs in this ROCm stack's IPC (measured between two processes on ONE device:
    2040 MiB maps in 0.1 ms, 2056 MiB never returns; the cross-device case is unobserved) -- joined into one array
    afterwards.  Every rank runs its plan once: the pipeline's last kernel writes the rank's channel slice into the first
    destination's block (``pbh_dedisperse_slices``; its own when it is a destination), and ``pbh_place`` pushes the slice
    from there to the other destinations -- ONE STREAM PER DESTINATION, so that in ``mode="all"`` a rank drives its seven
    xGMI links at once (the links are point to point: a single stream keeps one of them busy at a time).

    Reusable for repeated calls of one geometry (a stream of blocks): chunks and mappings are set up once
    (``shard.coherent_dedispersion_sharded`` keeps its gathers in a cache next to the plans).  The chunks are DOUBLE
    BUFFERED: run ``n`` writes set ``n & 1``.  A peer can start the writes of run ``n + 1`` while a destination is still
    reading run ``n``'s chunks, but it cannot start those of run ``n + 2`` before the destination has passed the closing
    collective of run ``n + 1``, which it enters only after a device-wide synchronisation: no read-out is ever torn.

    Failure handling: every collective step carries a status (``_agree``); a failure on one rank (allocation, mapping,
    a plan that does not fit) raises :class:`GatherError` on ALL ranks after the same number of collectives, so nobody
    is left waiting in a barrier.
    """

    def __init__(self, nout, nchan_local, npol, dtype, device, group=None, mode="all", root=0, chunk_bytes=None):
        import torch
        import torch.distributed as dist
        if chunk_bytes is None:   # PBH_GATHER_CHUNK_BYTES: tests force the chunked form with many chunks at small sizes
            chunk_bytes = int(os.environ.get("PBH_GATHER_CHUNK_BYTES", 1 << 30))
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        self.group, self.mode, self.root = group, mode, int(root)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.nout, self.npol, self.dtype, self.device = int(nout), int(npol), np.dtype(dtype), int(device)
        self._coll_device = torch.device("cuda", self.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.own, self.peers, self._streams, self._runs, self._broken = None, {}, {}, 0, False
        counts = [None] * self.world
        dist.all_gather_object(counts, int(nchan_local), group=group)
        self.counts = [int(c) for c in counts]
        self.nchan_total = sum(self.counts)
        self.chan_lo = sum(self.counts[:self.rank])
        self.nchan_local = int(nchan_local)
        self.is_dest = mode == "all" or self.rank == self.root
        # Destination blocks: ONE contiguous buffer per destination and buffer set, shared through file descriptors
        # (SharedBuffer: no size limit, no joining copy).  If that cannot be set up on some rank -- all ranks then agree on
        # it -- the older form: row-chunks of <= 1 GiB shared through hipIpc handles (NodeBuffer) and joined afterwards.
        self.shared = os.environ.get("PBH_GATHER_SHARED", "1") != "0" and "PBH_GATHER_CHUNK_BYTES" not in os.environ
        if self.shared and not self._setup_shared():
            self.shared = False
        if not self.shared:
            self._setup_chunks(chunk_bytes)

    def _setup_shared(self):
        """The contiguous form.  Returns False (on every rank) when it is not available; raises nothing."""
        import torch.distributed as dist
        self.part_rows = [0, self.nout] if self.nout > 0 else [0]
        shape = (self.nout, self.nchan_total, self.npol)
        err, fds = None, []
        try:
            if self.is_dest and self.nout > 0:
                self.own = [[SharedBuffer(shape, self.dtype, self.device)] for _ in range(2)]
                fds = [bufs[0].fd for bufs in self.own]
            elif self.is_dest:
                self.own = [[], []]
        except Exception as exc:
            err = exc
        if self._agree(err is not None):
            self._release(free=True)
            return False
        try:
            got = _exchange_fds(self.group, fds) if self.world > 1 else {}
            nbytes = max(int(np.prod(shape)) * self.dtype.itemsize, 16)
            for r, rfds in got.items():
                try:
                    self.peers[r] = [[SharedPeer(fd, nbytes, self.device)] for fd in rfds]
                finally:
                    for fd in rfds:
                        os.close(fd)
        except Exception as exc:
            err = exc
        finally:
            for bufs in (self.own or []):
                for b in bufs:
                    b.close_fd()
        if self._agree(err is not None):
            self._release(free=False)
            self._agree(False)          # nobody still maps this rank's buffers when they are freed
            self._release(free=True)
            return False
        return True

    def _setup_chunks(self, chunk_bytes):
        import torch.distributed as dist
        group = self.group
        row_bytes = self.row_elems * self.dtype.itemsize
        rows_per = max(1, min(int(chunk_bytes), MAX_NODE_BYTES) // max(row_bytes, 1))
        self.part_rows = list(range(0, self.nout, rows_per)) + [self.nout] if self.nout > 0 else [0]
        nparts = len(self.part_rows) - 1
        # 1. destinations allocate both chunk sets and export them; the handles travel WITH the status
        err, mine = None, None
        try:
            if self.is_dest:
                self.own = [[NodeBuffer((self.part_rows[i + 1] - self.part_rows[i], self.nchan_total, self.npol),
                                        self.dtype, self.device) for i in range(nparts)] for _ in range(2)]
                mine = [[b.handle() for b in bufs] for bufs in self.own]
        except Exception as exc:
            err = f"rank {self.rank}: {exc!r}"
        got = [None] * self.world
        dist.all_gather_object(got, (err, mine), group=group)
        errs = [e for e, _ in got if e]
        # 2. everybody maps the destinations' chunks
        if not errs:
            try:
                for r, (_, hs) in enumerate(got):
                    if hs is not None and r != self.rank:
                        self.peers[r] = [[PeerBuffer(h, self.device) for h in hset] for hset in hs]
            except Exception as exc:
                err = f"rank {self.rank}: {exc!r}"
        failed = self._agree(bool(errs) or err is not None)
        if failed:   # agreed by all ranks: everybody unmaps, meets once more, then frees
            self.close()
            raise GatherError("gather set-up failed: " + "; ".join(errs + ([err] if err else []) or ["on another rank"]))

    @property
    def row_elems(self):
        return self.nchan_total * self.npol

    def _agree(self, failed):
        """Collective: True on every rank if ``failed`` was true on any.  One small all-reduce -- this IS the step's
        barrier, and it keeps the number of collectives per rank fixed on the error paths."""
        import torch
        import torch.distributed as dist
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=self._coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return bool(flag.item())

    def _stream_for(self, r):
        import torch
        if r not in self._streams:
            self._streams[r] = torch.cuda.Stream(device=self.device)
        return self._streams[r]

    def run(self, plan, x, copy=True):
        """Dedisperse this rank's shard ``x`` with ``plan`` and deliver the slice to every destination.
        Returns the full-band DeviceArray on destination ranks, None elsewhere.  Collective.

        ``copy=False`` returns the destination's own buffer (contiguous form: the full-band DeviceArray itself, no extra HBM
        pass at all; chunked form: the list of row-chunks, rows ``part_rows[i] : part_rows[i+1]``) instead of a copy; it is
        overwritten by the next-but-one ``run`` of this gather."""
        import torch
        lib = _hip.lib()
        gen = self._runs & 1
        err = None
        try:
            if plan.nchan != self.nchan_local or plan.npol != self.npol or plan.nout != self.nout:
                raise ValueError("plan geometry does not match the gather")
            off = self.chan_lo * self.npol
            ncol = self.nchan_local * self.npol
            dests = ([(self.rank, [b.ptr for b in self.own[gen]])] if self.own is not None else []) + \
                    [(r, [p.ptr for p in ps[gen]]) for r, ps in sorted(self.peers.items())]
            if self.nout > 0 and ncol > 0 and dests:
                first = dests[0][1]
                plan.dedisperse_slices(x, first, self.part_rows, self.row_elems, off)
                if len(dests) > 1:
                    esz = self.dtype.itemsize
                    main = torch.cuda.current_stream(self.device)
                    done = torch.cuda.Event()
                    done.record(main)
                    code = _hip._dtype_code(self.dtype)
                    for r, d in dests[1:]:
                        st = self._stream_for(r)
                        st.wait_event(done)
                        for i, (src, dst) in enumerate(zip(first, d)):
                            rows = self.part_rows[i + 1] - self.part_rows[i]
                            _hip._check(lib.pbh_place(self.device, C.c_void_p(st.cuda_stream), code,
                                                      C.c_void_p(src + off * esz), self.row_elems,
                                                      C.c_void_p(dst + off * esz), self.row_elems, rows, ncol))
            torch.cuda.synchronize(self.device)   # this rank's writes (local and peer, all streams) have landed
        except Exception as exc:
            err = exc
        self._runs += 1
        try:
            failed = self._agree(err is not None)      # ... and so have everybody else's
        except BaseException:
            self._broken = True                        # the ranks did not agree: close() must not enter a collective alone
            raise
        if failed:
            raise GatherError(f"gather run failed on rank {self.rank}: {err!r}" if err is not None
                              else "gather run failed on another rank") from err
        if self.own is None:
            return None
        chunks = [b.array for b in self.own[gen]]
        if self.shared and chunks:   # one contiguous buffer: nothing to join
            return chunks[0] if not copy else DeviceArray(chunks[0].tensor.clone())
        if not copy:
            return chunks
        full = DeviceArray.empty((self.nout, self.nchan_total, self.npol), self.dtype, device=self.device)
        for i, c in enumerate(chunks):
            full.tensor[self.part_rows[i]:self.part_rows[i + 1]].copy_(c.tensor)
        return full

    def _release(self, free):
        for sets in getattr(self, "peers", {}).values():
            for ps in sets:
                for p in ps:
                    p.close()
        self.peers = {}
        if free:
            for bufs in (getattr(self, "own", None) or []):
                for b in bufs:
                    b.close()
            self.own = None

    def close(self):
        """Collective: unmap the peers' buffers, agree that everybody has, free the own ones.  After a failure that the
        ranks did not agree on (``_broken``: the closing all-reduce of a run itself raised) the collective is skipped and the
        own buffers are NOT freed here -- a peer may still map them; they go when the SharedBuffer / NodeBuffer objects are
        collected (their ``__del__``) or the process exits: a late free, not a hang."""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        self._release(free=False)
        ok = not self._broken
        if ok:
            try:
                self._agree(False)   # nobody still has this rank's chunks mapped when they are freed
            except Exception:
                ok = False
        if ok:
            self._release(free=True)
