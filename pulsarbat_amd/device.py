"""HBM-resident array that can sit in ``Signal.data``.

The reference's ``Signal`` touches only ``ndim/shape/dtype/astype/__len__/
__getitem__`` of its data plus ``numpy.asanyarray`` for host copies
(pulsarbat/core.py:59-97, 149-153, 207-230): the same seam its dask arrays use.
``DeviceArray`` implements that protocol over a torch ROCm tensor (torch is
used for device memory and streams only).  Keeping a signal's data as a
DeviceArray is this build's ``persist()``; ``numpy.asarray(x)`` / ``x.get()`` is
its ``compute()``.
"""

import numpy as np

__all__ = ["DeviceArray"]

_NP2T = None


def _maps():
    global _NP2T
    if _NP2T is None:
        import torch
        _NP2T = {np.dtype(np.complex64): torch.complex64, np.dtype(np.float32): torch.float32,
                 np.dtype(np.complex128): torch.complex128, np.dtype(np.float64): torch.float64}
    return _NP2T


class DeviceArray:
    __array_priority__ = 100

    def __init__(self, tensor):
        import torch
        if not isinstance(tensor, torch.Tensor) or not tensor.is_cuda:
            raise TypeError("DeviceArray wraps a torch tensor living on a HIP device")
        self._t = tensor

    # --- construction -------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype, device=None):
        import torch
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        return cls(torch.empty(tuple(int(s) for s in shape), dtype=_maps()[np.dtype(dtype)], device=dev))

    @classmethod
    def empty_series_major(cls, shape, dtype, device=None, align_start=0):
        """Uninitialised (n, ...) array stored series-major ("planar"): time is the fastest axis,
        element (t, i, j, ...) lives at ``off + series*pitch + t``; strides ``(1, ..., pitch)``.

        This is the layout the column passes work in, so a device-resident pipeline that keeps its
        arrays this way skips both layout passes of ``coherent_dedispersion`` (C ABI:
        ``pbh_dedisperse_layout``).  ``pitch`` is padded to a multiple of 16 elements and the storage
        offset is ``align_start % 16`` so that sample ``align_start`` of the PRODUCER's time axis
        falls on a 128-byte line (full-line stores for a cropped output)."""
        import torch
        shape = tuple(int(s) for s in shape)
        n, rest = shape[0], shape[1:]
        nser = int(np.prod(rest)) if rest else 1
        off = int(align_start) % 16
        pitch = -(-(n + off) // 16) * 16
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        flat = torch.empty(nser * pitch + 16, dtype=_maps()[np.dtype(dtype)], device=dev)
        strides = [1]
        acc = pitch
        for d in reversed(rest):
            strides.insert(1, acc)
            acc *= d
        return cls(flat.as_strided(shape, tuple(strides), off))

    def series_major_pitch(self):
        """Pitch (elements between consecutive series) if this array is stored series-major, else None."""
        t = self._t
        if t.dim() < 2 or t.shape[0] < 1:
            return None
        st = t.stride()
        if t.shape[0] > 1 and st[0] != 1:
            return None
        pitch = st[-1]
        if pitch < t.shape[0]:
            return None
        acc = pitch
        for d in range(t.dim() - 1, 0, -1):
            if t.shape[d] > 1 and st[d] != acc:
                return None
            acc *= t.shape[d]
        return int(pitch)

    def to_series_major(self, align_start=0):
        """Copy into series-major storage (one transposing copy); a no-op if already stored that way."""
        if self.series_major_pitch() is not None and not self._t.is_contiguous():
            return self
        out = DeviceArray.empty_series_major(self.shape, self.dtype, device=self.device_index, align_start=align_start)
        if self._t.is_contiguous() and self._t.is_complex() and self._t.numel() > 0 and out.series_major_pitch() is not None:
            from . import _hip
            return _hip.relayout(self, out)   # the pipeline's de-interleave kernel (a strided torch copy is ~9x slower)
        out._t.copy_(self._t)
        return out

    def raw_ptr(self):
        """Device pointer of element [0, 0, ...] whatever the strides."""
        return self._t.data_ptr()

    @classmethod
    def from_host(cls, a, device=None):
        import torch
        if isinstance(a, DeviceArray):
            return a
        a = np.ascontiguousarray(a)
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        # every dtype goes through the library's pinned bounce buffers (pbh_transfer), not torch's copy: pageable numpy
        # memory handed to the HIP runtime gets pinned on the fly and the pin is cached beyond the array's life
        # (DESIGN.md 6).  The torch dtype is whatever torch maps the numpy dtype to (TypeError if it has none).
        tdt = _maps().get(np.dtype(a.dtype)) or torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        t = torch.empty(a.shape, dtype=tdt, device=dev)
        if a.size:
            from . import _hip
            _hip.transfer(dev.index, t.data_ptr(), a.ctypes.data, a.nbytes, to_host=False)
        return cls(t)

    # --- array protocol used by Signal ---------------------------------------------------
    @property
    def tensor(self):
        return self._t

    @property
    def device_index(self):
        return self._t.device.index

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def dtype(self):
        for npd, td in _maps().items():
            if td == self._t.dtype:
                return npd
        raise TypeError(f"unsupported tensor dtype {self._t.dtype}")

    def _np_dtype(self):
        """numpy dtype of the tensor, including the ones the Signal protocol does not use (integer payloads, masks)."""
        try:
            return self.dtype
        except TypeError:
            import torch
            return torch.empty(0, dtype=self._t.dtype).numpy().dtype

    @property
    def size(self):
        return self._t.numel()

    @property
    def nbytes(self):
        return self._t.numel() * self._t.element_size()

    def __len__(self):
        return self._t.shape[0]

    def __getitem__(self, index):
        return DeviceArray(self._t[index])

    def astype(self, dtype, casting="unsafe", copy=True):
        dtype = np.dtype(dtype)
        if not np.can_cast(self.dtype, dtype, casting=casting):
            raise TypeError(f"Cannot cast array data from {self.dtype} to {dtype} according to the rule '{casting}'")
        if dtype == self.dtype and not copy:
            return self
        return DeviceArray(self._t.to(_maps()[dtype]))

    def data_ptr(self):
        """Device pointer of a C-contiguous view (what the C ABI takes)."""
        if not self._t.is_contiguous():
            raise ValueError("the C ABI needs a C-contiguous device array; call .contiguous() first")
        return self._t.data_ptr()

    def contiguous(self):
        if self._t.is_contiguous():
            return self
        if self._t.is_complex() and self._t.numel() > 0 and self.series_major_pitch() is not None:
            from . import _hip   # series-major -> C order with the pipeline's re-interleave kernel
            return _hip.relayout(self, DeviceArray.empty(self.shape, self.dtype, device=self.device_index))
        return DeviceArray(self._t.contiguous())

    def get(self):
        """Host copy as numpy."""
        t = self.contiguous()._t
        if t.numel() == 0:
            return np.empty(tuple(t.shape), dtype=self._np_dtype())
        from . import _hip
        out = np.empty(tuple(t.shape), dtype=self._np_dtype())
        _hip.transfer(t.device.index, out.ctypes.data, t.data_ptr(), out.nbytes, to_host=True)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __repr__(self):
        return f"DeviceArray<shape={self.shape}, dtype={self.dtype}, device={self._t.device}>"
