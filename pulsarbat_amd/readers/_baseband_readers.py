"""Baseband-file readers (reference pulsarbat/readers/_baseband_readers.py) with the payload unpacked on the
GPU: ``read`` returns a signal whose data is a ``DeviceArray``.

The reference reads through ``baseband`` into float32 / complex64 numpy arrays and post-processes on the host
(conjugate lower-sideband series :147-151, transpose to (time, chan, pol) :223-226, :268-275); here the file
bytes of the needed blocks go to the device as they are and one kernel (``pbh_decode``) unpacks, conjugates and
lays the result out -- optionally series-major, the layout ``coherent_dedispersion`` runs fastest on.
Real-sampled data continue into ``real_to_complex`` on the device (:139-141).
"""

import operator

import numpy as np

from .. import units as u
from .. import _hip
from ..device import DeviceArray
from ..core import Signal, BasebandSignal, IntensitySignal, DualPolarizationSignal, FullStokesSignal
from ..utils import real_to_complex
from ._base import BaseReader
from ._formats import open_raw

__all__ = ["BasebandReader", "GUPPIRawReader", "DADAStokesReader"]


class BasebandReader(BaseReader):
    """Reader of a DADA / GUPPI raw / VDIF file (``name``; a sequence of consecutive files for GUPPI)
    (_baseband_readers.py:19-165).

    ``signal_type`` / ``signal_kwargs`` / ``intensity`` / ``lower_sideband`` as in the reference;
    ``format`` overrides detection by content; ``squeeze`` (default True) drops unit sample axes;
    ``series_major`` makes ``read`` return time-fastest device arrays; ``device`` picks the GPU.
    """

    def __init__(self, name, /, *, signal_type=Signal, signal_kwargs=dict(), intensity=None, lower_sideband=False,
                 format=None, squeeze=True, series_major=False, device=None):
        self._name = name
        self._series_major = bool(series_major)
        self._device = device
        if intensity is None:
            self._intensity = issubclass(signal_type, IntensitySignal)
        else:
            self._intensity = bool(intensity)
            if issubclass(signal_type, BasebandSignal) and self.intensity:
                raise ValueError("intensity must be False when using pb.BasebandSignal")
            if issubclass(signal_type, IntensitySignal) and not self.intensity:
                raise ValueError("intensity must be True when using pb.IntensitySignal")
        raw = self._raw = open_raw(name, format)
        self._complex_data = bool(raw.complex_data)
        if self.intensity and self.complex_data:
            raise ValueError("Intensity data cannot be complex-valued!")
        shape1 = tuple(raw.sample_shape)
        self._squeeze = bool(squeeze)
        self._in_sample_shape = tuple(d for d in shape1 if d != 1) if squeeze else shape1
        if self.real_baseband:
            rate, length, dtype = (raw.sample_rate / 2).to(u.MHz), raw.nsample // 2, np.complex64
        else:
            rate, length = raw.sample_rate.to(u.MHz), raw.nsample
            dtype = np.complex64 if self.complex_data else np.float32
        self.lower_sideband = lower_sideband
        self._dtype = np.dtype(dtype)
        shape = (length,) + self._read_array(0, 0).shape[1:]
        super().__init__(shape=shape, dtype=dtype, signal_type=signal_type, sample_rate=rate,
                         start_time=raw.start_time, **signal_kwargs)

    # ---- description -----------------------------------------------------------------------------------
    complex_data = property(lambda self: self._complex_data, doc="Whether the stored samples are complex.")
    intensity = property(lambda self: self._intensity, doc="Whether the data are intensities, not voltages.")

    @property
    def real_baseband(self):
        return not (self.intensity or self.complex_data)

    @property
    def lower_sideband(self):
        return self._lower_sideband

    @lower_sideband.setter
    def lower_sideband(self, s):
        if type(s) is not bool:
            s = np.array(s).astype(bool)
            if s.shape != self._in_sample_shape:
                raise ValueError(f"Invalid lower_sideband shape. Got {s.shape}, expected {self._in_sample_shape}")
        self._lower_sideband = s

    # ---- reading ---------------------------------------------------------------------------------------
    def _axes(self):
        """Strides of the two output sample axes (axis1, axis2) in payload elements, and their extents.
        Subclasses reorder / flip axes here instead of transposing the decoded array."""
        raw = self._raw
        return raw.sample_shape, (raw.stride_1, raw.stride_2), raw.elem0

    def _conj_mask(self, shape1):
        lsb = self.lower_sideband
        if self.intensity or lsb is False:
            return None
        if lsb is True:
            return np.ones(shape1, bool)
        return np.asarray(lsb, bool).reshape(shape1)

    def _subset_layout(self, channels=None):
        """``(shape1, layout, byte_range, conj_mask)`` of a read / stream of ``channels`` (a slice of axis1; None = all): the
        subset is expressed in the decode layout (``elem0``, channel count), and where its bytes are a small contiguous part
        of every block's payload (channel-major blocks: GUPPI raw) ``byte_range`` is that part and the layout describes the
        PACKED blocks ``RawStream.fetch(byte_range=...)`` returns -- only those bytes are read from the file and cross PCIe."""
        raw = self._raw
        shape1, strides, elem0 = self._axes()
        conj = self._conj_mask(shape1) if raw.complex_data else None
        if channels is not None:
            lo, hi, _ = channels.indices(shape1[0])
            shape1 = (max(hi - lo, 0), shape1[1])
            elem0 = elem0 + lo * strides[0]
            conj = None if conj is None else conj[lo:hi]
        lay = raw.layout()
        lay.update(elem0=elem0, stride_c=strides[0], stride_p=strides[1])
        byte_range = None
        if channels is not None and raw.gather is None and shape1[0] > 0:
            # element indices a block's samples of these channels reach -> the payload bytes that hold them
            ends = [elem0 + sum(st * e for st, e in zip((raw.stride_t, strides[0], strides[1]), c))
                    for c in ((a, b, d) for a in (0, raw.blk_samples - 1) for b in (0, shape1[0] - 1) for d in (0, shape1[1] - 1))]
            bits = lay["nbits"] * lay["ncomp"]
            plo, phi = (min(ends) * bits // 8) // 16 * 16, -(-(max(ends) + 1) * bits // 8)
            if 0 <= plo < phi <= raw.blk_bytes and (phi - plo) * 4 <= raw.blk_bytes * 3:
                byte_range = (plo, phi)
                lay.update(blk_stride=-(-(phi - plo) // 16) * 16, elem0=elem0 - plo * 8 // bits)
        return shape1, lay, byte_range, conj

    def _decode(self, offset, n, ncomp_real_factor=1, channels=None):
        """Device array (n * factor, axis1, axis2) of the stored samples from ``offset * factor`` (``channels``: a slice of
        axis1, see ``_subset_layout``)."""
        raw = self._raw
        shape1, lay, byte_range, conj = self._subset_layout(channels)
        if n == 0 or shape1[0] == 0:
            return np.empty((n * ncomp_real_factor,) + tuple(shape1), np.complex64 if raw.complex_data else np.float32)
        buf, first = raw.fetch(offset * ncomp_real_factor, n * ncomp_real_factor, byte_range=byte_range)
        return _hip.decode(buf, lay, first, n * ncomp_real_factor, shape1[0], shape1[1], conj=conj, scale=raw.scale,
                           series_major=self._series_major and raw.complex_data, device=self._device)

    def _finish(self, z):
        """Drop unit sample axes when squeezing."""
        if not self._squeeze:
            return z
        keep = (z.shape[0],) + tuple(d for d in z.shape[1:] if d != 1)
        if keep == tuple(z.shape):
            return z
        if isinstance(z, np.ndarray):
            return z.reshape(keep)
        return type(z)(z.tensor.reshape(keep) if z.tensor.is_contiguous() else z.tensor.squeeze())

    def _read_baseband(self, offset, n, /, **kwargs):
        """n samples from ``offset``: real data are read at twice the rate and converted
        (_baseband_readers.py:136-153)."""
        channels = kwargs.get("channels")
        if self.real_baseband:
            z = self._decode(offset, n, 2, channels=channels)
            if n and z.shape[1]:
                z = real_to_complex(z, axis=0)
                mask = self._conj_mask(self._axes()[0])
                if mask is not None and channels is not None:
                    mask = mask[channels]
                if mask is not None and mask.any():
                    z = type(z)(_conj_where(z.tensor, mask))
            else:
                z = z.astype(np.complex64)[:n]
        else:
            z = self._decode(offset, n, channels=channels)
        return z

    def read(self, offset, n, /, channels=None, **kwargs):
        """``n`` samples from ``offset`` (reference readers/_base.py:298-333).  ``channels`` (a slice of the channel axis,
        e.g. ``shard.channel_slice(nchan, world, rank)``): the signal of those channels only, equal to ``read(offset,
        n)[:, channels]`` -- what a rank of a channel-sharded job needs -- without reading, uploading or decoding the
        other channels where the file's blocks are channel-major (GUPPI raw)."""
        if channels is None:
            return super().read(offset, n, **kwargs)
        if not isinstance(channels, slice) or channels.step not in (None, 1):
            raise TypeError("channels must be a contiguous slice")
        if self._squeeze and len(self._in_sample_shape) != len(self._raw.sample_shape):
            raise ValueError("channels= needs the unsqueezed (channel, polarisation) sample axes")
        full = super().read(offset, 0, **kwargs)   # bounds checks and the metadata of the whole band, no data
        for ignored in ("use_dask", "chunks"):
            kwargs.pop(ignored, None)
        if operator.index(offset) + operator.index(n) > len(self):
            from ._base import OutOfBoundsError
            raise OutOfBoundsError("Cannot read beyond end of stream")
        data = self._read_array(operator.index(offset), operator.index(n), channels=channels, **kwargs)
        # frequency bookkeeping of the subset through the container's own channel slicing (core.py:479-498)
        meta = full[:, channels] if hasattr(full, "channel_freqs") else full
        return type(full).like(meta, data, start_time=self.time_at(operator.index(offset)))

    def _read_array(self, offset, n, /, **kwargs):
        return self._finish(self._read_baseband(offset, n, **kwargs))

    def __getstate__(self):
        return self.__dict__.copy()


def _conj_where(t, mask):
    """Conjugate the series of device tensor ``t`` selected by the boolean ``mask`` (real-sampled data only: their
    complex form exists only after real_to_complex, so the decode pass cannot do it)."""
    import torch
    m = DeviceArray.from_host(np.broadcast_to(mask, t.shape[1:]).copy(), device=t.device.index).tensor
    return torch.where(m, t.conj(), t).resolve_conj()


class GUPPIRawReader(BasebandReader):
    """Dual-polarisation baseband data in GUPPI raw format; ``name`` may be a sequence of consecutive files
    (_baseband_readers.py:168-226).  Samples come out as (time, channel, polarisation)."""

    def __init__(self, name, /, **kwargs):
        hdr = open_raw(name, "guppi").header
        signal_kwargs = {"center_freq": u.Quantity(float(hdr["OBSFREQ"]), u.MHz), "freq_align": "center",
                         "pol_type": {"LIN": "linear", "CIRC": "circular"}[str(hdr["FD_POLN"]).strip()]}
        super().__init__(name, signal_type=DualPolarizationSignal, signal_kwargs=signal_kwargs,
                         lower_sideband=float(hdr["OBSBW"]) < 0, format="guppi", squeeze=False, **kwargs)

    def _axes(self):
        raw = self._raw   # stored (pol, chan) -> presented (chan, pol): swap the strides, no transpose pass
        return (raw.sample_shape[1], raw.sample_shape[0]), (raw.stride_2, raw.stride_1), raw.elem0


class DADAStokesReader(BasebandReader):
    """Full-Stokes intensity data in DADA format (NPOL = 4, NDIM = 1), presented as (time, channel, Stokes) with
    the channel axis reversed for a negative bandwidth (_baseband_readers.py:229-275)."""

    def __init__(self, name, /, **kwargs):
        hdr = open_raw(name, "dada").header
        if not (int(hdr["NPOL"]) == 4 and int(hdr["NDIM"]) == 1):
            raise ValueError("Does not look like Full Stokes data")
        bw, nchan = float(hdr["BW"]), int(hdr["NCHAN"])
        lsb = bw < 0
        signal_kwargs = {"center_freq": u.Quantity(float(hdr["FREQ"]), u.MHz),
                         "chan_bw": u.Quantity(abs(bw / nchan), u.MHz), "freq_align": "top" if lsb else "bottom"}
        super().__init__(name, signal_type=FullStokesSignal, signal_kwargs=signal_kwargs, lower_sideband=lsb,
                         format="dada", squeeze=False, **kwargs)

    def _axes(self):
        raw = self._raw   # stored (stokes, chan) -> (chan, stokes), channels reversed when the band is inverted
        npol, nchan = raw.sample_shape
        if self.lower_sideband:
            return (nchan, npol), (-raw.stride_2, raw.stride_1), raw.elem0 + (nchan - 1) * raw.stride_2
        return (nchan, npol), (raw.stride_2, raw.stride_1), raw.elem0
