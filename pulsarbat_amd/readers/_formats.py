"""Host-side parsing of baseband file formats: headers and the position / geometry of the sample payloads.
(The reference gets this from the third-party ``baseband`` package -- ``baseband.open(name, 'rs', ...)`` at
_baseband_readers.py:104 -- which is not a dependency here.)  Nothing in this module touches sample values: it
yields a ``RawStream`` -- where the payload blocks are and how elements are laid out inside one -- and
``RawStream.fetch`` reads the bytes of the needed blocks; unpacking them is ``pbh_decode`` on the GPU.

Formats, written from their public specifications:
  * DADA (PSRDADA): one ASCII header of HDR_SIZE bytes (``KEY value`` lines), then samples ordered
    time, polarisation, channel, (re, im); 8-bit two's complement.
  * GUPPI raw: a sequence of blocks, each 80-character ``KEY = value`` cards up to ``END`` (padded to 512 bytes
    when DIRECTIO is set) followed by BLOCSIZE bytes ordered channel, time, polarisation, (re, im), 8-bit two's
    complement; the last OVERLAP time samples of a block repeat at the start of the next.
  * VDIF: frames of a 32-byte (16-byte legacy) header and a payload of 32-bit little-endian words filled from
    the low bits up; one thread per frame, threads of the same instant form a frame set; 2-bit samples are the
    4-level code, 8-bit samples offset binary.
"""

import os
from dataclasses import dataclass, field

import numpy as np

from .. import units as u
from ..time import Time

__all__ = ["RawStream", "open_raw", "sniff_format"]


@dataclass
class RawStream:
    """Geometry of a stream of payload blocks.  Sample axes after decoding: (time, axis1, axis2) with
    ``shape1`` the (axis1, axis2) extents, and element (t, i, j) of a block at element index
    ``elem0 + t*stride_t + i*stride_1 + j*stride_2``."""
    fmt: str
    header: dict
    nsample: int                 # valid time samples in the whole stream
    sample_shape: tuple          # (axis1, axis2) as the format's own reader would present them (unsqueezed)
    complex_data: bool
    nbits: int
    code: int                    # pbh_decode payload code
    sample_rate: object          # Quantity: rate of the samples as stored (real samples for real data)
    start_time: object           # Time of sample 0
    blk_samples: int             # valid time samples per block
    blk_bytes: int               # payload bytes per block
    elem0: int
    stride_t: int
    stride_1: int
    stride_2: int
    blocks: list = field(default_factory=list, repr=False)   # (file index, byte position of the payload)
    files: list = field(default_factory=list, repr=False)
    gather: object = field(default=None, repr=False)          # optional per-block reader (VDIF frame sets)
    scale: float = 1.0           # factor applied to the unpacked integers (VDIF 4- / 8-bit normalisation)
    bytes_fetched: int = field(default=0, repr=False, compare=False)   # of the last fetch()

    def layout(self):
        return dict(nbits=self.nbits, ncomp=2 if self.complex_data else 1, code=self.code, blk_samples=self.blk_samples,
                    blk_stride=self.blk_bytes, hdr_bytes=0, elem0=self.elem0, stride_t=self.stride_t,
                    stride_c=self.stride_1, stride_p=self.stride_2)

    def fetch(self, offset, n, byte_range=None):
        """Payload bytes of the blocks holding samples [offset, offset + n), one after the other, and the
        index of sample ``offset`` counted from the start of the first of them.  ``byte_range = (lo, hi)`` reads only
        bytes [lo, hi) of every block's payload (a channel subset of a channel-major block), packed ``hi - lo`` rounded
        up to 16 bytes apart.  ``bytes_fetched`` holds the number of bytes the last call read from the files."""
        if n <= 0:
            return np.empty(0, np.uint8), 0
        b0, b1 = offset // self.blk_samples, (offset + n - 1) // self.blk_samples
        lo, hi = (0, self.blk_bytes) if byte_range is None or self.gather is not None else byte_range
        if not 0 <= lo < hi <= self.blk_bytes:
            raise ValueError("byte_range outside the payload")
        width = hi - lo
        pitch = self.blk_bytes if width == self.blk_bytes else -(-width // 16) * 16
        buf = np.zeros((b1 - b0 + 1) * pitch, np.uint8) if pitch != width else np.empty((b1 - b0 + 1) * pitch, np.uint8)
        handles = {}
        try:
            for k, b in enumerate(range(b0, b1 + 1)):
                dst = memoryview(buf)[k * pitch:k * pitch + width]
                if self.gather is not None:
                    self.gather(self, b, dst, handles)
                    continue
                fi, pos = self.blocks[b]
                fh = handles.get(fi)
                if fh is None:
                    fh = handles[fi] = open(self.files[fi], "rb", buffering=0)
                fh.seek(pos + lo)
                if fh.readinto(dst) != width:
                    raise EOFError(f"{self.files[fi]}: short read of block {b}")
        finally:
            for fh in handles.values():
                fh.close()
        self.bytes_fetched = (b1 - b0 + 1) * width
        return buf, offset - b0 * self.blk_samples


def _as_files(name):
    if isinstance(name, (str, os.PathLike)):
        return [os.fspath(name)]
    return [os.fspath(f) for f in name]


def sniff_format(path):
    with open(path, "rb") as fh:
        head = fh.read(80)
    if head.startswith(b"HEADER") or b"HDR_SIZE" in head:
        return "dada"
    if len(head) == 80 and head[8:10] == b"= ":
        return "guppi"
    return "vdif"


# ---- DADA ---------------------------------------------------------------------------------------------------
def _dada_header(path):
    with open(path, "rb") as fh:
        text = fh.read(4096)
        hdr = {}
        size = 4096
        for line in text.decode("ascii", "replace").split("\n"):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            parts = line.split(None, 1)
            if len(parts) == 2:
                hdr.setdefault(parts[0], parts[1].strip())
        size = int(hdr.get("HDR_SIZE", 4096))
    return hdr, size


def _open_dada(files):
    if len(files) != 1:
        raise ValueError("DADA: one file per stream")
    hdr, hsize = _dada_header(files[0])
    nbit, ndim = int(hdr["NBIT"]), int(hdr["NDIM"])
    npol, nchan = int(hdr["NPOL"]), int(hdr["NCHAN"])
    if nbit != 8:
        raise ValueError(f"DADA: NBIT={nbit} is not supported (8-bit samples only)")
    elem_bytes = ndim * nbit // 8
    sample_bytes = elem_bytes * npol * nchan
    nsample = (os.path.getsize(files[0]) - hsize) // sample_bytes
    rate = u.Quantity(1.0 / float(hdr["TSAMP"]), u.MHz)
    y, mo, d, hms = hdr["UTC_START"].split("-", 3)
    t0 = Time(f"{y}-{mo}-{d}T{hms}", format="isot", precision=9)
    t0 = t0 + u.Quantity(int(hdr.get("OBS_OFFSET", 0)) / (sample_bytes * float(rate.to(u.Hz).value)), u.s)
    return RawStream(fmt="dada", header=hdr, nsample=nsample, sample_shape=(npol, nchan), complex_data=ndim == 2,
                     nbits=8, code=0, sample_rate=rate, start_time=t0, blk_samples=max(nsample, 1),
                     blk_bytes=nsample * sample_bytes, elem0=0, stride_t=npol * nchan, stride_1=nchan, stride_2=1,
                     blocks=[(0, hsize)], files=files)


# ---- GUPPI raw ------------------------------------------------------------------------------------------------
def _guppi_cards(fh, pos):
    """Header cards of the block starting at byte ``pos``: (dict, position of the payload)."""
    fh.seek(pos)
    hdr = {}
    nbytes = 0
    while True:
        card = fh.read(80)
        if len(card) < 80:
            return None, pos
        nbytes += 80
        if card.startswith(b"END"):
            break
        key, _, val = card.decode("ascii", "replace").partition("=")
        val = val.strip()
        if val.startswith("'"):
            val = val.strip("'").strip()
        else:
            try:
                val = int(val)
            except ValueError:
                try:
                    val = float(val)
                except ValueError:
                    pass
        hdr[key.strip()] = val
    if hdr.get("DIRECTIO", 0):
        nbytes = -(-nbytes // 512) * 512
    return hdr, pos + nbytes


def _open_guppi(files):
    blocks, first = [], None
    for fi, path in enumerate(files):
        size = os.path.getsize(path)
        with open(path, "rb") as fh:
            pos = 0
            while pos < size:
                hdr, pay = _guppi_cards(fh, pos)
                if hdr is None:
                    break
                bloc = int(hdr["BLOCSIZE"])
                if hdr.get("DIRECTIO", 0):
                    nxt = pay + -(-bloc // 512) * 512
                else:
                    nxt = pay + bloc
                if pay + bloc > size:
                    break   # truncated last block
                if first is None:
                    first = hdr
                elif any(hdr.get(k) != first.get(k) for k in ("BLOCSIZE", "OBSNCHAN", "NPOL", "NBITS", "OVERLAP")):
                    raise ValueError(f"{path}: block geometry changes inside the stream")
                blocks.append((fi, pay))
                pos = nxt
    if first is None:
        raise ValueError("GUPPI: no complete block found")
    nchan = int(first["OBSNCHAN"])
    npol = 1 if int(first["NPOL"]) == 1 else 2
    nbits = int(first["NBITS"])
    if nbits != 8:
        raise ValueError(f"GUPPI: NBITS={nbits} is not supported (8-bit samples only)")
    bloc = int(first["BLOCSIZE"])
    ntime = bloc // (2 * npol * nchan)
    overlap = int(first.get("OVERLAP", 0))
    valid = ntime - overlap
    rate = u.Quantity(1.0 / float(first["TBIN"]), u.Hz)
    per_packet = int(first["PKTSIZE"]) * 8 // (nbits * 2 * npol * nchan) if "PKTSIZE" in first else 0
    t0 = Time(int(first["STT_IMJD"]), (float(first["STT_SMJD"]) + float(first.get("STT_OFFS", 0))) / 86400.0, format="mjd")
    t0 = t0 + u.Quantity(int(first.get("PKTIDX", 0)) * per_packet / float(rate.value), u.s)
    # every block contributes its first `valid` samples (the trailing overlap of the last block is not served)
    nsample = valid * len(blocks)
    return RawStream(fmt="guppi", header=first, nsample=nsample, sample_shape=(npol, nchan), complex_data=True,
                     nbits=8, code=0, sample_rate=rate, start_time=t0, blk_samples=valid, blk_bytes=bloc, elem0=0,
                     stride_t=npol, stride_1=1, stride_2=ntime * npol, blocks=blocks, files=files)


# ---- VDIF ---------------------------------------------------------------------------------------------------
def _vdif_words(fh, pos):
    fh.seek(pos)
    raw = fh.read(32)
    if len(raw) < 32:
        return None
    return np.frombuffer(raw, "<u4")


def _vdif_gather(stream, b, dst, handles):
    """Payloads of the frames of frame set ``b`` in thread order, one after the other."""
    fh = handles.get(0)
    if fh is None:
        fh = handles[0] = open(stream.files[0], "rb", buffering=0)
    pay = stream.header["payload_bytes"]
    for k, pos in enumerate(stream.blocks[b]):
        fh.seek(pos)
        if fh.readinto(dst[k * pay:(k + 1) * pay]) != pay:
            raise EOFError(f"{stream.files[0]}: short read in frame set {b}")


def _open_vdif(files):
    if len(files) != 1:
        raise ValueError("VDIF: one file per stream")
    size = os.path.getsize(files[0])
    sets, order = {}, []
    with open(files[0], "rb") as fh:
        w = _vdif_words(fh, 0)
        if w is None:
            raise ValueError("VDIF: file too short")
        legacy = bool((w[0] >> 30) & 1)
        hbytes = 16 if legacy else 32
        frame_bytes = int(w[2] & 0xFFFFFF) * 8
        nchan = 1 << int((w[2] >> 24) & 0x1F)
        bps = int((w[3] >> 26) & 0x1F) + 1
        complex_data = bool((w[3] >> 31) & 1)
        edv = 0 if legacy else int(w[4] >> 24)
        first = w.copy()
        pos = 0
        while pos + frame_bytes <= size:
            w = _vdif_words(fh, pos)
            if w is None:
                break
            if not (w[0] >> 31) & 1:   # invalid-data frames are skipped
                key = (int(w[0] & 0x3FFFFFFF), int(w[1] & 0xFFFFFF))
                if key not in sets:
                    sets[key] = {}
                    order.append(key)
                sets[key][int((w[3] >> 16) & 0x3FF)] = pos + hbytes
            pos += frame_bytes
    threads = sorted(sets[order[0]])
    order = [k for k in order if sorted(sets[k]) == threads]   # complete frame sets only
    if bps not in (2, 4, 8):
        raise ValueError(f"VDIF: {bps}-bit samples are not supported (2, 4 or 8)")
    pay = frame_bytes - hbytes
    ncomp = 2 if complex_data else 1
    per_frame = pay * 8 // (bps * ncomp * nchan)
    # sample rate: EDV 1 / 3 carry it; otherwise count the frames of one second
    if edv in (1, 3):
        val = int(first[4] & 0x7FFFFF) * (1e6 if (first[4] >> 23) & 1 else 1e3)
        rate_hz = val if complex_data else 2 * val   # the field is the bandwidth: real data are sampled at twice it
    else:
        sec0 = order[0][0]
        per_sec = max(k[1] for k in order if k[0] == sec0) + 1
        rate_hz = per_sec * per_frame
    epoch = int((first[1] >> 24) & 0x3F)
    t0 = Time(f"{2000 + epoch // 2}-{'01' if epoch % 2 == 0 else '07'}-01T00:00:00", format="isot", precision=9)
    t0 = t0 + u.Quantity(order[0][0] + order[0][1] * per_frame / rate_hz, u.s)
    blocks = [[sets[k][t] for t in threads] for k in order]
    hdr = {"payload_bytes": pay, "threads": threads, "edv": edv, "frame_bytes": frame_bytes}
    return RawStream(fmt="vdif", header=hdr, nsample=per_frame * len(order), sample_shape=(len(threads), nchan),
                     complex_data=complex_data, nbits=bps, code=1 if bps == 8 else 0,
                     sample_rate=u.Quantity(rate_hz, u.Hz), start_time=t0, blk_samples=per_frame,
                     blk_bytes=pay * len(threads), elem0=0, stride_t=nchan, stride_1=per_frame * nchan, stride_2=1,
                     blocks=blocks, files=files, gather=_vdif_gather,
                     # multi-bit VDIF samples are scaled to the 2-bit convention (low level = 1), as `baseband` does:
                     # 4 bits: (v - 8) / 2.95, 8 bits: (v - 128) / 35.5
                     scale={2: 1.0, 4: 1.0 / 2.95, 8: 2.0 / 71.0}[bps])


def open_raw(name, format=None):
    """Parse the headers of ``name`` (a path or, for GUPPI, a sequence of consecutive files)."""
    files = _as_files(name)
    fmt = (format or sniff_format(files[0])).lower()
    if fmt == "dada":
        return _open_dada(files)
    if fmt == "guppi":
        return _open_guppi(files)
    if fmt == "vdif":
        return _open_vdif(files)
    raise ValueError(f"unsupported baseband format {fmt!r}")
