"""CPU restatement of the reader-side decode -- TEST INFRASTRUCTURE ONLY (never imported by pulsarbat_amd).

What the reference does with a baseband file: ``baseband`` unpacks the payload into float32 / complex64 with the
format's own sample shape, then pulsarbat/readers/_baseband_readers.py post-processes on the host:
  :136-153  _read_baseband: real data -> real_to_complex; conjugate the lower-sideband series; astype
  :223-226  GUPPIRawReader._read_array: (time, pol, chan) -> (time, chan, pol)
  :268-275  DADAStokesReader._read_array: flip channels if lower sideband, then (time, stokes, chan) -> (time, chan, stokes)
``baseband`` (a third-party dependency of the reference, requirements: baseband>=4.1.3) is absent from this
image, so its unpacking is restated from the formats' published layouts, in plain numpy, one whole file at a
time and independently of pulsarbat_amd/readers/_formats.py.  PARITY PINNING: the reference's own tests hold
these files (tests/data, copied to tests/golden/readers) but only assert shapes, times, rates and the
conjugation symmetry (tests/test_baseband_readers.py:16-145), which tests/test_readers.py reproduces; the
decoded VALUES are pinned by the format definitions only.
"""

import numpy as np

VDIF_2BIT = np.array([-3.316505, -1.0, 1.0, 3.316505], np.float32)   # baseband.base.encoding.OPTIMAL_2BIT_HIGH


def dada_samples(path):
    """(time, pol, chan) float32 / complex64 of an 8-bit DADA file."""
    blob = open(path, "rb").read()
    hdr = {}
    for line in blob[:4096].decode("ascii", "replace").splitlines():
        words = line.split("#")[0].split()
        if len(words) >= 2 and words[0] not in hdr:
            hdr[words[0]] = words[1]
    npol, nchan, ndim = int(hdr["NPOL"]), int(hdr["NCHAN"]), int(hdr["NDIM"])
    x = np.frombuffer(blob[int(hdr["HDR_SIZE"]):], np.int8).astype(np.float32)
    x = x[:x.size // (npol * nchan * ndim) * (npol * nchan * ndim)].reshape(-1, npol, nchan, ndim)
    return (x[..., 0] + 1j * x[..., 1]).astype(np.complex64) if ndim == 2 else x[..., 0]


def guppi_samples(paths):
    """(time, pol, chan) complex64 of consecutive 8-bit GUPPI raw files (overlap dropped)."""
    out = []
    for path in paths:
        blob = open(path, "rb").read()
        pos = 0
        while pos < len(blob):
            hdr = {}
            while True:
                card = blob[pos:pos + 80].decode("ascii", "replace")
                pos += 80
                if card.startswith("END"):
                    break
                hdr[card[:8].strip()] = card[9:].strip().strip("'").strip()
            nchan, bloc = int(hdr["OBSNCHAN"]), int(hdr["BLOCSIZE"])
            npol = 1 if int(hdr["NPOL"]) == 1 else 2
            blk = np.frombuffer(blob[pos:pos + bloc], np.int8).astype(np.float32).reshape(nchan, -1, npol, 2)
            pos += bloc
            keep = blk.shape[1] - int(hdr.get("OVERLAP", 0))
            z = (blk[..., 0] + 1j * blk[..., 1]).astype(np.complex64)[:, :keep]
            out.append(z.transpose(1, 2, 0))
    return np.concatenate(out, axis=0)


def vdif_samples(path):
    """(time, thread, chan) float32 of a real-sampled 2-bit VDIF file with 32-byte headers."""
    blob = np.frombuffer(open(path, "rb").read(), np.uint8)
    w = blob[:32].view("<u4")
    fbytes = int(w[2] & 0xFFFFFF) * 8
    nchan = 1 << int((w[2] >> 24) & 0x1F)
    frames = {}
    for pos in range(0, blob.size - fbytes + 1, fbytes):
        w = blob[pos:pos + 32].view("<u4")
        key = (int(w[0] & 0x3FFFFFFF), int(w[1] & 0xFFFFFF))
        pay = blob[pos + 32:pos + fbytes]
        codes = (pay[:, None] >> np.array([0, 2, 4, 6], np.uint8)) & 3      # low bits first
        frames.setdefault(key, {})[int((w[3] >> 16) & 0x3FF)] = VDIF_2BIT[codes.reshape(-1)].reshape(-1, nchan)
    sets = [np.stack([fs[t] for t in sorted(fs)], axis=1) for _, fs in sorted(frames.items())]
    return np.concatenate(sets, axis=0)


def sideband(z, lower_sideband):
    """_baseband_readers.py:146-151."""
    if lower_sideband is True:
        return z.conj()
    if lower_sideband is not False:
        z = z.copy()
        z[:, lower_sideband] = z[:, lower_sideband].conj()
    return z


def unpack_general(raw, layout, first, n, nchan, npol):
    """The addressing contract of pbh_decode (include/pbhip.h), element by element in numpy."""
    raw = np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw
    nc, nbits = layout["ncomp"], layout["nbits"]
    g = first + np.arange(n, dtype=np.int64)
    blk, w = g // layout["blk_samples"], g % layout["blk_samples"]
    e = (layout["elem0"] + w[:, None, None] * layout["stride_t"] + np.arange(nchan)[None, :, None] * layout["stride_c"]
         + np.arange(npol)[None, None, :] * layout["stride_p"])
    base = (blk * layout["blk_stride"] + layout["hdr_bytes"])[:, None, None]
    comp = []
    for k in range(nc):
        ci = e * nc + k
        if nbits == 8:
            v = raw[base + ci]
            comp.append((v.astype(np.int16) - 128).astype(np.float32) if layout["code"] else v.view(np.int8).astype(np.float32))
        elif nbits == 4:
            comp.append((((raw[base + (ci >> 1)] >> (4 * (ci & 1)).astype(np.uint8)) & 15).astype(np.float32) - 8))
        else:
            comp.append(VDIF_2BIT[(raw[base + (ci >> 2)] >> (2 * (ci & 3)).astype(np.uint8)) & 3])
    return (comp[0] + 1j * comp[1]).astype(np.complex64) if nc == 2 else comp[0]
