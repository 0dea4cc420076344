"""Random-case parity for the stand-alone transforms (pb.fft.fft / ifft of any length and batch) and the payload
decode (random layouts) -- the parts tools/fuzz_parity.py does not reach.  usage: fuzz_fft_decode.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import _hip
from oracle import reader_oracle as ro

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
nfft = ndec = bad = 0
worst = 0.0
while time.time() < t_end:
    # ---- fft ----
    kind = rng.integers(0, 4)
    if kind == 0:
        n = 1 << int(rng.integers(1, 22))
    elif kind == 1:
        n = int(rng.choice([3, 5, 7])) << int(rng.integers(10, 20))
    elif kind == 2:
        n = int(rng.integers(2, 300000))
    else:
        n = (1 << int(rng.integers(14, 21))) + int(rng.integers(-2, 3))
    batch = int(rng.integers(1, 9)) if n > 4096 else int(rng.integers(1, 200))
    while n * batch > (1 << 24):
        batch = max(1, batch // 2)
    dtype = np.complex64 if rng.random() < 0.75 else np.complex128
    x = (rng.standard_normal((n, batch)) + 1j * rng.standard_normal((n, batch))).astype(dtype)
    inv = bool(rng.integers(0, 2))
    got = np.asarray((pb.fft.ifft if inv else pb.fft.fft)(pb.DeviceArray.from_host(x), axis=0))
    want = (np.fft.ifft if inv else np.fft.fft)(x.astype(np.complex128), axis=0)
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    tol = 3e-6 if dtype == np.complex64 else 1e-11
    nfft += 1
    if dtype == np.complex64:
        worst = max(worst, err)
    if not err < tol:
        bad += 1
        print(f"FFT BAD n={n} batch={batch} {np.dtype(dtype).name} inverse={inv}: {err:.3e}", flush=True)
    # ---- decode ----
    nchan, npol = int(rng.integers(1, 40)), int(rng.integers(1, 5))
    nbits = int(rng.choice([8, 8, 2, 4]))
    ncomp = int(rng.integers(1, 3))
    code = int(rng.integers(0, 2)) if nbits == 8 else 0
    per = 4 // ncomp if nbits == 2 else 1          # elements per byte (2-bit) -- keep block payloads whole bytes
    blk_t = int(rng.integers(1, 3000)) * 4
    nblk = int(rng.integers(1, 12))
    hdr = int(rng.integers(0, 200))
    order = rng.permutation(3)                        # which axis is fastest
    dims = [blk_t, nchan, npol]
    strides = [0, 0, 0]
    acc = 1
    for ax in order:
        strides[ax] = acc
        acc *= dims[ax]
    flip_c = bool(rng.integers(0, 2))
    elem0 = 0
    if flip_c:
        elem0 = (nchan - 1) * strides[1]
        strides[1] = -strides[1]
    pay = (acc * ncomp * nbits + 7) // 8
    stride = hdr + pay + int(rng.integers(0, 64))
    raw = rng.integers(0, 256, nblk * stride, dtype=np.uint8)
    lay = dict(nbits=nbits, ncomp=ncomp, code=code, blk_samples=blk_t, blk_stride=stride, hdr_bytes=hdr, elem0=elem0,
               stride_t=strides[0], stride_c=strides[1], stride_p=strides[2])
    tot = blk_t * nblk
    first = int(rng.integers(0, tot))
    n = int(rng.integers(1, tot - first + 1))
    sm = bool(rng.integers(0, 2))
    conj = rng.integers(0, 2, (nchan, npol)).astype(bool) if ncomp == 2 and rng.random() < 0.5 else None
    got = np.asarray(_hip.decode(raw, lay, first, n, nchan, npol, conj=conj, series_major=sm))
    want = ro.unpack_general(raw, lay, first, n, nchan, npol)
    if conj is not None:
        want = np.where(conj[None], want.conj(), want)
    ndec += 1
    if not np.array_equal(got, want):
        bad += 1
        print(f"DECODE BAD {lay} first={first} n={n} nchan={nchan} npol={npol} sm={sm}", flush=True)
print(f"fft cases {nfft} (worst complex64 rel err {worst:.2e}), decode cases {ndec}, bad {bad}", flush=True)
sys.exit(1 if bad else 0)
