"""Random payload layouts through pbh_dedisperse_stream_raw against pbh_dedisperse_stream over the numpy-decoded
samples.  usage: fuzz_stream_raw.py [seconds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from pulsarbat_amd.transforms.dedispersion import _crop_bounds, _plan_for, clear_plan_cache
from oracle import reader_oracle as ro

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
t_end = time.time() + budget
ok = bad = 0
while time.time() < t_end:
    nchan, npol = int(rng.integers(1, 6)), int(rng.integers(1, 3))
    code = int(rng.integers(0, 2))
    chunk = 1 << int(rng.integers(12, 18))
    blk_t = int(rng.integers(chunk // 8, 3 * chunk))
    hdr = int(rng.integers(0, 100))
    order = rng.permutation(3)
    dims = [blk_t, nchan, npol]
    strides = [0, 0, 0]
    acc = 1
    for ax in order:
        strides[ax] = acc
        acc *= dims[ax]
    pay = acc * 2
    stride = hdr + pay + int(rng.integers(0, 40))
    total = int(rng.integers(chunk, 6 * chunk))
    first = int(rng.integers(0, blk_t))
    nblk = (first + total + blk_t - 1) // blk_t
    raw = rng.integers(0, 256, nblk * stride, dtype=np.uint8)
    lay = dict(nbits=8, ncomp=2, code=code, blk_samples=blk_t, blk_stride=stride, hdr_bytes=hdr, elem0=0,
               stride_t=strides[0], stride_c=strides[1], stride_p=strides[2])
    conj = rng.integers(0, 2, (nchan, npol)).astype(bool) if rng.random() < 0.5 else None
    x = ro.unpack_general(raw, lay, first, total, nchan, npol) * np.float32(1 / 32)
    if conj is not None:
        x = np.where(conj[None], x.conj(), x)
    z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear") if npol == 2 else \
        pb.BasebandSignal(x[:, :, 0], sample_rate=1 * u.MHz, center_freq=1 * u.GHz)
    dm = float(rng.uniform(0.5, 40.0)) * chunk / 65536
    head = z[:chunk]
    try:
        crop = _crop_bounds(head, pb.DM(dm), head.center_freq)
    except Exception:
        continue
    if crop[1] - crop[0] < 16:
        continue
    plan, _ = _plan_for(head, pb.DM(dm), head.center_freq, crop)
    ya, _ = plan.dedisperse_stream(np.ascontiguousarray(x.reshape(total, nchan, npol)))
    yb, _ = plan.dedisperse_stream_raw(raw, lay, total, first=first, conj=conj, scale=1 / 32)
    e = np.linalg.norm(yb - ya) / max(np.linalg.norm(ya), 1e-30)
    if not (yb.shape == ya.shape and e < 3e-6):
        bad += 1
        print(f"BAD lay={lay} first={first} total={total} chunk={chunk} nchan={nchan} npol={npol} err={e:.2e}", flush=True)
    else:
        ok += 1
    if (ok + bad) % 40 == 0:
        clear_plan_cache()
print(f"cases ok {ok}, bad {bad}", flush=True)
sys.exit(1 if bad else 0)
