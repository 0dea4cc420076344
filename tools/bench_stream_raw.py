"""BASELINE configs[3] (overlap-save streaming from host memory) fed with 8-bit payload bytes instead of complex64:
2^26 samples x 8 x 2 in 2^22-sample chunks, hop 615 915 (DM 56.77, 50 MHz channels at 1.4 GHz)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import units as u
from pulsarbat_amd.transforms.dedispersion import _crop_bounds, _plan_for

total_log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
total, chunk, nchan, npol = 1 << total_log2, 1 << 22, 8, 2
blank = np.broadcast_to(np.complex64(0), (chunk, nchan, npol))
head = pb.DualPolarizationSignal(blank, sample_rate=50 * u.MHz, center_freq=1.4 * u.GHz, pol_type="linear")
dm = pb.DM(56.77)
plan, _ = _plan_for(head, dm, head.center_freq, _crop_bounds(head, dm, head.center_freq))
hop = plan.nout
rng = np.random.default_rng(0)
for name, st in (("time-major payload [time][chan][pol]", dict(stride_t=nchan * npol, stride_c=npol, stride_p=1, blk=total)),
                 ("GUPPI blocks [chan][time][pol], 2^19 samples each", dict(stride_t=npol, stride_c=(1 << 19) * npol, stride_p=1, blk=1 << 19))):
    blk = st.pop("blk")
    raw = rng.integers(0, 256, total * nchan * npol * 2, dtype=np.uint8)
    lay = dict(nbits=8, ncomp=2, code=0, blk_samples=blk, blk_stride=blk * nchan * npol * 2, hdr_bytes=0, elem0=0, **st)
    y, ms = plan.dedisperse_stream_raw(raw, lay, total)
    t0 = time.perf_counter()
    y, ms = plan.dedisperse_stream_raw(raw, lay, total, out=y)
    wall = (time.perf_counter() - t0) * 1e3
    nchunk = len(y) // hop
    print(f"{name}: {nchunk} chunks, {ms:.0f} ms (events) / {wall:.0f} ms (wall): "
          f"{nchunk * chunk * nchan * npol / ms / 1e6:.2f} Gsamples/s through the GPU, "
          f"{len(y) * nchan * npol / ms / 1e6:.2f} Gsamples/s valid, upload {nchunk * chunk * nchan * npol * 2 / ms / 1e6:.1f} GB/s", flush=True)
x = np.zeros((total, nchan, npol), np.complex64)
y, ms = plan.dedisperse_stream(x)
y, ms = plan.dedisperse_stream(x, out=y)
nchunk = len(y) // hop
print(f"complex64 host input (pbh_dedisperse_stream): {ms:.0f} ms: {nchunk * chunk * nchan * npol / ms / 1e6:.2f} Gsamples/s through the GPU, "
      f"{len(y) * nchan * npol / ms / 1e6:.2f} Gsamples/s valid, upload {nchunk * chunk * nchan * npol * 8 / ms / 1e6:.1f} GB/s", flush=True)
