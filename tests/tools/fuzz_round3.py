"""Randomised checks of round 3's additions (test infrastructure; a few minutes on the GPU box):
  * the upload-once overlap-save stream (pbh_dedisperse_stream) with random epoch lengths, lengths (2^k, m*2^k, arbitrary),
    series counts and both precisions, against the concatenation of one-chunk calls of the same plan AND the oracle;
  * the same from raw 8-bit payloads in random block layouts (pbh_dedisperse_stream_raw) against the stream over the
    decoded array;
  * dedisperse_istft (fused and fall-back geometries) against istft(coherent_dedispersion(.)) of the product and the oracle.
usage: python tests/tools/fuzz_round3.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import pulsarbat_amd as pb
from pulsarbat_amd import _hip, units as u
from pulsarbat_amd.transforms.dedispersion import _crop_bounds, _plan_for, clear_plan_cache
from oracle import dedisp_oracle as orc, reader_oracle as ro

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t_end = time.time() + budget
ok = {"stream": 0, "raw": 0, "istft": 0}
bad = 0


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-30))


def sig(x, sr, fc, **kw):
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz, **kw)
    if x.ndim == 3 and x.shape[2] == 2:
        return pb.DualPolarizationSignal(x, pol_type="linear", **kw)
    return pb.BasebandSignal(x, **kw)


while time.time() < t_end:
    which = rng.integers(0, 3)
    try:
        if which == 0:     # ---- complex stream
            dtype = np.complex64 if rng.random() < 0.8 else np.complex128
            kind = rng.integers(0, 3)
            chunk = (1 << int(rng.integers(10, 17))) if kind == 0 else \
                    (int(rng.choice([3, 5, 7])) << int(rng.integers(9, 13))) if kind == 1 else int(rng.integers(1500, 70000))
            nchan = int(rng.integers(1, 6))
            tail = (nchan,) + ((2,) if rng.random() < 0.6 else ())
            sr, fc = 1e6, 1e9
            dm = float(rng.uniform(1, 40))
            x0 = np.zeros((chunk,) + tail, dtype)
            head = sig(x0, sr, fc)
            start, stop = _crop_bounds(head, pb.DM(dm), head.center_freq)
            hop = stop - start
            if hop <= 0:
                continue
            nchunk = int(rng.integers(2, 9))
            total = chunk + (nchunk - 1) * hop + int(rng.integers(0, hop))
            if total * int(np.prod(tail)) > (1 << 23):
                continue
            x = (rng.standard_normal((total,) + tail) + 1j * rng.standard_normal((total,) + tail)).astype(dtype)
            plan, _ = _plan_for(head, pb.DM(dm), head.center_freq, (start, stop))
            os.environ["PBH_STREAM_EPOCH"] = str(int(rng.integers(1, 6)))
            y, ms = plan.dedisperse_stream(x)
            st = plan.stream_stats()
            row = x.itemsize * int(np.prod(tail))
            assert st["h2d_bytes"] == (chunk + (nchunk - 1) * hop) * row, "upload volume"
            parts = [np.asarray(plan.dedisperse(np.ascontiguousarray(x[k * hop:k * hop + chunk]))) for k in range(nchunk)]
            assert np.array_equal(y, np.concatenate(parts)), "stream differs from per-chunk calls"
            k = int(rng.integers(0, nchunk))
            want = orc.coherent_dedispersion(x[k * hop:k * hop + chunk], dm, sr, fc)[0]
            tol = 1e-5 if dtype == np.complex64 else 1e-8
            assert rel(y[k * hop:(k + 1) * hop], want) < tol, "stream differs from the oracle"
            ok["stream"] += 1
            if dtype == np.complex64 and rng.random() < 0.5:   # ---- the same stream detected (pbh_plan_stream_detect)
                ns = int(rng.choice([1, 64, 128, 256]))
                mode = str(rng.choice(["intensity", "I", "linear", "circular"])) if len(tail) == 2 else "intensity"
                stop2 = stop - (stop - start) % ns
                if stop2 - start >= ns and total >= chunk:
                    plan2, _ = _plan_for(head, pb.DM(dm), head.center_freq, (start, stop2))
                    try:
                        plan2.stream_detect(mode, ns)
                    except NotImplementedError:
                        plan2 = None          # no fused tail for this plan (one-tile, few series, 7-smooth with nscrunch 1 ...)
                    if plan2 is not None:
                        try:
                            d, _ = plan2.dedisperse_stream(x)
                        finally:
                            plan2.stream_detect(None)
                        v, _ = plan2.dedisperse_stream(x)          # the voltages over the same (shortened) valid regions
                        v = v.reshape(len(v), nchan, -1)
                        vd = v.astype(np.complex128)
                        pw = vd.real ** 2 + vd.imag ** 2
                        if mode == "intensity":
                            w = pw
                        else:
                            ab = np.conj(vd[..., 0]) * vd[..., 1]
                            dd = pw[..., 0] - pw[..., 1]
                            w = {"I": pw.sum(-1), "linear": np.stack([pw.sum(-1), dd, 2 * ab.real, 2 * ab.imag], -1),
                                 "circular": np.stack([pw.sum(-1), 2 * ab.real, 2 * ab.imag, dd], -1)}[mode]
                        w = w.reshape((len(w) // ns, ns) + w.shape[1:]).sum(1)
                        assert d.size == w.size, "detected stream: shape"
                        assert np.abs(d.reshape(w.shape) - w).max() < 3e-5 * np.abs(w).max() * max(1.0, ns ** 0.5 / 4), "detected stream differs"
                        ok["detected"] = ok.get("detected", 0) + 1
        elif which == 1:   # ---- raw 8-bit stream in blocks
            nchan, npol = int(rng.integers(1, 5)), int(rng.choice([1, 2]))
            blk_t = int(rng.integers(200, 5000))
            hdr = int(rng.choice([0, 16, 64, 96]))
            order = rng.integers(0, 2)   # 0: time-major payload, 1: channel-major (GUPPI-like)
            if order == 0:
                st_t, st_c, st_p = nchan * npol, npol, 1
            else:
                st_t, st_c, st_p = npol, blk_t * npol, 1
            pay = blk_t * nchan * npol * 2
            stride = hdr + pay + int(rng.choice([0, 2, 14]))
            chunk = 1 << int(rng.integers(11, 15))
            sr, fc, dm = 1e6, 1e9, float(rng.uniform(1, 30))
            head = sig(np.zeros((chunk, nchan, npol), np.complex64), sr, fc)
            start, stop = _crop_bounds(head, pb.DM(dm), head.center_freq)
            hop = stop - start
            if hop <= 0:
                continue
            nchunk = int(rng.integers(2, 8))
            first = int(rng.integers(0, 3 * blk_t))
            total = chunk + (nchunk - 1) * hop + int(rng.integers(0, hop))
            nblk = (first + total + blk_t - 1) // blk_t + 1
            raw = rng.integers(0, 256, nblk * stride, dtype=np.uint8)
            lay = dict(nbits=8, ncomp=2, code=int(rng.integers(0, 2)), blk_samples=blk_t, blk_stride=stride, hdr_bytes=hdr, elem0=0,
                       stride_t=st_t, stride_c=st_c, stride_p=st_p)
            x = ro.unpack_general(raw, lay, first, total, nchan, npol)
            plan, _ = _plan_for(head, pb.DM(dm), head.center_freq, (start, stop))
            os.environ["PBH_STREAM_EPOCH"] = str(int(rng.integers(1, 5)))
            ya, _ = plan.dedisperse_stream(np.ascontiguousarray(x.astype(np.complex64)))
            yb, _ = plan.dedisperse_stream_raw(raw, lay, total, first=first)
            assert yb.shape == ya.shape and rel(yb, ya) < 2e-6, "raw stream differs from the stream over decoded samples"
            ok["raw"] += 1
        else:              # ---- dedisperse_istft
            M = 1 << int(rng.integers(1, 10))
            nchan_out = int(rng.integers(1, 9))
            tail = (2,) if rng.random() < 0.6 else ()
            nseg = (1 << int(rng.integers(9, 17))) + (0 if rng.random() < 0.7 else int(rng.integers(1, 50)))
            if nseg * M * nchan_out * max(int(np.prod(tail)), 1) > (1 << 23):
                continue
            sr, fc, dm = 8e6 / M, 1.3e9, float(rng.uniform(1, 60))
            ch = (rng.standard_normal((nseg, nchan_out * M) + tail) + 1j * rng.standard_normal((nseg, nchan_out * M) + tail)).astype(np.complex64)
            zc = sig(ch, sr, fc, freq_align="bottom", start_time=pb.Time(56000.0, format="mjd")).to_device()
            start, stop = _crop_bounds(zc, pb.DM(dm), zc.center_freq)
            if stop - start <= 0:
                continue
            y = pb.contrib.dedisperse_istft(zc, pb.DM(dm), nperseg=M)
            two = pb.contrib.istft(pb.coherent_dedispersion(zc, pb.DM(dm)), nperseg=M)
            assert y.shape == two.shape and np.allclose(np.asarray(y), np.asarray(two), atol=3e-5 * np.sqrt(M)), "fused differs from two steps"
            assert y.start_time.isclose(two.start_time) and u.isclose(y.sample_rate, two.sample_rate)
            mid = orc.coherent_dedispersion(ch, dm, sr, fc, freq_align="bottom")[0]
            assert rel(y, orc.istft(mid, M)) < 1e-5, "fused differs from the oracle"
            ok["istft"] += 1
    except AssertionError as exc:
        bad += 1
        print("FAIL", which, exc, flush=True)
    if sum(ok.values()) % 40 == 0:
        clear_plan_cache()
os.environ.pop("PBH_STREAM_EPOCH", None)
print(f"fuzz_round3: {ok} ok, {bad} failures")
sys.exit(1 if bad else 0)
