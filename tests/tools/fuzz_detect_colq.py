"""Randomised check of the detection inside the inverse column pass (k_colq<1024, INV, DET> + k_detect_reduce; test
infrastructure, a few tens of ms per case on the GPU box): blocks of 2^22, 2^23 and 2^24 samples with random channel / polarisation counts,
DMs (crop starts of every residue mod 16, crops from a few rows to most of the block), reference frequencies, scrunch
factors 1 (the detecting layout pass) and 64 ... 16384, every detect mode, sample-major and series-major input, against the scrunched power of the
voltages the ordinary call returns (float64 sums on the host).
usage: python tests/tools/fuzz_detect_colq.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
residues = set()
while time.time() < t_end:
    n = 1 << int(rng.choice([20, 21, 22, 23, 24]))   # 64- to 1024-row column tiles (256 to 16 columns)
    nchan = int(rng.integers(1, 4))
    npol = int(rng.choice([1, 2]))
    sr = float(rng.choice([1e6, 6.25e6, 50e6]))
    fc = float(rng.uniform(0.4e9, 2e9))
    dm = float(10 ** rng.uniform(-1, 3))
    mode = str(rng.choice(["I", "I", "intensity", "linear", "circular"])) if npol == 2 else "intensity"
    ns = 1 if rng.random() < 0.25 else 1 << int(rng.integers(6, 15))   # 1: the last layout pass detects (k_reinterleave_p2<.., DET>)
    r = rng.random()
    rf = None if r < 0.6 else (fc + sr * nchan / 2 if r < 0.8 else fc - sr * nchan / 2) * u.Hz
    shape = (n, nchan) + ((2,) if npol == 2 else ())
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    xt = torch.view_as_complex(torch.randn(shape + (2,), device="cuda", generator=g))
    kw = dict(sample_rate=sr * u.Hz, center_freq=fc * u.Hz)
    z = (pb.DualPolarizationSignal(pb.DeviceArray(xt), pol_type="linear", **kw) if npol == 2
         else pb.BasebandSignal(pb.DeviceArray(xt), **kw))
    try:
        y = pb.coherent_dedispersion(z, pb.DM(dm), ref_freq=rf)
    except ValueError:
        continue   # nothing left after the crop
    if len(y) < ns:
        continue
    yt = y.data.tensor.reshape(len(y), nchan, npol)
    nout = len(y) // ns
    pw = (yt.real.double() ** 2 + yt.imag.double() ** 2)[:nout * ns].reshape(nout, ns, nchan, npol).sum(1)
    scale = pw.sum(-1, keepdim=True).cpu().numpy()
    if mode in ("linear", "circular"):   # all four parameters (core.py:930-966)
        ab = (yt[..., 0].conj().to(torch.complex128) * yt[..., 1].to(torch.complex128))[:nout * ns].reshape(nout, ns, nchan).sum(1)
        d = pw[..., 0] - pw[..., 1]
        quv = (d, 2 * ab.real, 2 * ab.imag) if mode == "linear" else (2 * ab.real, 2 * ab.imag, d)
        want = torch.stack((pw.sum(-1),) + quv, dim=-1).cpu().numpy()
    else:
        want = (pw.sum(-1) if mode == "I" else (pw if npol == 2 else pw[..., 0])).cpu().numpy()
        scale = want
    got, start = pb.dedisperse_detect(z, pb.DM(dm), ref_freq=rf, mode=mode, nscrunch=ns)
    got = np.asarray(got).reshape(want.shape)
    zs = type(z).like(z, z.data.to_series_major())
    got_s, _ = pb.dedisperse_detect(zs, pb.DM(dm), ref_freq=rf, mode=mode, nscrunch=ns)
    # bound as in tests/_detect_check.py: every output against its own Stokes I from 16 summed samples on, against the MEAN power
    # below that (a single |z|^2 can be arbitrarily small, and the sample-major block runs the four-pass schedule since round 4:
    # its voltages agree with the series-major route's to 2e-7 of the rms, not to the last bit)
    ref_scale = scale if ns >= 16 else np.full_like(scale, float(np.mean(scale)))
    err = float(np.max(np.abs(got - want) / ref_scale))
    same = float(np.max(np.abs(np.asarray(got_s).reshape(want.shape) - want) / ref_scale)) < 3e-5
    residues.add(start % 16)
    cases += 1
    if not (err < 3e-5 and same):
        bad += 1
        print(f"FAIL seed {seed} case {cases}: nchan {nchan} npol {npol} sr {sr:g} fc {fc:.6g} dm {dm:.6g} mode {mode} ns {ns} "
              f"ref {rf} start {start} nout {nout}: max rel {err:.2e}, series-major input within tolerance {same}", flush=True)
    del xt, z, y, yt, pw, zs
print(f"fuzz_detect_colq seed {seed}: {cases} cases, {bad} failures, crop-start residues mod 16 seen: {sorted(residues)}")
sys.exit(1 if bad else 0)
