"""Generates tests/golden/*.npz from the CPU oracle (oracle/dedisp_oracle.py).

The reference ships no stored vectors for this path and cannot be imported here (astropy/dask
absent: ordinary ModuleNotFoundError), so these fixtures are outputs of the oracle -- the
numpy/scipy.fft restatement that tests/test_oracle.py pins against the reference's own known
answers and property tests.  They freeze the oracle's results so that (a) a later change to the
oracle is caught and (b) the GPU parity tests have inputs+expected outputs that do not depend on
the installed numpy/scipy build.  Contents follow SURVEY.md 8(c):
  delays.npz       DM = 2.41e-4 * a delay table (tests/test_dedispersion.py:13-32 values)
  chirp_spots.npz  f64 phase and c64 chirp at 64 bins/channel for configs 2, 3 and 5
  small_*.npz      full input + output, (8192, 4, 2) c64, DM in {10, 20, 50}, ref in {min, center, max}
  config1.npz      2^20 x 1 x 1, DM 0: seed, crop, sparse samples, L2 norm
  stokes.npz       hand-computed Stokes vectors (tests/test_polarization.py:38-48)
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dedisp_oracle as orc  # noqa: E402

MHz, GHz = 1e6, 1e9


def main():
    f = np.array([0.1, 1.0, 10.0])
    np.savez(os.path.join(HERE, "delays.npz"), f_mhz=f, dm=2.41e-4,
             delay_vs_inf=orc.time_delay(2.41e-4, f * MHz, np.inf),
             delay_2_vs_1=orc.time_delay(2.41e-4, 2 * MHz, 1 * MHz),
             scaled=np.array([orc.time_delay(2.41e-4 * a, 1 * MHz, np.inf) for a in (10, 20, 100)]))

    N = 1 << 24
    spots = {}
    for name, dm, nchan in (("config2", 56.77, 8), ("config3", 56.77, 64), ("config5", 1000.0, 64)):
        sr = 400 * MHz / nchan
        freqs = orc.channel_freqs(1.4 * GHz, sr, nchan)
        rng = np.random.default_rng(len(name))
        bins = np.unique(np.concatenate([[0, 1, N // 2 - 1, N // 2, N // 2 + 1, N - 1],
                                         rng.integers(0, N, 58)]))
        chans = [0, nchan // 2, nchan - 1]
        ph = np.stack([orc.phase_cycles(dm, N, 1 / sr, freqs[c], 1.4 * GHz, bins) for c in chans])
        spots[name + "_bins"] = bins
        spots[name + "_chans"] = np.array(chans)
        spots[name + "_phase_cycles"] = ph
        spots[name + "_chirp"] = np.exp(-2j * np.pi * (ph - np.rint(ph))).astype(np.complex64)
        spots[name + "_crop"] = np.array(orc.crop_bounds(dm, N, nchan, sr, 1.4 * GHz, 1.4 * GHz))
    np.savez(os.path.join(HERE, "chirp_spots.npz"), **spots)

    shape, sr, fc = (8192, 4, 2), 1 * MHz, 1 * GHz
    fmin, fmax = orc.band_edges(fc, sr, 4)
    for dm in (10, 20, 50):
        x = orc.synthetic_block(shape, 20260000 + dm)
        out = {"x": x, "dm": dm, "sample_rate": sr, "center_freq": fc}
        for tag, ref in (("min", fmin), ("center", fc), ("max", fmax)):
            y, start, stop = orc.coherent_dedispersion(x, float(dm), sr, fc, ref_freq_hz=ref)
            out["y_" + tag] = y
            out["crop_" + tag] = np.array([start, stop])
            out["ref_" + tag] = ref
        np.savez_compressed(os.path.join(HERE, f"small_dm{dm}.npz"), **out)

    x = orc.synthetic_block((1 << 20, 1, 1), 20260001)
    y, start, stop = orc.coherent_dedispersion(x, 0.0, 400 * MHz, 1.4 * GHz)
    idx = np.random.default_rng(1).integers(0, 1 << 20, 4096)
    np.savez(os.path.join(HERE, "config1.npz"), seed=20260001, crop=np.array([start, stop]),
             idx=idx, y_at_idx=y[idx, 0, 0], x_at_idx=x[idx, 0, 0], l2=np.linalg.norm(y))

    x = np.array([[[1 + 1j, 2 + 1j]], [[3 + 0j, 0 + 4j]], [[0 + 2j, 3 + 1j]]], dtype=np.complex128)
    np.savez(os.path.join(HERE, "stokes.npz"), x=x,
             linear=np.array([[[7, -3, 6, -2]], [[25, -7, 0, 24]], [[14, -6, 4, -12]]]),
             circular=np.array([[[7, 6, -2, -3]], [[25, 0, 24, -7]], [[14, 4, -12, -6]]]))


if __name__ == "__main__":
    main()
