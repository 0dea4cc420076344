"""Freeze the reader oracle's output for the reference's data files (tests/golden/readers) as a fixture:
first / last samples of every file after the reference's post-processing.  Run from the repository root:
    python tests/golden/make_golden_readers.py
(The files themselves are the reference's own test data, tests/data in its tree.)"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import reader_oracle as ro

D = Path(__file__).parent / "readers"
out = {}
x = ro.dada_samples(D / "sample.dada")
out["dada_head"], out["dada_tail"] = x[:32], x[-32:]
x = ro.vdif_samples(D / "sample.vdif")
out["vdif_head"], out["vdif_tail"] = x[:64], x[-64:]
x = ro.guppi_samples(sorted(D.glob("fake.*.raw"))).transpose(0, 2, 1)
out["guppi_head"], out["guppi_mid"], out["guppi_tail"] = x[:32], x[8192 - 16:8192 + 16], x[-32:]
x = np.flip(ro.dada_samples(D / "stokes_ef.dada"), axis=-1).transpose(0, 2, 1)
out["stokes_first"], out["stokes_last"] = x[0], x[-1]
np.savez_compressed(Path(__file__).parent / "readers_expected.npz", **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
