#!/usr/bin/env python3
"""Root-mode gather, shared buffer vs row-chunks, in ONE process.

The round-3 rehearsal (two ranks time-slicing one GPU) showed root mode 7.4 ms with the contiguous SharedBuffer and 5.2-7.1 ms
with hipIpc row-chunks, the repeats of one form spreading as widely as the forms.  What differs between the forms on the
ROOT is only where the pipeline's pitched last pass writes and what it costs to hand the result out: here a rank's share
(2^24 x 8 x 2) is dedispersed into a full-band (nout, 16, 2) destination of either kind, timed with events, alternating."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pulsarbat_amd as pb  # noqa: E402
from pulsarbat_amd import _hip  # noqa: E402
from pulsarbat_amd.node import MAX_NODE_BYTES, NodeBuffer, SharedBuffer  # noqa: E402

n, nchan, npol, total = 1 << 24, 8, 2, 16
start, stop = 1408404, 14607231
plan = _hip.Plan(n, nchan, npol, start, stop)
freqs = 1.4e9 + 25e6 * (np.arange(total) + 0.5 - total / 2)
plan.chirp_generate(56.77 / 2.41e-4 * 1e12, 1 / 25e6, freqs[:nchan], 1.4e9)
x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
nout, row = plan.nout, total * npol

shared = SharedBuffer((nout, total, npol), np.complex64, 0)
rows_per = min(1 << 30, MAX_NODE_BYTES) // (row * 8)
part_rows = list(range(0, nout, rows_per)) + [nout]
chunks = [NodeBuffer((part_rows[i + 1] - part_rows[i], total, npol), np.complex64, 0) for i in range(len(part_rows) - 1)]
full = pb.DeviceArray.empty((nout, total, npol), np.complex64)


def ev(fn, reps=8):
    fn()
    fn()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return sorted(out)[len(out) // 2]


def into_shared():
    plan.dedisperse_slices(x, [shared.ptr], [0, nout], row, 0)


def into_chunks():
    plan.dedisperse_slices(x, [c.ptr for c in chunks], part_rows, row, 0)


def join_chunks():
    for i, c in enumerate(chunks):
        full.tensor[part_rows[i]:part_rows[i + 1]].copy_(c.array.tensor)


def clone_shared():
    full.tensor.copy_(shared.array.tensor)


print(f"destination (nout={nout}, {total} chan, {npol} pol) = {nout * row * 8 / 1e9:.2f} GB; row-chunks: {len(chunks)} of <= {rows_per} rows")
for rnd in range(3):
    a, b = ev(into_shared), ev(into_chunks)
    c, d = ev(clone_shared), ev(join_chunks)
    print(f"round {rnd}: last pass into the shared buffer {a:.3f} ms, into row-chunks {b:.3f} ms | hand-out copy: shared {c:.3f} ms, "
          f"join of chunks {d:.3f} ms | root total (copy=True): shared {a + c:.3f}, chunked {b + d:.3f}; (copy=False): {a:.3f} / {b:.3f}")
