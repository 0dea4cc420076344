"""Secondary measurements for DESIGN.md: BASELINE.json configs[3] (streaming overlap-save) and
configs[4] (dedisperse + Stokes-I + 1024x scrunch, DM 1000) on ONE MI355X."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from pulsarbat_amd import _hip
from pulsarbat_amd.device import DeviceArray
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

def crop(dm, n, band, center, sr):
    d = pb.DM(dm)
    top = d.sample_delay((center + band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    bot = d.sample_delay((center - band / 2) * u.Hz, center * u.Hz, sr * u.Hz)
    import math
    return math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))

def config5(steps=10):
    # this GPU's 8 of the 64 channels of 6.25 MHz (BASELINE configs[4] geometry)
    n, nchan_tot, nchan, npol, dm, band, center = 1 << 24, 64, 8, 2, 1000.0, 400e6, 1.4e9
    sr = band / nchan_tot
    start, stop = crop(dm, n, band, center, sr)
    freqs = (center + sr * (np.arange(nchan_tot) + 0.5 - nchan_tot / 2))[:nchan]
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda") * 0.7071))
    plan = _hip.Plan(n, nchan, npol, start, stop)
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    out = DeviceArray.empty(((stop - start) // 1024, nchan), np.float32)
    for _ in range(3):
        plan.dedisperse_detect(x, nscrunch=1024, mode="I", out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan.dedisperse_detect(x, nscrunch=1024, mode="I", out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    y = DeviceArray.empty((stop - start, nchan, npol), np.complex64)
    for _ in range(3):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan.dedisperse(x, out=y)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / steps
    xs = x.to_series_major()
    for _ in range(3):
        plan.dedisperse_detect(xs, nscrunch=1024, mode="I", out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        plan.dedisperse_detect(xs, nscrunch=1024, mode="I", out=out)
    torch.cuda.synchronize(); dt3 = (time.perf_counter() - t0) / steps
    ns = n * nchan * npol
    return {"ms_fused_detect_series_major_input": dt3 * 1e3, "Msamples_per_s_series_major_input": ns / dt3 / 1e6,
            "config": "configs[4] per-GPU share: 2^24 x 8 of 64 chan x 2 pol, DM 1000, Stokes-I + 1024x scrunch",
            "crop": [start, stop], "out_shape": list(out.shape), "ms_fused_detect": dt * 1e3,
            "Msamples_per_s_fused": ns / dt / 1e6, "ms_voltage_output": dt2 * 1e3,
            "alg_bytes_per_sample": 60.0, "GBps_at_60B": 60.0 * ns / dt / 1e9}

def config4(total_log2=26, chunk_log2=22, detect=None, nscrunch=1024):
    """BASELINE configs[3]: overlap-save stream from pinned host memory; every input row crosses PCIe once.
    detect: the same stream as a filterbank stream (pbh_plan_stream_detect): the download is the detected rows only."""
    nchan, npol, dm, band, center = 8, 2, 56.77, 400e6, 1.4e9
    sr = band / nchan
    n = 1 << chunk_log2
    start, stop = crop(dm, n, band, center, sr)
    freqs = center + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    total = 1 << total_log2
    # one random 2^24-sample block repeated with a different complex factor per repeat (34 GB of randn takes minutes)
    blk = min(total, 1 << 24)
    base = torch.randn((blk, nchan, npol, 2), dtype=torch.float32).numpy().view(np.complex64).reshape(blk, nchan, npol)
    x = np.empty((total, nchan, npol), np.complex64)
    for k in range(total // blk):
        np.multiply(base, np.complex64(np.exp(0.37j * k) * (1 + 0.01 * k)), out=x[k * blk:(k + 1) * blk])
    if detect:
        stop -= (stop - start) % nscrunch
    plan = _hip.Plan(n, nchan, npol, start, stop)
    plan.chirp_generate(dm / 2.41e-4 * 1e12, 1 / sr, freqs, center)
    hop = stop - start
    if detect:
        tail = plan.stream_detect(detect, nscrunch)
        out = np.empty((((total - n) // hop + 1) * (hop // nscrunch),) + tail, np.float32)
    else:
        out = np.empty((((total - n) // hop + 1) * hop, nchan, npol), np.complex64)
    out[::4096] = 0      # touch the pages: first-touch faults are the allocator's cost, not the stream's
    t0 = time.perf_counter()
    y, ms = plan.dedisperse_stream(x, out=out)
    wall = time.perf_counter() - t0
    st = plan.stream_stats()
    nchunk = st["nchunk"]
    what = f", Stokes-{detect} + {nscrunch}x scrunch inside every chunk (filterbank stream)" if detect else ""
    return {"config": f"configs[3]: 2^{total_log2} samples x 8 x 2 streamed in 2^{chunk_log2} chunks (overlap-save, upload once){what}",
            "hop": hop, "nchunk": nchunk, "valid_fraction_of_a_chunk": hop / n, "ms_stream_events": ms,
            "wall_s_incl_pinning": wall, "H2D_GB": st["h2d_bytes"] / 1e9, "D2H_GB": st["d2h_bytes"] / 1e9,
            "input_GB": x.nbytes / 1e9, "H2D_GBps": st["h2d_GBps"], "D2H_GBps": st["d2h_GBps"],
            "h2d_ms": st["h2d_ms"], "d2h_ms": st["d2h_ms"], "kernel_ms": st["kernel_ms"],
            "kernel_ms_per_chunk": st["kernel_ms"] / nchunk, "d2d_GB": st["d2d_bytes"] / 1e9,
            "overlap_efficiency": st["overlap_efficiency"],
            "valid_Msamples_per_s": nchunk * hop * nchan * npol / (ms * 1e-3) / 1e6,
            "input_Msamples_per_s": total * nchan * npol / (ms * 1e-3) / 1e6}

if __name__ == "__main__":
    which = sys.argv[1:] or ["5", "4"]
    if "5" in which:
        print(json.dumps(config5()), flush=True)
    if "4" in which:
        print(json.dumps(config4()), flush=True)
    if "4full" in which:
        print(json.dumps(config4(28)), flush=True)
    if "4det" in which:
        print(json.dumps(config4(26, detect="I")), flush=True)
    if "4fulldet" in which:
        print(json.dumps(config4(28, detect="I")), flush=True)
