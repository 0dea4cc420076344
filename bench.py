#!/usr/bin/env python3
"""bench.py -- headline benchmark of the coherent-dedispersion hot path on MI355X.

Metric (BASELINE.json): complex Msamples/s dedispersed, where one sample is one complex64
element of the input (nsample, nchan, npol) block (samples later cropped are counted), with
the HBM roofline of the dominant kernel beside it.

Workload: BASELINE.json configs[1] at N=1: 2^24 samples x 8 chan x 2 pol complex64, DM 56.77,
400 MHz at 1.4 GHz.  At N>1 the same 400 MHz band is cut into 8*N channels (N=8 is
configs[2]: 64 channels of 6.25 MHz); every rank holds 8 channels x 2 pol x 2^24 samples of
it, so per-GPU work is fixed ("weak").  Channels are independent, so there is no data-path
collective: ranks only meet at the timing barriers.

One "step" = one call of the product's sharded entry point,
pulsarbat_amd.shard.coherent_dedispersion_sharded (FFT -> chirp -> IFFT -> crop on this rank's
channels with the full band's crop; at N=1 the rank holds the whole band), on a block already
resident in HBM, output left resident and sharded.  Plan and chirp are cached per geometry after the first call;
chirp generation is timed separately (reported as chirp_ms), and so is the plan's first call (plan_first_call_ms: one-time
buffer placement, part of plan set-up).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu]
  N>1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P bench.py --gpus N --steps K --warmup W
"""

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy
NSAMPLE = 1 << 24
NCHAN_PER_GPU = 8
NPOL = 2
DM = 56.77
BAND_HZ = 400e6
CENTER_HZ = 1.4e9


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline():
    """Oracle (numpy + scipy.fft restatement of the reference) on a bounded sample.

    Sample: ONE full workload block, (2^24, 8, 2) complex64, chirp precomputed (timed separately),
    scipy.fft workers=None (one thread: what the reference runs, pulsarbat/fft.py:36-38); about
    10-20 s of CPU work on the GPU node's host.  A second figure uses all host cores (not reference
    behaviour).
    """
    from oracle import dedisp_oracle as orc
    n = 1 << 24
    shape = (n, NCHAN_PER_GPU, NPOL)
    x = orc.synthetic_block(shape, 20260002)
    sr = BAND_HZ / NCHAN_PER_GPU
    t0 = time.perf_counter()
    chirp = orc.chirp_from_signal(DM, shape, sr, CENTER_HZ)
    t_chirp = time.perf_counter() - t0
    # warm-up on a 2^22-sample slice of the same block (pocketfft plan caches, page faults of the allocator), SURVEY.md 8d
    orc.coherent_dedispersion(x[:1 << 22], DM, sr, CENTER_HZ, chirp=chirp[:1 << 22])
    t0 = time.perf_counter()
    orc.coherent_dedispersion(x, DM, sr, CENTER_HZ, chirp=chirp)
    t1 = time.perf_counter() - t0
    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    orc.coherent_dedispersion(x, DM, sr, CENTER_HZ, chirp=chirp, workers=ncores)
    tall = time.perf_counter() - t0
    # (B) dask-equivalent: one task per (chan, pol) series on a thread pool, each a
    # single-threaded scipy.fft -- what rechunk() + the threaded scheduler give the reference
    from concurrent.futures import ThreadPoolExecutor
    import scipy.fft as _sfft
    nthreads = min(ncores, NCHAN_PER_GPU * NPOL)
    start, stop = orc.crop_bounds(DM, n, NCHAN_PER_GPU, sr, CENTER_HZ, CENTER_HZ)

    def one(cp):
        c, p = cp
        return _sfft.ifft(_sfft.fft(x[:, c, p]) * chirp[:, c].reshape(-1))[start:stop]

    t0 = time.perf_counter()
    try:
        with ThreadPoolExecutor(nthreads) as ex:
            list(ex.map(one, [(c, p) for c in range(NCHAN_PER_GPU) for p in range(NPOL)]))
        tpool = time.perf_counter() - t0
    except Exception:
        tpool = None
    nsamp = float(np.prod(shape))
    return {
        "value": nsamp / t1 / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": "oracle (numpy+scipy.fft, workers=None) on one full (2^24, 8, 2) c64 block, "
                  "DM 56.77, chirp precomputed, after a warm-up call on 2^22 samples; %.2f s" % t1,
        "all_cores": {"value": nsamp / tall / 1e6, "cores": ncores, "seconds": tall},
        "series_thread_pool": None if tpool is None else {"value": nsamp / tpool / 1e6, "cores": nthreads, "seconds": tpool},
        "chirp_seconds": t_chirp, "host_cpus": ncores,
    }


def extra_configs4_share(steps=10):
    """BASELINE configs[4], this GPU's share: 8 of the 64 channels of 6.25 MHz x 2 pol x 2^24 samples, DM 1000, fused
    dedispersion + Stokes-I + 1024x time scrunch (60 B/sample algorithmic, the last pass writes ~0)."""
    import torch
    from pulsarbat_amd import _hip
    from pulsarbat_amd.device import DeviceArray
    import pulsarbat_amd as pb
    from pulsarbat_amd import units as u
    n, nchan_tot, nchan, npol, dmv = 1 << 24, 64, 8, 2, 1000.0
    sr = BAND_HZ / nchan_tot
    dm = pb.DM(dmv)
    top = dm.sample_delay((CENTER_HZ + BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    bot = dm.sample_delay((CENTER_HZ - BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    freqs = (CENTER_HZ + sr * (np.arange(nchan_tot) + 0.5 - nchan_tot / 2))[:nchan]
    g = torch.Generator(device="cuda").manual_seed(20260004)
    x = DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), generator=g, device="cuda") * 0.7071))
    with _hip.Plan(n, nchan, npol, start, stop) as plan:
        plan.chirp_generate(dmv / 2.41e-4 * 1e12, 1 / sr, freqs, CENTER_HZ)
        out = DeviceArray.empty(((stop - start) // 1024, nchan), np.float32)
        for _ in range(3):
            plan.dedisperse_detect(x, nscrunch=1024, mode="I", out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.dedisperse_detect(x, nscrunch=1024, mode="I", out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    ns = float(n) * nchan * npol
    return {"workload": "configs[4] per-GPU share: 2^24 x 8 of 64 chan x 2 pol, DM 1000, Stokes-I + 1024x scrunch, fused tail",
            "ms_per_step": ms, "value": ns / ms / 1e3, "unit": "Msamples/s", "crop": [start, stop], "out_shape": list(out.shape),
            "alg_bytes_per_sample": 60.0, "achieved_GBps": 60.0 * ns / ms / 1e6, "frac_of_8TBps": 60.0 * ns / ms / 1e6 / HBM_PEAK_GBPS}


def extra_configs3_stream(total_log2=26):
    """BASELINE configs[3], bounded: 2^26 samples x 8 x 2 (the full config is 2^28) streamed from pinned host memory in
    2^22-sample chunks, overlap-save with every input row uploaded once (pbh_dedisperse_stream)."""
    import torch
    from pulsarbat_amd import _hip
    import pulsarbat_amd as pb
    from pulsarbat_amd import units as u
    nchan, npol, n, total = NCHAN_PER_GPU, NPOL, 1 << 22, 1 << total_log2
    sr = BAND_HZ / nchan
    dm = pb.DM(DM)
    top = dm.sample_delay((CENTER_HZ + BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    bot = dm.sample_delay((CENTER_HZ - BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    start, stop = math.ceil(-min(0, top, bot)), n - math.ceil(max(0, top, bot))
    hop = stop - start
    freqs = CENTER_HZ + sr * (np.arange(nchan) + 0.5 - nchan / 2)
    blk = min(total, 1 << 24)   # one random block repeated with a different complex factor (randn of 8 GB takes a minute)
    base = torch.randn((blk, nchan, npol, 2), dtype=torch.float32).numpy().view(np.complex64).reshape(blk, nchan, npol)
    x = np.empty((total, nchan, npol), np.complex64)
    for k in range(total // blk):
        np.multiply(base, np.complex64(np.exp(0.37j * k) * (1 + 0.01 * k)), out=x[k * blk:(k + 1) * blk])
    out = np.empty((((total - n) // hop + 1) * hop, nchan, npol), np.complex64)
    out[::4096] = 0
    with _hip.Plan(n, nchan, npol, start, stop) as plan:
        plan.chirp_generate(DM / 2.41e-4 * 1e12, 1 / sr, freqs, CENTER_HZ)
        y, ms = plan.dedisperse_stream(x, out=out)
        st = plan.stream_stats()
    # the same stream as a filterbank stream: configs[4]'s tail (Stokes I, 1024x) at the end of every chunk
    fb = None
    try:
        ns = 1024
        stop_fb = stop - (stop - start) % ns
        with _hip.Plan(n, nchan, npol, start, stop_fb) as plan:
            plan.chirp_generate(DM / 2.41e-4 * 1e12, 1 / sr, freqs, CENTER_HZ)
            plan.stream_detect("I", ns)
            yd, ms_fb = plan.dedisperse_stream(x)
            sf = plan.stream_stats()
        fb = {"workload": "the same stream with Stokes-I + 1024x scrunch inside every chunk (pbh_plan_stream_detect): float32 rows come back",
              "ms_total": ms_fb, "h2d_GBps": sf["h2d_GBps"], "d2h_GB": sf["d2h_bytes"] / 1e9, "kernel_ms": sf["kernel_ms"],
              "overlap_efficiency": sf["overlap_efficiency"], "out_shape": list(yd.shape),
              "input_Msamples_per_s": float(total) * nchan * npol / ms_fb / 1e3}
    except Exception as exc:   # an extra of an extra
        fb = {"error": repr(exc)}
    return {"filterbank_stream": fb, "workload": ("configs[3] FULL SIZE" if total_log2 == 28 else "configs[3] bounded") +
                        ": 2^%d samples x 8 x 2 in 2^22-sample chunks, hop %d, %d chunks (full config: 2^28, 430 chunks)" % (total_log2, hop, st["nchunk"]),
            "ms_total": ms, "input_GB": x.nbytes / 1e9, "h2d_GB": st["h2d_bytes"] / 1e9, "d2h_GB": st["d2h_bytes"] / 1e9,
            "h2d_GBps": st["h2d_GBps"], "d2h_GBps": st["d2h_GBps"], "kernel_ms": st["kernel_ms"],
            "overlap_efficiency": st["overlap_efficiency"],
            "input_Msamples_per_s": float(total) * nchan * npol / ms / 1e3,
            "valid_Msamples_per_s": float(len(y)) * nchan * npol / ms / 1e3,
            "note": "PCIe-bound (H2D and D2H of equal size run concurrently); never the headline value"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-series", action="store_true", help="skip the series-major secondary figure")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4] extras (after the timed region)")
    ap.add_argument("--gather", nargs="?", const="all", default=None, choices=["all", "root"],
                    help="(multi-rank) also time one step that ends with the gather of the outputs by direct peer writes: "
                         "every rank gets the full band (all) or rank 0 does (root)")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--log2n", type=int, default=24, help="(debug) nsample = 2^log2n")
    ap.add_argument("--dm", type=float, default=DM, help="(debug) dispersion measure")
    args = ap.parse_args()

    # keep stdout to the one JSON line: RCCL (a C library) prints its banner/warnings to fd 1, so
    # fd 1 is pointed at stderr for the run and the JSON goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    t_start = time.perf_counter()
    if os.environ.get("PBH_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack if the run is still alive after N s
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PBH_BENCH_WATCHDOG"]), exit=True)

    def mark(what):  # wall-clock trail on stderr: where a slow launch spends its time (imports, RCCL init, ...)
        log(f"[bench +{time.perf_counter() - t_start:7.1f}s] {what}")

    import torch
    import torch.distributed as dist
    mark("torch imported")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks")
            sys.exit(2)
    distributed = world > 1 or "RANK" in os.environ  # under torchrun: exercise RCCL even with one rank
    # PBH_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("PBH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        mark("process group ready")

    from pulsarbat_amd import _hip
    from pulsarbat_amd.device import DeviceArray
    import pulsarbat_amd as pb
    from pulsarbat_amd import units as u

    nsample = 1 << args.log2n
    nchan_total = NCHAN_PER_GPU * world
    sr = BAND_HZ / nchan_total
    # geometry through the product's own host code (RadioSignal.channel_freqs / crop rule)
    freqs_all = CENTER_HZ + sr * (np.arange(nchan_total) + 0.5 - nchan_total / 2)
    freqs = freqs_all[rank * NCHAN_PER_GPU:(rank + 1) * NCHAN_PER_GPU]
    dm = pb.DM(args.dm)
    d_top = dm.sample_delay((CENTER_HZ + BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    d_bot = dm.sample_delay((CENTER_HZ - BAND_HZ / 2) * u.Hz, CENTER_HZ * u.Hz, sr * u.Hz)
    start = math.ceil(-min(0, d_top, d_bot))
    stop = nsample - math.ceil(max(0, d_top, d_bot))
    coeff = args.dm / 2.41e-4 * 1e12

    gen = torch.Generator(device="cuda")
    gen.manual_seed(20260002 + rank)
    x = torch.randn((nsample, NCHAN_PER_GPU, NPOL, 2), generator=gen, device="cuda",
                    dtype=torch.float32) * (2 ** -0.5)
    x = DeviceArray(torch.view_as_complex(x))
    # this rank's shard as the product sees it: a DualPolarizationSignal of its 8 channels
    from pulsarbat_amd import shard
    from pulsarbat_amd.transforms.dedispersion import _plan_for
    z_local = pb.DualPolarizationSignal(x, sample_rate=sr * u.Hz, center_freq=float(np.mean(freqs)) * u.Hz,
                                        pol_type="linear")
    band = dict(band_min=(CENTER_HZ - BAND_HZ / 2) * u.Hz, band_max=(CENTER_HZ + BAND_HZ / 2) * u.Hz,
                ref_freq=CENTER_HZ * u.Hz)

    def step():
        return shard.coherent_dedispersion_sharded(z_local, dm, variant=args.variant, **band)

    # the plan the entry point uses (its per-thread cache): for the chirp timing, the per-kernel profile and the info
    plan, _ = _plan_for(z_local, dm, band["ref_freq"], (start, stop), variant=args.variant, device=local_rank)
    assert (plan.crop_start, plan.crop_stop) == (start, max(stop, start))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.chirp_generate(coeff, 1.0 / sr, freqs, CENTER_HZ)
    torch.cuda.synchronize()
    chirp_ms = (time.perf_counter() - t0) * 1e3
    # plan set-up ends with ONE priming call: a plan's first call looks for its second work buffer among fresh allocations and
    # times its four buffer-role assignments (one-time, 0.05-0.3 s; DESIGN.md 6d d).  That belongs to plan creation, not to the
    # W warm-up steps -- with --warmup 0 it would otherwise land in the timed region.
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    prime_ms = (time.perf_counter() - t0) * 1e3
    mark("plan + chirp ready")

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    y = None
    for _ in range(args.warmup):
        y = step()
    barrier()
    mark("warm-up done (first barrier includes RCCL communicator set-up)")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    barrier()
    elapsed = time.perf_counter() - t0
    mark("timed region done")
    if distributed:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same step timed with HIP events on the stream the kernels run on, one pair per step: median of >= 20
    ev_n = max(20, args.steps)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ev_n)]
    for a, b in evs:
        a.record()
        y = step()
        b.record()
    torch.cuda.synchronize()
    ev_ms = sorted(a.elapsed_time(b) for a, b in evs)
    event_median_ms = ev_ms[len(ev_ms) // 2]
    y = y.data

    # optional: the one real exchange step of the sharded path, an all-gather of the outputs along the channel
    # axis (results are left sharded by default; the gather is timed separately)
    gather_ms, gather_form = None, None
    if distributed and args.gather:
        try:
            from pulsarbat_amd.node import ChannelGather
            mark("gather: setting up the destination blocks and their mappings")
            g = ChannelGather(plan.nout, NCHAN_PER_GPU, NPOL, np.complex64, local_rank, mode=args.gather)
            mark("gather: first run")
            g.run(plan, x)
            barrier()
            mark("gather: timed run")
            t1 = time.perf_counter()
            g.run(plan, x)     # dedispersion + delivery of the slice to the destination block(s) + closing collective
            gather_ms = (time.perf_counter() - t1) * 1e3
            gather_form = "one contiguous shared buffer per destination" if g.shared else "hipIpc row-chunks + join"
            g.close()
            del g
        except Exception as exc:   # never lose the main line to the optional figure
            gather_ms = repr(exc)

    # secondary figure: the same block kept series-major (time fastest) in HBM at both ends, as a
    # device-resident pipeline would keep it -- the two layout passes disappear (3 kernels)
    series_major = None
    if rank == 0 and plan.supports_series_major and not args.no_series:
        xs = x.to_series_major()
        ys = plan.dedisperse(xs)
        for _ in range(args.warmup):
            plan.dedisperse(xs, out=ys)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            plan.dedisperse(xs, out=ys)
        torch.cuda.synchronize()
        sm_ms = (time.perf_counter() - t1) / args.steps * 1e3
        series_major = {"ms_per_step": sm_ms, "value": float(nsample) * NCHAN_PER_GPU * NPOL / sm_ms / 1e3,
                        "unit": "Msamples/s", "note": "input and output stored series-major (time fastest) on the "
                        "device: k_col_fwd reads the input and k_col_inv writes the cropped output directly"}
        del xs, ys
    if distributed:
        dist.barrier()

    # per-kernel HIP-event timing on the plan's stream (same launches as the timed region)
    kern = plan.profile(x, y, iters=max(3, min(args.steps, 10)))
    info = plan.info
    samples_gpu = float(nsample) * NCHAN_PER_GPU * NPOL
    # algorithmic bytes per sample of each kernel (DESIGN.md "Kernels"): reads + writes it cannot avoid
    crop_frac = plan.nout / nsample
    alg = {"k_col_fwd": 16.0, "k_row_fused": 16.0 + 8.0 / NPOL, "k_col_inv": 8.0 + 8.0 * crop_frac,
           "k_deinterleave": 16.0, "k_reinterleave": 16.0 * crop_frac, "k_small": 8.0 + 8.0 * crop_frac + 8.0 / NPOL}
    dom_name, dom_ms = max(kern, key=lambda kv: kv[1])
    dom_bytes = alg.get(dom_name, 16.0) * samples_gpu   # an unlisted pass (k_radix_*, k_pad ...) moves 8 r + 8 w
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic, prof_us = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dom_name)
            prof_us = (tj.get("_kernel_us") or {}).get(dom_name)
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": "stored: profiles/traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                  "command (tools/prof.sh); not collected in this run",
                "alg_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms,
                # the same fraction from the TRACKED rocprofv3 summary (profiles/traffic.json "_kernel_us": mean duration of
                # this kernel in the kernel-trace pass of tools/prof.sh): reproducible from the committed files alone
                "frac_rocprof": None if not prof_us else dom_bytes / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "rocprof_us_per_launch": prof_us}
    total_kernel_ms = sum(ms for _, ms in kern)
    path_bytes = info["alg_bytes_per_sample"] * samples_gpu  # 68 B/sample accounting figure
    # what the five passes really move, and what that costs at the chip's own copy rate, measured now
    moved = {"k_deinterleave": 16.0, "k_col_fwd": 16.0, "k_row_fused": 16.0 + 4.0 / NPOL, "k_col_inv": 8.0 + 8.0 * crop_frac,
             "k_reinterleave": 16.0 * crop_frac, "k_small": 8.0 + 8.0 * crop_frac + 8.0 / NPOL}
    # optional figures: a plan with a pass this table does not know (k_radix_*, k_pad, k_bs_* at other sizes) loses the floor, not the line
    known = all(k in moved for k, _ in kern)
    moved_bytes = sum(moved[k] for k, _ in kern) * samples_gpu if known else None
    path = {"alg_bytes_per_sample": info["alg_bytes_per_sample"],
            "achieved": path_bytes / (total_kernel_ms * 1e-3) / 1e9, "unit": "GB/s",
            "frac": path_bytes / (total_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "kernel_ms": {k: round(v, 4) for k, v in kern}, "kernel_ms_total": total_kernel_ms,
            "moved_bytes_per_step": moved_bytes}
    try:
        # the chip's own streaming rates, measured now (pbh_stream_bench, 2 GiB): a copy between two buffers, and a
        # read-modify-write of one buffer in place (what the three middle passes do to the planar work buffer)
        copy_ms = _hip.stream_bench(1 << 31, iters=10, device=local_rank, mode="copy")
        rmw_ms = _hip.stream_bench(1 << 31, iters=10, device=local_rank, mode="rmw")
        copy_gbps = 2.0 * (1 << 31) / copy_ms / 1e6
        rmw_gbps = 2.0 * (1 << 31) / rmw_ms / 1e6
        path.update({"copy_ceiling_GBps": copy_gbps, "rmw_ceiling_GBps": rmw_gbps})
        if known:
            inplace = {"k_col_fwd", "k_row_fused", "k_col_inv"}
            b_in = sum(moved[k] for k, _ in kern if k in inplace) * samples_gpu
            path.update({"floor_ms": moved_bytes / copy_gbps / 1e6,
                         "floor_ms_rmw": (moved_bytes - b_in) / copy_gbps / 1e6 + b_in / rmw_gbps / 1e6,
                         "floor_note": "floor_ms: bytes the passes move / the float4 copy rate of this run; floor_ms_rmw: the "
                                       "in-place passes' bytes priced at the in-place read-modify-write rate instead"})
    except Exception as exc:
        path["ceiling_error"] = repr(exc)

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_gpu * world * args.steps / elapsed / 1e6
        result = {
            "metric": "complex Msamples/s dedispersed (2^24x8chx2pol c64); HBM GB/s vs peak",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "c64", "data": "synthetic",
            "config": {"workload": "configs[1]: 2^%d samples x 8 chan x 2 pol complex64 per GPU, DM=56.77, "
                                   "400 MHz band @ 1.4 GHz split into %d channels; device-resident in/out"
                                   % (args.log2n, nchan_total),
                       "nsample": nsample, "nchan_per_gpu": NCHAN_PER_GPU, "npol": NPOL,
                       "nchan_total": nchan_total, "crop": [start, stop], "variant": info["variant"],
                       "n1": info["n1"], "n2": info["n2"], "sharding": "channels across ranks, no collective"},
            "roofline": roofline, "path_roofline": path, "chirp_ms": chirp_ms, "plan_first_call_ms": prime_ms,
            "ms_per_step_event_median": event_median_ms, "event_steps": ev_n,
            "step": "pulsarbat_amd.shard.coherent_dedispersion_sharded (cached plan + chirp), output left sharded",
        }
        if series_major is not None:
            result["series_major_io"] = series_major
        if gather_ms is not None:
            result["step_with_gather_ms"] = {"mode": args.gather, "ms": gather_ms, "destinations": gather_form}
        if world == 1 and not args.no_extras and args.log2n == 24:
            del x, y, z_local
            torch.cuda.empty_cache()
            extras = [("configs4_share", extra_configs4_share), ("configs3_stream", extra_configs3_stream)]
            # BASELINE configs[3] at FULL size (2^28 samples x 8 x 2 in 2^22-sample chunks = 430 chunks: 34.4 GB in, 33.9 GB out, both
            # page-locked for the call) when the host has the memory for it; the bounded 2^26 run above stays as the fallback
            try:
                import psutil
                avail = psutil.virtual_memory().available
            except Exception:
                avail = 0
            if avail >= 110e9:
                extras.append(("configs3_stream_full", lambda: extra_configs3_stream(28)))
            for key, fn in extras:
                try:
                    result[key] = fn()
                except Exception as exc:   # extras never cost the main line
                    result[key] = {"error": repr(exc)}
                mark(f"extra {key} done")
            if avail < 110e9:
                result["configs3_stream_full"] = {"skipped": "host memory available %.0f GB < 110 GB" % (avail / 1e9)}
        if world == 1 and not args.no_cpu:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # the baseline is a reported extra; never lose the GPU line
                result["cpu_baseline"] = {"error": repr(exc)}
        mark("extras done")
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
