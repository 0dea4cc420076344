"""Device-memory leak check: a mix of operations repeated, caches cleared, free memory compared with the start."""
import sys, gc
sys.path.insert(0, ".")
import numpy as np, torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u, _hip
from pulsarbat_amd import shard
from pulsarbat_amd.transforms.dedispersion import clear_plan_cache
import os, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
dist.init_process_group("gloo", rank=0, world_size=1)

def free():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]

rng = np.random.default_rng(0)
def one_round(k):
    n = [1 << 18, 3 << 19, 100003, 1 << 20, 625 << 7, 2025 << 5, 81000, 234375][k % 8]   # (the last four: 7-smooth -- one and two
                                                                                      #  column levels, mixed-radix rows, odd)
    x = (rng.standard_normal((n, 2, 2)) + 1j * rng.standard_normal((n, 2, 2))).astype(np.complex64)
    z = pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear").to_device()
    y = pb.coherent_dedispersion(z, pb.DM(2.0))
    pb.dedisperse_detect(z, pb.DM(2.0), mode="I", nscrunch=64)
    pb.coherent_dedispersion(type(z).like(z, z.data.to_series_major()), pb.DM(2.0)) if n in (1 << 18, 1 << 20) else None
    pb.time_shift(z, 3.3); pb.freq_shift(z, 10 * u.kHz); pb.incoherent_dedispersion(z, pb.DM(2.0)); z.to_stokes()
    pb.contrib.istft(pb.contrib.stft(z, nperseg=[256, 16384, 1 << 15, 1000][k % 4]), nperseg=[256, 16384, 1 << 15, 1000][k % 4])
    pb.fft.ifft(pb.fft.fft(z.data, axis=0), axis=0)
    pb.utils.real_to_complex(pb.DeviceArray.from_host(x.real.copy()), axis=0)
    raw = rng.integers(0, 256, n * 8, dtype=np.uint8)
    lay = dict(nbits=8, ncomp=2, code=0, blk_samples=n, blk_stride=raw.size, hdr_bytes=0, elem0=0, stride_t=4, stride_c=2, stride_p=1)
    _hip.decode(raw, lay, 0, n, 2, 2, conj=np.array([[1, 0], [0, 1]], bool))
    pb.coherent_dedispersion_stream(pb.DualPolarizationSignal(x, sample_rate=1 * u.MHz, center_freq=1 * u.GHz, pol_type="linear"),
                                    pb.DM(2.0), chunk=1 << 15)
    # round-2 entry points: fused channelise + dedisperse, user chirps (per polarisation), the sharded call with its gather
    if n in (1 << 18, 1 << 20):
        pb.contrib.stft_dedisperse(z, pb.DM(0.5), nperseg=[32, 64][k % 2])
    c = np.exp(2j * np.pi * rng.random((n, 2, 2))).astype(np.complex64)
    pb.coherent_dedispersion(z, pb.DM(2.0), chirp=c)
    zl = shard.shard_signal(z, 2, k % 2)
    shard.coherent_dedispersion_sharded(zl, pb.DM(2.0), band_min=z.min_freq, band_max=z.max_freq, ref_freq=z.center_freq,
                                        gather=[False, True, "root"][k % 3])

one_round(0); one_round(1); one_round(2); one_round(3)
clear_plan_cache()
base = free()
for k in range(24):
    one_round(k)
clear_plan_cache()
end = free()
print(f"free before {base / 2**20:.1f} MiB, after 24 rounds {end / 2**20:.1f} MiB, difference {(base - end) / 2**20:.2f} MiB")
sys.exit(1 if base - end > 64 * 2**20 else 0)
