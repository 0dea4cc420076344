"""One call each of the round-2 next-row paths (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs): incoherent dedispersion,
freq_shift, stft + coherent_dedispersion (two calls) and stft_dedisperse (fused), 2^24 x 8 x 2 complex64 on the device."""
import sys
sys.path.insert(0, ".")
import torch
import pulsarbat_amd as pb
from pulsarbat_amd import units as u

n, nchan, npol = 1 << 24, 8, 2
x = pb.DeviceArray(torch.view_as_complex(torch.randn((n, nchan, npol, 2), device="cuda")))
z = pb.DualPolarizationSignal(x, sample_rate=50 * u.MHz, center_freq=1.4 * u.GHz, pol_type="linear")
which = sys.argv[1]
reps = 3
for _ in range(reps):
    if which == "incoherent":
        pb.incoherent_dedispersion(z, pb.DM(56.77))
    elif which == "freq_shift":
        pb.freq_shift(z, 1 * u.MHz)
    elif which == "stft_two_calls":
        pb.coherent_dedispersion(pb.contrib.stft(z, nperseg=64), pb.DM(56.77))
    elif which == "stft_fused":
        pb.contrib.stft_dedisperse(z, pb.DM(56.77), nperseg=64)
torch.cuda.synchronize()
