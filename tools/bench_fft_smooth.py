"""pb.fft.fft of 7-smooth lengths, (n, 16) complex64 on the device; PBH_MIXED=0 PBH_ROWMIX=0 gives the convolution ring."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_fft as b
for n, bt in ((625 << 14, 16), (729 << 14, 16), (600 << 14, 16), (10935000, 16), (13671875, 16)):
    b.run(n, bt)
