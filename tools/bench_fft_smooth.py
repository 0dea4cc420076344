import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_fft as b
for n, bt in ((625 << 14, 16), (729 << 14, 16), (600 << 14, 16)):
    b.run(n, bt)
