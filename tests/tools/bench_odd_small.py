"""m * 2^k lengths below the native range (2^k < 2^19): cost per sample through the padded convolution."""
import sys
sys.path.insert(0, "tests/tools"); sys.path.insert(0, ".")
import bench_arbitrary as b
for n in (3 << 18, 5 << 17, 3 << 16, 7 << 15, 1 << 19, 1 << 18):
    b.run(n)
