import sys
sys.path.insert(0, "tests/tools"); sys.path.insert(0, ".")
import bench_arbitrary as b
for n in (3 << 22, 5 << 21, 7 << 21, 1 << 24, 10_000_000, 16_000_000, 5_000_000):
    b.run(n)
