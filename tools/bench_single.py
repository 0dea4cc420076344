"""One series (S = 1) at several lengths."""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
for lg in (20, 22, 24, 26):
    b.run(lg, 1, 1, dm=5.0 if lg < 24 else 56.77, nchan_total=8)
