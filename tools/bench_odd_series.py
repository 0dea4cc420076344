"""Odd series counts (no 16-byte pairs per time sample for complex64): which layout kernels run and how fast."""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
for nchan, npol in ((3, 1), (5, 1), (7, 1), (3, 2), (9, 1), (15, 1)):
    b.run(24, nchan, npol, nchan_total=max(nchan, 8))
