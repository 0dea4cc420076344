import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
b.run(24, 8, 2)
b.run(26, 2, 2, nchan_total=8)
b.run(25, 8, 2)
b.run(27, 1, 2, nchan_total=8)
