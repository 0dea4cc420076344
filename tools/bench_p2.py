import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
b.run(25, 8, 2)
b.run(25, 2, 2, nchan_total=8)
