import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
for lg, nchan in [(19, 64), (20, 32), (21, 16), (22, 16), (23, 8)]:
    b.run(lg, nchan, 2, dm=5.0, nchan_total=nchan)
