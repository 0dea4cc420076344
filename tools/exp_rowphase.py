"""A/B of the phase-fed row pass on short series (PBH_ROW_PHASE from the environment): per-kernel times of a (2^18, 512, 2) plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "tools")
import bench_shapes as b
for lg, nchan in ((18, 512), (16, 2048), (15, 4096)):
    b.run(lg, nchan, 2, dm=56.77 if lg >= 18 else 5.0, nchan_total=nchan)
