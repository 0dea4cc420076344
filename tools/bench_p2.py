"""N1 = 2 x Q plans (complex64 2^25): folded radix-2 split (default) vs unsplit (PBH_RADIX_FUSE=0), by series count."""
import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
for nchan, npol in ((1, 2), (2, 2), (4, 2), (8, 2), (16, 2), (32, 2)):
    b.run(25, nchan, npol, nchan_total=max(nchan, 8))
