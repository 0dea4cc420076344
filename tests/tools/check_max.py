"""Largest supported block: 2^28 samples x 1 channel x 2 pols (P = 16 split, folded into the layout passes):
one series against scipy.fft on the host, plus timing."""
import sys
sys.path.insert(0, "tests/tools"); sys.path.insert(0, ".")
import bench_arbitrary as b
b.run(1 << 28, nchan=1, npol=2, dm=56.77, band=50e6, center=1.4e9, check=True)
b.run(1 << 27, nchan=2, npol=2, dm=56.77, band=100e6, center=1.4e9, check=True)
