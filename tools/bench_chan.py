import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_shapes as b
# channelised blocks: many narrow channels, short time axis (what stft -> coherent_dedispersion produces)
b.run(18, 512, 2, dm=56.77, nchan_total=512)
b.run(17, 1024, 2, dm=56.77, nchan_total=1024)
b.run(20, 128, 2, dm=56.77, nchan_total=128)
b.run(16, 4096, 1, dm=56.77, nchan_total=4096)
