"""Summarise the k_row phase stamps of a -DPBH_DIAGNOSTIC build run with PBH_ROW_ABL=4 PBH_ROW_DBG=<file>."""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 64, 8).astype(np.int64)
last = 4 if (len(sys.argv) > 3 and sys.argv[3] == 'col') else 6
ok = (d[:, :, last] > 0)
names = ["wait x", "chirp issue + fwd FFT", "wait chirp", "mul + issue next x", "inv FFT", "issue stores", "loop tail"]
if len(sys.argv) > 3 and sys.argv[3] == "col":
    names = ["top: tables, (inv: twiddle mul)", "FFT (+ next loads, inv: stores)", "fwd: tables + twiddle + stores", "move next -> v", "-", "-", "loop tail"]
clk = float(sys.argv[2]) if len(sys.argv) > 2 else 100e6   # s_memtime counts at the constant 100 MHz clock on gfx9xx
print("blocks", d.shape[0], "iterations with stamps per block: median", int(np.median(ok.sum(1))))
ph = np.diff(d[:, :, :last + 1], axis=2)           # phases inside the iteration
tail = d[:, 1:, 0] - d[:, :-1, last]            # loop tail (move nx -> v, index arithmetic)
sel = ok[:, 1:] & ok[:, :-1]
sel[:, :2] = False                           # skip the warm-up iterations
tot = d[:, 1:, 0] - d[:, :-1, 0]
for i in range(last):
    x = ph[:, 1:, i][sel]
    print(f"{names[i]:24s} mean {x.mean() / clk * 1e6:7.2f} us   p10 {np.percentile(x, 10) / clk * 1e6:7.2f}   p90 {np.percentile(x, 90) / clk * 1e6:7.2f}")
x = tail[sel]
print(f"{names[6]:24s} mean {x.mean() / clk * 1e6:7.2f} us")
x = tot[sel]
print(f"{'iteration':24s} mean {x.mean() / clk * 1e6:7.2f} us   p10 {np.percentile(x, 10) / clk * 1e6:7.2f}   p90 {np.percentile(x, 90) / clk * 1e6:7.2f}")
