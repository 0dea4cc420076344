"""Which route the 7-smooth lengths between 2^20 and 2^25 take (plan geometry as the library reports it)."""
import sys, collections
sys.path.insert(0, ".")
from pulsarbat_amd import _hip
from pulsarbat_amd.utils import _smooth_7

tally = collections.Counter()
for n in [v for v in _smooth_7(1 << 25) if v >= 1 << 20]:
    p = _hip.Plan(n, 1, 1, 0, n, device=0)
    i = p.info
    p.close()
    n1, n2 = i["n1"], i["n2"]
    if n & (n - 1) == 0:
        kind = "power of two"
    elif n1 == 1:
        kind = "padded convolution"
    elif n2 & (n2 - 1):
        kind = "mixed-radix rows and columns (k_rowmix + k_colmix)"
    elif any(n == m << k for m in (3, 5, 7) for k in range(19, 25)):
        kind = "m * 2^k (radix-m stage in the layout passes)"
    else:
        kind = "mixed-radix columns, one level" if n1 <= 1024 else "mixed-radix columns, two levels"
    tally[kind] += 1
total = sum(tally.values())
for k, v in tally.most_common():
    print(f"{v:5d}  {100 * v / total:5.1f} %  {k}")
print(f"{total:5d}  7-smooth lengths in [2^20, 2^25]")
