"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (build.log) per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
keys = [('vgpr', r'VGPRs'), ('spill', r'VGPR Spill'), ('sgpr', r'SGPRs'),
        ('scratch', r'ScratchSize \[bytes/lane\]'), ('occ', r'Occupancy \[waves/SIMD\]')]
for b in blocks:
    name = b.split('\n')[0].strip()
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'^void pbh::', '', dn).split('(')[0]
    vals = []
    for k, pat in keys:
        m = re.search(pat + r': (\d+)', b)
        vals.append(f"{k}={m.group(1) if m else '?'}")
    print(f"{dn:32s} " + ' '.join(vals))
