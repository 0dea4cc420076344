#!/usr/bin/env python3
"""Scan the gfx950 code object of a built object file for a hazard hipcc (ROCm 7.2) does not guard:

  a buffer store of more than 64 bits per lane WITH an SGPR offset, directly followed by a VALU write of one of its data
  registers -- lanes 12-15 of every 16 then store the NEW value (found with complex128 stores in round 1, again with the
  16-byte complex64 pair stores of k_rowq16 in round 4; LLVM's hazard recognizer only covers the form without an soffset).

usage: tools/isa_hazards.py [pbhip32.o pbhip64.o ...]   (exit code 1 and one line per kernel if the pattern occurs)
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
STORE = re.compile(r"\s+buffer_store_dwordx([34]) v\[(\d+):(\d+)\], (\S+), s\[\d+:\d+\], (\S+)")
VALU = re.compile(r"\s+(v_\S+)\s+v\[?(\d+)(?::(\d+))?")
STOP = re.compile(r"\s+(s_nop|s_waitcnt|buffer_|ds_|global_|scratch_)")


def scan(path_s):
    lines = open(path_s).read().split("\n")
    heads = [i for i, l in enumerate(lines) if l.startswith("0000") and l.endswith(">:")] + [len(lines)]
    found = []
    for a, b in zip(heads[:-1], heads[1:]):
        body = lines[a:b]
        hits = n = 0
        for i, l in enumerate(body):
            m = STORE.match(l)
            if not m:
                continue
            n += 1
            lo, hi = int(m.group(2)), int(m.group(3))
            if not m.group(5).startswith("s"):
                continue   # no SGPR offset: the form LLVM guards
            for l2 in body[i + 1:i + 3]:
                m2 = VALU.match(l2)
                if m2:
                    d0 = int(m2.group(2))
                    d1 = int(m2.group(3)) if m2.group(3) else d0
                    if d0 <= hi and d1 >= lo:
                        hits += 1
                        break
                if STOP.match(l2):
                    break
        if hits:
            found.append((re.sub(r"^\S+ <", "", lines[a]).rstrip(">:"), hits, n))
    return found


def main():
    objs = sys.argv[1:] or ["pbhip32.o", "pbhip64.o", "pbhip32_measure.o"]
    bad = 0
    for o in objs:
        res = subprocess.run([os.path.join(HERE, "disasm.sh"), o], capture_output=True, text=True)
        if res.returncode != 0:     # a unit without device code (pbhip32_stream.o): nothing to scan
            continue
        out = res.stdout.strip().splitlines()[-1]
        for name, hits, n in scan(out):
            print(f"{o}: {name}: {hits} of {n} wide stores are followed by a VALU write of their data registers")
            bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
