#!/usr/bin/env python3
"""Instruction mix of one kernel in a disassembly made by tools/disasm.sh.
usage: tools/isa_count.py <file.s> <substring of the mangled kernel name> [--loop]
Counts are static (the persistent kernels' loop bodies are fully unrolled, so static counts of the body = instructions
per tile); --loop restricts them to the largest backward-branch loop."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    loop_only = "--loop" in sys.argv
    lines, on = [], False
    for ln in open(path):
        if ln.startswith("0000") and ln.rstrip().endswith(">:"):
            on = key in ln
            if on:
                lines.append([])
            continue
        if on and lines and ln.startswith("\t"):
            lines[-1].append(ln)
    if not lines:
        sys.exit("kernel not found")
    for body in lines:
        ins = []
        for ln in body:
            m = re.match(r"\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        lo, hi = 0, len(ins)
        if loop_only:
            best = None
            for i, (addr, op, args) in enumerate(ins):
                if op.startswith("s_cbranch") or op == "s_branch":
                    mm = re.match(r"(\d+)", args)
                    if mm:
                        off = int(mm.group(1))
                        if off >= 32768:
                            tgt = addr + 4 + 4 * (off - 65536)
                            j = next((k for k, x in enumerate(ins) if x[0] == tgt), None)
                            if j is not None and (best is None or i - j > best[1] - best[0]):
                                best = (j, i)
            if best:
                lo, hi = best
        c = collections.Counter(op for _, op, _ in ins[lo:hi])
        fp = {k: v for k, v in c.items() if re.match(r"v_(add|sub|mul|fma|fmac|fmamk|fmaak|mad|pk_|cos|sin|subrev|rcp|fract|rndne|cvt|ldexp|max|min)", k)}
        print(f"instructions {hi - lo}  (of {len(ins)})")
        def tot(pred):
            return sum(v for k, v in c.items() if pred(k))
        print("  VALU total      ", tot(lambda k: k.startswith("v_")))
        print("  FP              ", sum(fp.values()), dict(sorted(fp.items(), key=lambda kv: -kv[1])[:14]))
        print("  v_mov / v_accvgpr", tot(lambda k: k.startswith("v_mov") or k.startswith("v_accvgpr")))
        print("  ds_*            ", {k: v for k, v in c.items() if k.startswith("ds_")})
        print("  buffer/global   ", {k: v for k, v in c.items() if k.startswith("buffer_") or k.startswith("global_") or k.startswith("scratch_")})
        print("  s_waitcnt       ", c.get("s_waitcnt", 0), " s_barrier", c.get("s_barrier", 0), " s_nop", c.get("s_nop", 0))


if __name__ == "__main__":
    main()
