#!/usr/bin/env python3
"""A/B of library switches on ONE box: fresh bench.py processes, the configurations alternating.
usage: tools/ab_env.py [--rounds R] [--bench-args "..."] CONFIG [CONFIG ...]
CONFIG is '-' (no switch) or NAME=VAL[+NAME=VAL...]; prints ms_per_step, the event median and the per-kernel times."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, bargs = 3, []
    while args and args[0].startswith("--"):
        if args[0] == "--rounds":
            rounds = int(args[1]); args = args[2:]
        elif args[0] == "--bench-args":
            bargs = args[1].replace("+", " ").split(); args = args[2:]   # items joined by + (or spaces)
        else:
            sys.exit("unknown option " + args[0])
    res = {c: [] for c in args}
    for r in range(rounds):
        for c in args:
            env = dict(os.environ)
            if c != "-":
                for kv in c.split("+"):
                    k, v = kv.split("=", 1)
                    env[k] = v
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu", "--no-series"] + bargs,
                                 env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
            if out.returncode != 0:
                print(c, "FAILED", out.stderr[-600:], flush=True)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            k = d["path_roofline"]["kernel_ms"]
            res[c].append(d["ms_per_step"])
            print(f"{c:34s} step {d['ms_per_step']:.4f}  ev {d['ms_per_step_event_median']:.4f}  " +
                  " ".join(f"{n[2:]}={v:.4f}" for n, v in k.items()), flush=True)
    for c, v in res.items():
        if v:
            print(f"== {c:34s} median {sorted(v)[len(v) // 2]:.4f}  min {min(v):.4f}  n={len(v)}")


if __name__ == "__main__":
    main()
