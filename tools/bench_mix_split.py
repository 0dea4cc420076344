"""Two-level 7-smooth plans (N1 = P x Q), a child process per length and setting:
  split   the most balanced P x Q (forced through PBH_MIX_Q) against the library's choice (largest Q <= 512 with P <= 512,
          else largest Q <= 1024);
  radixp  lengths whose P is 3, 5 or 7: the P-point stage as a k_colmix tile (PBH_MIX_RADIXP=0) against the elementwise
          k_radix_p (default).
usage: python tools/bench_mix_split.py split|radixp [count] [seed]"""
import json, os, subprocess, sys
import numpy as np

def smooth_odd(limit):
    out = []
    a = 1
    while a <= limit:
        b = a
        while b <= limit:
            c = b
            while c <= limit:
                out.append(c)
                c *= 7
            b *= 5
        a *= 3
    return sorted(out)

def old_q(n1):
    for d in range(512, 1, -1):
        if n1 % d == 0 and n1 // d <= 512:
            return d
    for d in range(1024, 1, -1):
        if n1 % d == 0 and n1 // d <= 1024:
            return d
    return 0

def new_q(n1):
    for d in range(2, 1025):
        if n1 % d == 0 and d * d >= n1 and n1 // d <= 1024:
            return d
    return 0

def run(n, q, extra=None):
    env = dict(os.environ)
    env.update(extra or {})
    if q:
        env["PBH_MIX_Q"] = str(q)
    r = subprocess.run([sys.executable, "tools/bench_smooth.py", str(n)], env=env, capture_output=True, text=True, timeout=120)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    return None

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "radixp"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    cands = []
    for m in smooth_odd(1 << 20):
        for k in range(5, 15):
            n = m << k
            if not ((1 << 22) <= n <= (1 << 25) and m > 1024 and old_q(m)):
                continue
            if what == "split" and old_q(m) != new_q(m):
                cands.append((n, m, k))
            if what == "radixp" and m // old_q(m) in (3, 5, 7):
                cands.append((n, m, k))
    print(f"{what}: {len(cands)} two-level lengths in [2^22, 2^25] concerned", flush=True)
    gains = []
    for i in rng.choice(len(cands), size=min(count, len(cands)), replace=False):
        n, m, k = cands[i]
        if what == "split":
            a, b = run(n, 0), run(n, new_q(m))
            la, lb = f"{m // old_q(m)} x {old_q(m)}", f"{m // new_q(m)} x {new_q(m)}"
        else:
            a, b = run(n, 0, {"PBH_MIX_RADIXP": "0"}), run(n, 0)
            la, lb = f"{m // old_q(m)} x {old_q(m)}, tile kernel", "k_radix_p"
        if a and b:
            gains.append(a["ms"] / b["ms"])
            print(f"{n:9d} = 2^{k} * {m}: {la}: {a['ms']:.3f} ms {a['Gsamples_per_s']:.1f} Gsamples/s  ->  {lb}: {b['ms']:.3f} ms "
                  f"{b['Gsamples_per_s']:.1f}   (stage: {a['kernels'].get('k_radix_fwd')} / {a['kernels'].get('k_radix_inv')} -> "
                  f"{b['kernels'].get('k_radix_fwd')} / {b['kernels'].get('k_radix_inv')} ms)", flush=True)
    if gains:
        print(f"speed-up: min {min(gains):.2f}, median {float(np.median(gains)):.2f}, max {max(gains):.2f}")
