"""The committed measurement evidence is self-consistent: the rocprofv3 kernel-trace average of the dominant kernel
agrees with the duration bench.py measured for it, and the PMC traffic is close to the algorithmic bytes."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def test_rocprof_agrees_with_bench_line():
    bench = json.loads(open(os.path.join(P, "r02z3_bench.json")).read().strip().splitlines()[-1])
    roof = bench["roofline"]
    assert roof["kernel"] == "k_row_fused"
    rows = list(csv.DictReader(open(os.path.join(P, "r02z3_planar5_kernel_stats.csv"))))
    row = [r for r in rows if "k_rowp16<" in r["Name"]]
    assert len(row) == 1
    avg_ms = float(row[0]["AverageNs"]) / 1e6
    assert abs(avg_ms - roof["ms_per_launch"]) / roof["ms_per_launch"] < 0.10
    # HBM traffic from the counters vs the algorithmic bytes of the launch (the phase trick moves fewer)
    traffic = json.load(open(os.path.join(P, "traffic.json")))
    assert 0.8 < traffic["k_row_fused"] / roof["alg_bytes_per_launch"] < 1.1
    assert abs(roof["traffic"] - traffic["k_row_fused"]) / traffic["k_row_fused"] < 0.02
    # the five kernels of the step add up to the step
    assert abs(bench["path_roofline"]["kernel_ms_total"] - bench["ms_per_step"]) / bench["ms_per_step"] < 0.05
