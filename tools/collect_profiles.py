#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/ into profiles/ (tracked): the bench line,
the rocprofv3 kernel statistics, the raw PMC CSVs and their per-kernel summary."""
import collections
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


def avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if m:
                agg[m.group(1)].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, f"bench_{tag}.json"))
# timed workload only (python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras) and the same with the extras:
# tools/profile_round.sh already reduced the trace databases to the --stats tables (tools/rocpd_stats.py) on the GPU box
for src, name in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("kernel_stats_with_extras.csv", f"{tag}_kernel_stats_with_extras.csv")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, name))
shutil.copy(os.path.join(G, "prof_bench.json"), os.path.join(P, f"bench_{tag}_under_rocprofv3.json"))
os.makedirs(os.path.join(P, f"{tag}_pmc"), exist_ok=True)
wcsv, fcsv = os.path.join(G, "pmc_w", "write_size_counter_collection.csv"), os.path.join(G, "pmc_f", "fetch_size_counter_collection.csv")
shutil.copy(wcsv, os.path.join(P, f"{tag}_pmc"))
shutil.copy(fcsv, os.path.join(P, f"{tag}_pmc"))
w, f = avg(wcsv, "WRITE_SIZE"), avg(fcsv, "FETCH_SIZE")
out = {
    "source": f"rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, raw CSVs in profiles/{tag}_pmc/), python3 bench.py "
              "--steps 3 --warmup 1 --no-cpu-baseline, MI355X; averages over the red6 and the standard-alphabet steps of the bench",
    "units": "counter values are KiB per dispatch; bytes = value*1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
             "half of wide streaming reads; for the random 8-byte reads of k_gram_sparse and k_basis_scatter the factor is "
             "uncalibrated: raw and doubled values both given)",
}
for k in ("k_cosine_write", "k_cosine_heavy", "k_gram_sparse", "k_basis_scatter", "k_basis_scatter_fused", "k_compact_rows", "k_head_count", "k_count_short"):
    if k not in w or k not in f:
        continue
    out[k] = {"WRITE_SIZE_KiB": w[k][0], "FETCH_SIZE_KiB_raw": f[k][0], "launches_sampled": w[k][1], "write_bytes": w[k][0] * 1024,
              "fetch_bytes_raw": f[k][0] * 1024, "fetch_bytes_doubled": 2 * f[k][0] * 1024}
out["k_cosine_write_bytes_per_launch"] = out["k_cosine_write"]["write_bytes"] + out["k_cosine_write"]["fetch_bytes_doubled"]
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only for the hash of the kernel sources this summary belongs to)

out["source_sha16"] = bench.kernel_source_sha()
out["note"] = f"rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE (separate passes), profiles/{tag}_pmc/, kernel sources {out['source_sha16']}"
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
b = json.load(open(os.path.join(P, f"bench_{tag}.json")))
print(f"{b['ms_per_step']:.3f} ms/step  {b['value']:.4g} {b['unit']}  roofline frac {b['roofline']['frac']:.3f}")
