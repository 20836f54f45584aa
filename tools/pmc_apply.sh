#!/bin/bash
# PMC of the learn / apply chain's kernels (tools/bench_apply.py under rocprofv3, separate passes per counter as
# MI355X_MICROARCH.md prescribes); summary -> gpurun_out/r6/pmc_apply.txt (copied to profiles/r06_pmc/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_apply_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_apply_$c -o p -- python3 $R/tools/bench_apply.py > /dev/null 2> $O/pmc_apply_$c.err || exit 1
done
python3 - <<PY > $O/pmc_apply.txt
import collections, csv, glob, re
out = collections.defaultdict(dict)
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pmc_apply_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
                if m:
                    agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c] = (sum(v) / len(v) * 1024.0, len(v))
print("rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes) of tools/bench_apply.py: 100 k queries x 1000 family totals; bytes per dispatch")
print("(counter values are KiB; FETCH_SIZE raw - gfx950 reports half of wide streaming reads, random 8-16-byte gathers are uncalibrated)")
for k in ("k_apply_top2", "k_apply_columns", "k_apply_top2_dense", "k_gp_lists", "k_gp_long", "k_gp_compact", "k_pc_keys", "k_pc_gather", "k_pass", "k_cosine_strip"):
    if k in out:
        w, f = out[k].get("WRITE_SIZE", (0, 0)), out[k].get("FETCH_SIZE", (0, 0))
        print(f"{k:22s} written {w[0] / 1e9:8.4f} GB  fetched (raw) {f[0] / 1e9:8.4f} GB  fetched x2 {2 * f[0] / 1e9:8.4f} GB   dispatches sampled {w[1]}")
PY
cat $O/pmc_apply.txt
