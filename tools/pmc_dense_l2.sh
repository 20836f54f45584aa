#!/bin/bash
# L2 behaviour of the dense i8 MFMA cosine (tools/ab_dense.py runs variants 3, 5, 4 in one process): hit/miss counts and
# the bytes fetched from beyond L2, one counter group per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_l2a $O/pmc_l2b
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_l2a -o a -- python3 $R/tools/ab_dense.py hydro 14 32768 > $O/pmc_l2a.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_l2b -o b -- python3 $R/tools/ab_dense.py hydro 14 32768 > $O/pmc_l2b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections, re
for d in ("$O/pmc_l2a", "$O/pmc_l2b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cosine_dense" in r["Kernel_Name"]:
                m = re.search(r"(k_cosine_dense_i8[_a-z0-9]*<[^>]*>)", r["Kernel_Name"])
                agg[m.group(1) if m else "other"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        print(k)
        for c, v in sorted(dd.items()):
            print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
