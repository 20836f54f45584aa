#!/bin/bash
# tools only: PMC of the small int8 GEMM cosine (tools/bench_dense_fixed.py shapes): bytes from beyond L2, L2 hits, MFMA busy.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_ds1 $O/pmc_ds2 $O/pmc_ds3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_ds1 -o a -- python3 $R/tools/bench_dense_fixed.py > $O/pmc_ds1.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_ds2 -o b -- python3 $R/tools/bench_dense_fixed.py > $O/pmc_ds2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_ds3 -o c -- python3 $R/tools/bench_dense_fixed.py > $O/pmc_ds3.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ("$O/pmc_ds1", "$O/pmc_ds2", "$O/pmc_ds3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cosine_dense_i8_v5" in r["Kernel_Name"]:
                agg[(r["Grid_Size"] if "Grid_Size" in r else "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in sorted(agg.items()):
        print("grid", k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in sorted(dd.items())})
PY
