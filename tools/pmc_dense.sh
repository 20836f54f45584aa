#!/bin/bash
# PMC of the dense i8 MFMA cosine variants (tools/ab_dense.py runs variants 3, 5, 4 in one process).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_d1 $O/pmc_d2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_d1 -o a -- python3 $R/tools/ab_dense.py hydro 14 32768 > $O/pmc_d1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU \
  --output-format csv -d $O/pmc_d2 -o b -- python3 $R/tools/ab_dense.py hydro 14 32768 > $O/pmc_d2.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ("$O/pmc_d1", "$O/pmc_d2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    order = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cosine_dense" in r["Kernel_Name"]:
                import re as _re; _m = _re.search(r"(k_cosine_dense_i8[_a-z0-9]*<[^>]*>)", r["Kernel_Name"]); key = _m.group(1) if _m else "other"
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        print(k)
        for c, v in sorted(dd.items()):
            print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
