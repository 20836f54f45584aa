#!/bin/bash
# PMC passes for the heavy-row panel pipeline on the skewed workload (k_panel_* and k_cosine_heavy with the panels on):
# one counter group per pass, rocprofv3 launches python3 directly (the variable is exported here, not through `env`).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export SKM_HEAVY_PANEL=1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_panel_$i -o p -- python3 $R/tools/bench_skewed.py 100000 2 > /dev/null 2> $R/gpurun_out/pmc_panel_$i.err || echo "pass $i failed"
done
for k in k_cosine_heavy k_panel_key k_panel_dict k_panel_rows k_panel_gemm; do
  python3 $R/tools/pmc_summary.py $R/gpurun_out/ $k | grep -A8 "^$k" | head -12
done
