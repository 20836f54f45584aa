#!/bin/bash
# PMC passes for k_cosine_heavy on the skewed workload (one counter group per pass; rocprofv3 launches python3 directly).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_heavy_$i -o h -- python3 $R/tools/bench_skewed.py 100000 2 > /dev/null 2> $R/gpurun_out/pmc_heavy_$i.err || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/ k_cosine_heavy
