#!/bin/bash
# PMC passes for k_gram_sparse (one counter group per pass; rocprofv3 must launch python3 directly).
# usage: tools/pmc_gram.sh [tag]   (SKM_GRAM_ABLATE from the environment selects a diagnostic build)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-full}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_VMEM_RD SQ_LDS_ATOMIC_RETURN" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  [ -n "$PMC_PASSES" ] && [ $i -gt $PMC_PASSES ] && break
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_gram_${TAG}_$i -o g -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc > /dev/null 2> $R/gpurun_out/pmc_gram_${TAG}_$i.err || echo "pass $i failed"
done
echo "== $TAG"
python3 $R/tools/pmc_summary.py $R/gpurun_out/ k_gram_sparse | awk '/^k_gram_sparse$/{f=1;next} /^k_/{f=0} f'
