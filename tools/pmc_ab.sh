#!/bin/bash
# PMC comparison of the pair-loop generations (tools/ab_pipeline.py runs both in one process).
# usage: tools/pmc_ab.sh   -> gpurun_out/pmc_ab_*.txt   (rocprofv3 must launch python3 directly)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_ab1 $O/pmc_ab2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $O/pmc_ab1 -o a -- python3 $R/tools/ab_pipeline.py 100000 1 > $O/pmc_ab1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU \
  --output-format csv -d $O/pmc_ab2 -o b -- python3 $R/tools/ab_pipeline.py 100000 1 > $O/pmc_ab2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $O/pmc_ab1 gram > $O/pmc_ab_1.txt
python3 $R/tools/pmc_summary.py $O/pmc_ab2 gram > $O/pmc_ab_2.txt
cat $O/pmc_ab_1.txt $O/pmc_ab_2.txt
