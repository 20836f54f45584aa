#!/bin/bash
# tools only: kernel trace of the bench's timed region (no extras) -> per-queue busy times (tools/trace_streams.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tb
rocprofv3 --kernel-trace -d /tmp/tb -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc > $O/tb_bench.json 2> $O/tb.err || exit 1
db=$(ls -t /tmp/tb/*/*.db /tmp/tb/*.db 2>/dev/null | head -1)
python3 $R/tools/trace_streams.py $db 8 > $O/trace_streams.txt
python3 $R/tools/rocpd_stats.py $db $O/tb_stats.csv
cat $O/trace_streams.txt
