#!/bin/bash
# GPU box: kernel traces of small steps (tools/trace_small.py) reduced to a timeline + stats under gpurun_out/trace_<what>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  rm -rf /tmp/tr_$what
  rocprofv3 --kernel-trace -d /tmp/tr_$what -o t -- python3 $R/tools/trace_small.py $what 20 > /dev/null 2> $O/trace_$what.err || exit 1
  db=$(ls -t /tmp/tr_$what/*/*.db /tmp/tr_$what/*.db 2>/dev/null | head -1)
  python3 $R/tools/trace_timeline.py $db > $O/trace_$what.txt || exit 1
  python3 $R/tools/rocpd_stats.py $db $O/trace_${what}_stats.csv
done
