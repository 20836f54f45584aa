#!/bin/bash
# One GPU-box run that produces everything profiles/ holds for a round:
#   the bench line, rocprofv3 --kernel-trace --stats of the same workload (once without the after-region
#   extras so that per-kernel averages are those of the timed steps, once with them for the config-5 scatter and
#   the MFMA kernels), and the two PMC passes (WRITE_SIZE, FETCH_SIZE; separate passes as
#   MI355X_MICROARCH.md prescribes).
# rocprofv3 must launch python3 directly (no env/bash wrapper between it and the program).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O && rm -rf $O/prof $O/prof_x $O/pmc_w $O/pmc_f
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
# (the trace databases are reduced to the --stats table on the box and deleted: with the small-batch extras the second
# one holds tens of thousands of dispatches, and gpurun only brings back 64 MiB)
stats() { python3 $R/tools/rocpd_stats.py "$(ls -t $1/*/*.db $1/*.db 2>/dev/null | head -1)" $2 && rm -rf $1; }
rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-live-pmc > $O/prof_bench.json 2> $O/prof.err || exit 1
stats $O/prof $O/kernel_stats.csv || exit 1
rocprofv3 --kernel-trace --stats -d $O/prof_x -o x -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc > $O/prof_x_bench.json 2> $O/prof_x.err || exit 1
stats $O/prof_x $O/kernel_stats_with_extras.csv || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o write_size -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc > /dev/null 2> $O/pmc_w.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o fetch_size -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-live-pmc > /dev/null 2> $O/pmc_f.err || exit 1
cut -c1-300 $O/bench.json
