#!/bin/bash
# A/B of two builds of the library on ONE GPU box (box-to-box variation is 3 %): the bench's timed region, three
# alternating runs each.  usage: tools/ab_lib.sh [other_lib.so]   (default snekmer_amd/libsnekmer_hip_ab.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-$R/snekmer_amd/libsnekmer_hip_ab.so}
for i in 1 2 3; do
  for lib in "" "$B"; do
    SNEKMER_HIP_LIB=$lib python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc 2>/dev/null | python3 $R/tools/ab_print.py "${lib:-default}"
  done
done
