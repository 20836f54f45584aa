#!/bin/bash
# A/B of two builds of the library on ONE GPU box (box-to-box variation is 3 %): the bench's timed region, three
# alternating runs each.  usage: tools/ab_lib.sh [other_lib.so]   (default snekmer_amd/libsnekmer_hip_ab.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-$R/snekmer_amd/libsnekmer_hip_ab.so}
for i in 1 2 3; do
  for lib in "" "$B"; do
    SNEKMER_HIP_LIB=$lib python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-live-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('${lib:-default}'.split('/')[-1], round(d['ms_per_step'],3), {k:round(v,3) for k,v in s.items() if k in ('k_gram_sparse','k_cosine_heavy','k_cosine_write')})"
  done
done
