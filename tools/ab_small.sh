#!/bin/bash
# tools only: tools/small_batch_gap.py under several builds of the library on one box.  usage: tools/ab_small.sh name1 name2 ... (default = the product)
R=${GRAFT_REPO_ROOT:-/root/repo}
for name in default "$@"; do
  lib=""; [ "$name" != default ] && lib=$R/snekmer_amd/libsnekmer_hip_$name.so
  SNEKMER_HIP_LIB=$lib python3 $R/tools/small_batch_gap.py 1000 3383 10000 2>/dev/null | python3 -c "
import json,sys
for r in json.load(sys.stdin): print('$name', r['n'], round(r['wall_ms_per_step'],4), {k:v for k,v in r['stages'].items() if 'sort' in k})
"
done
