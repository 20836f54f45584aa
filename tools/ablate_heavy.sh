# tools only: timing-only ablations of k_cosine_heavy on synth_skewed (libsnekmer_hip_diag.so, `make -C snekmer_amd/csrc diag`;
# results NOT valid).  SKM_HEAVY_ABLATE bits: 1 no global stores, 2 no walk of long lists, 4 no panel row, 8 no short lists /
# cache, 16 nothing after a row's set-up.  usage: tools/ablate_heavy.sh [library] [pack values] [ablations]
export SNEKMER_HIP_LIB=${1:-$PWD/snekmer_amd/libsnekmer_hip_diag.so}
for pk in ${2:-1 0}; do for abl in ${3:-0 1 2 4 8 14 15 16}; do
  echo -n "pack=$pk ablate=$abl: "
  SKM_HEAVY_PACK=$pk SKM_HEAVY_ABLATE=$abl timeout -k 10 200 python tools/bench_skewed.py 100000 3 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],2), 'heavy', r['stages'].get('k_cosine_heavy'))"
done; done
