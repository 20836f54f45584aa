# tools only: the sort of a mid-size batch (5 000 - 20 000 sequences = 1.4 - 5.8 M pairs): rocPRIM against the own one-sweep sort
for srt in rocprim onesweep; do
  echo "== SKM_SORT=$srt"
  SKM_SORT=$srt timeout -k 10 150 python tools/small_batch_gap.py 3383 5000 7000 8500 10000 12000 14000 20000 | python -c "
import json,sys
for r in json.loads(sys.stdin.readlines()[-1]): print(r['n'], round(r['wall_ms_per_step'],4), {k:v for k,v in r['stages'].items() if 'sort' in k})
"; done
