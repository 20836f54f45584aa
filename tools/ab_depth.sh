for cfg in "1 0.6" "2 0.6" "2 0.8" "2 1.0" "3 1.0" "2 0.9"; do set -- $cfg; SKM_AB_DEPTH=$1 SKM_AB_FRACTION=$2 timeout -k 10 150 python tools/ab_overlapped.py 100000 20 2>&1 | tail -1; done
