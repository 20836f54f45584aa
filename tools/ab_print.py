#!/usr/bin/env python3
"""One line per bench run for tools/ab_lib.sh: label, ms per step, per-stage ms."""
import json
import sys

d = json.loads(sys.stdin.read())
s = d["stage_ms_per_step"]
keep = ("k_count_short", "k_compact_rows", "rocprim_radix_sort_codes", "k_basis_scatter", "k_gram_sparse", "k_cosine_write")
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in s.items() if k in keep})
