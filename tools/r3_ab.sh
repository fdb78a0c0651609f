#!/bin/bash
# the default bench line + the same step behind a separate k_stash_bwd launch, same box
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench_line_ab.json 2>/dev/null
NGM_NO_FUSED_COMP=1 python bench.py --no-cpu-baseline --no-aux-hash > $O/bench_line_ab_unfused.json 2>/dev/null
NGM_NO_FUSED_COMP=1 python bench.py --variant hash --no-cpu-baseline > $O/bench_line_ab_hash_unfused.json 2>/dev/null
python - <<PY
import json
for f in ("bench_line_ab","bench_line_ab_unfused","bench_line_ab_hash_unfused"):
    d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]/1e9,3), d["kernels_us"], (d.get("aux_hash") or {}).get("ms_per_step"))
PY
