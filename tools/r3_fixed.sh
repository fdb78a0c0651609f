#!/bin/bash
# fixed-cost experiments: forward waves per workgroup on small grids, k_stash_bwd prefetch
O=gpurun_out/r3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or neus or density or golden or seeded" > $O/pytest_fixed.txt 2>&1; tail -3 $O/pytest_fixed.txt
for w in none 4 4b; do
  if [ $w = none ]; then unset NGM_FWD_WAVES; else export NGM_FWD_WAVES=$w; fi
  timeout 200 python bench.py --scene-sim --steps 50 > $O/scene_w$w.json 2> $O/scene_w$w.err
  python - <<PY
import json
d=json.load(open("$O/scene_w$w.json"))
for k,v in d["configs"].items():
    sw=v["ms_per_step_vs_active_fields"]
    print("$w", k, " ".join(f"{a}={b}" for a,b in sw.items() if "graph" in a), "| F=1", sw["F=1,kernels_us"], "| F=4", sw["F=4,kernels_us"])
PY
done
unset NGM_FWD_WAVES
timeout 300 python bench.py --no-cpu-baseline --no-aux-hash | tail -c 400
