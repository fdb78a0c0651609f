#!/bin/bash
# two ranks sharing the one GPU of the test box: the loss exchange as torch.distributed (gloo here) vs the peer kernel inside one graph
O=gpurun_out/r3; mkdir -p $O
export NGM_BENCH_SHARE_GPU=1 NGM_DIST_BACKEND=gloo
for ex in rccl peer; do
  timeout 300 python bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline --no-aux-hash --exchange $ex > $O/bench_2ranks_$ex.json 2> $O/bench_2ranks_$ex.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_2ranks_$ex.json").read().strip().splitlines()[-1]); print("$ex", d["ms_per_step"], d["value"], d["config"]["launch"])
except Exception as e:
    print("$ex failed", e); print(open("$O/bench_2ranks_$ex.err").read()[-1500:])
PY
done
timeout 300 python bench.py --gpus 2 --scene-sim --steps 50 --warmup 10 --exchange peer > $O/scene_sim_2ranks_peer.json 2> $O/scene_sim_2ranks_peer.err
tail -c 1500 $O/scene_sim_2ranks_peer.json; tail -3 $O/scene_sim_2ranks_peer.err
