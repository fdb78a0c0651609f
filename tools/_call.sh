cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "quadrature or render_image or neus or composite" 2>&1 | tail -3
python tools/stage_bench.py --out gpurun_out/r04_stage_bench.json 2>gpurun_out/stage.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1])
for k,v in d['stages'].items(): print(k, {a:b for a,b in v.items() if a!='note'})"
tail -3 gpurun_out/stage.err
