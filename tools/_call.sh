cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_like.json 2> gpurun_out/driver_like.err ) 2>&1 | tail -3
python -c "
import json; d=json.loads(open('gpurun_out/driver_like.json').read().strip().split('\n')[-1])
print({k:(v if not isinstance(v,(dict,list)) else '...') for k,v in d.items()})
print(d['ms_per_step_windows']); print(d['roofline']['frac'], d['roofline']['traffic']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:80])
print(d['aux_default']['roofline_fwd']['l2_hit_rate'], d['aux_default']['roofline_fwd']['traffic'])"
