cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "permuto or hash or cfg2 or cfg3 or reduced_precision" 2>&1 | tail -4
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['aux_hash']['ms_per_step'], d['aux_hash']['kernels_us']); print('aux_default', d['aux_default']['ms_per_step'], d['aux_default']['kernels_us'])"
