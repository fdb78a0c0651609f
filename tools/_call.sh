cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['ms_per_step_windows']['min'], d['kernels_us'], d['aux_hash']['ms_per_step'], d['aux_default']['ms_per_step'])"
