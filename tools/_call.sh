cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "permuto or hash or cfg2 or cfg3" 2>&1 | tail -3
for v in 0 1; do if [ $v = 1 ]; then export NGM_NO_SIDE_STREAM=1; fi; python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('noside=$v', d['ms_per_step'], 'aux_hash', d['aux_hash']['ms_per_step'], 'aux_default', d['aux_default']['ms_per_step'])"; done
