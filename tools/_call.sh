cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py -x -q -k "encode" 2>&1 | tail -3
