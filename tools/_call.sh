cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -k "timeout or peer" 2>&1 | tail -3
