cd $GRAFT_REPO_ROOT
python tools/_dbg.py 2>&1 | grep -v "amdgpu.ids\|Warning\|Consider\|print(" | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "render_ijs_autograd or neus_staged or end_to_end" 2>&1 | tail -5
