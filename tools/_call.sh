cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "knn or render_image or mesh or checkpoint or evaluate" 2>&1 | tail -3
python tools/eval_bench.py > gpurun_out/r04_eval_bench.json 2> gpurun_out/eval.err; cat gpurun_out/r04_eval_bench.json | head -c 1500
