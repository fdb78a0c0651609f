for v in "" "-DNGM_ABLH_NOSIMPLEX" "-DNGM_ABLH_NOSCATTER" "-DNGM_ABLH_NOSIMPLEX -DNGM_ABLH_NOSCATTER"; do
  NGM_HIPCC_EXTRA="$v" python -m neural_graph_mapping_amd.build > /tmp/build.log 2>&1 || { echo "build failed: $v"; tail -3 /tmp/build.log; continue; }
  echo "== [$v] $(python bench.py --variant hash --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_us'])")"
done
