#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>  -- one rocprofv3 PMC pass over the torch-free timing harness
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/gpu_check.py ${NGM_CHECK:-time} --out=/tmp/t.json > /tmp/pmc_$tag.log 2>&1
tail -2 /tmp/pmc_$tag.log | cut -c1-300
