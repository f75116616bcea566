#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/r1_kernel_stats.csv + bench json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cp /tmp/prof_$tag/*stats*.csv $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/
grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.log
