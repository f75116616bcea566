#!/bin/bash
# rocprofv3 kernel stats of the train_s2 bench (configs[2]) -> gpurun_out/r3_train/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o r1 -- python $GRAFT_REPO_ROOT/bench.py --workload train_s2 --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r3_train_bench.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r3_train
cp /tmp/prof_tr/*kernel_stats*.csv $GRAFT_REPO_ROOT/gpurun_out/r3_train/kernel_stats.csv 2>/dev/null || find /tmp/prof_tr -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r3_train/ \;
ls $GRAFT_REPO_ROOT/gpurun_out/r3_train
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r3_train_bench.log | cut -c1-300
