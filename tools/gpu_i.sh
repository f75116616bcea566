#!/bin/bash
# round-2 profile pass 1: counter list, rocprof stats of the fitting bench, FETCH/WRITE per kernel, MFMA counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/p1; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC|TA|TD|SQ|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/counters.txt; wc -l $O/counters.txt
bash $GRAFT_REPO_ROOT/tools/prof.sh r02 > $O/prof.log 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/prof_r02/*kernel_stats*.csv $O/ 2>/dev/null; tail -2 $O/prof.log
bash $GRAFT_REPO_ROOT/tools/pmc.sh r02 > $O/pmc.log 2>&1; tail -25 $O/pmc.log | cut -c1-200
for k in blend_fwd_kernel bwd_joint_kernel; do
  bash $GRAFT_REPO_ROOT/tools/pmc2.sh r02 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $k python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 > $O/mfma_$k.txt 2>&1
  echo "== $k"; cat $O/mfma_$k.txt
done
