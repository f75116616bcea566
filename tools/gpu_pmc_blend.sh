#!/bin/bash
# round 6: matrix-pipe counters of the two split-product blend kernels in the default bench (one rocprofv3 --pmc pass per counter)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_blend; mkdir -p $O
C="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for k in blend_fwd_h_kernel fit_bwd_joint_kernel; do
  echo "## $k (bench.py --steps 20 --warmup 3, B = 32)"
  bash tools/pmc2.sh pb "$C" $k python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0
done > $O/pmc_blend.txt 2>&1
cat $O/pmc_blend.txt
