#!/bin/bash
# round 6: the same counters as tools/gpu_pmc_conv.sh for the stride-1 3x3 kernels of the fp32 mode that split their operands once per
# workgroup (conv.hip: conv3x3s_kernel forward / input gradient, conv3x3_wrw3_kernel weight gradient), layer-1 and layer-2 shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv; mkdir -p $O
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
for ch in 64 128; do
  H=32; [ $ch = 128 ] && H=16
  echo "## conv3x3s_kernel, $ch -> $ch channels, ${H}x${H} maps, batch 128: forward launches (conv_one.py fwd)"
  bash tools/pmc2.sh c3s "$C" conv3x3s_kernel python $GRAFT_REPO_ROOT/tools/conv_one.py $ch $ch 3 1 1 $H 3 fwd
  echo "## conv3x3_wrw3_kernel, same shape (conv_one.py bwd)"
  bash tools/pmc2.sh c3w "$C" conv3x3_wrw3_kernel python $GRAFT_REPO_ROOT/tools/conv_one.py $ch $ch 3 1 1 $H 3 bwd
done > $O/pmc_conv3x3_split.txt 2>&1
cat $O/pmc_conv3x3_split.txt
