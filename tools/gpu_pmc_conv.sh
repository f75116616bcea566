#!/bin/bash
# round 6: counters of the general convolution kernels of the fp32 (three-term) mode at the layer-1 shape of the train_s2 step (64 -> 64
# channels, 3x3, stride 1, 32 x 32 maps, batch 128): vector / matrix instruction counts and matrix-pipe busy cycles per launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv; mkdir -p $O
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
bash tools/pmc2.sh convf "$C" conv_gemm_kernel python $GRAFT_REPO_ROOT/tools/conv_one.py 64 64 3 1 1 32 3 fwd > $O/pmc_conv_gemm.txt 2>&1
bash tools/pmc2.sh convb "$C" conv_wgrad_kernel python $GRAFT_REPO_ROOT/tools/conv_one.py 64 64 3 1 1 32 3 bwd > $O/pmc_conv_wgrad.txt 2>&1
cat $O/pmc_conv_gemm.txt; echo; cat $O/pmc_conv_wgrad.txt
