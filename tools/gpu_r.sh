#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops_gpu.py tests/test_fitting_gpu.py tests/test_parity_gaps_gpu.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_abn.sh 2 head.so default
