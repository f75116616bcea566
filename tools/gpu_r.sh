#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lbs_gpu.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_abn.sh 2 head.so default
