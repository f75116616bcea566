#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/i; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_linear_gpu.py tests/test_models_cpu.py tests/test_training_gpu.py tests/test_parity_gaps_gpu.py -m gpu -q -x --timeout 900 ) > $O/new_tests.log 2>&1; echo "new rc=$?" >> $O/new_tests.log
tail -30 $O/new_tests.log
