#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/f; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_entrypoints_gpu.py tests/test_dist_gpu.py -m gpu -q -x --timeout 900 ) > $O/new_tests.log 2>&1; echo "new rc=$?" >> $O/new_tests.log
tail -30 $O/new_tests.log
