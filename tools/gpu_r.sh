#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final3; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
bash tools/gpu_abn.sh 1 head.so default
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
