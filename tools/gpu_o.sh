#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fitting_gpu.py -m gpu -q -x -k "many or golden or packed" 2>&1 | tail -3
timeout 600 python tools/time_files.py 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
