#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fitting_gpu.py tests/test_entrypoints_gpu.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python tools/time_files.py 2>&1 | tail -10
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('habitat', d['value'], d['ms_per_step'])"
