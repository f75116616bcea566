#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
( time timeout 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n1_k100.json 2> $O/bench_n1_k100.err; echo "n1k100 rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c/bench_n1_k100.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernels_us'])
PY
