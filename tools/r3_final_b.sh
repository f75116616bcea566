#!/bin/bash
# round-3 closing pass after the train_s2 glue work: the driver's bench command, train_s2 kernel stats, smoke
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final3b; mkdir -p $O
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
bash tools/r3_prof_train.sh > $O/prof_train.log 2>&1; cp gpurun_out/r3_train/kernel_stats.csv $O/train_s2_kernel_stats.csv
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/final3b/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['secondary']['train_s2']['ms_per_step'], d['secondary']['train_s2'].get('conv_kernel_roofline'), d['cpu_baseline']['value'])
PY
timeout 300 python tools/time_generation.py > $O/gen.log 2>&1; cp gpurun_out/generation_times.json $O/ 2>/dev/null; tail -2 $O/gen.log
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E " passed| failed" $O/pytest_gpu.log
