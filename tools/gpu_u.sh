#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final3; mkdir -p $O
bash tools/prof.sh r02i > $O/prof.log 2>&1; cp gpurun_out/prof_r02i/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary']['fitting_smplx_sparse_weights'].get('value'))"
