#!/bin/bash
# round-4 measurement pass behind profiles/r04_*: GPU suite, rocprofv3 kernel stats (fitting + steady-state train_s2), PMC traffic, counters of the
# skinning + SDF kernel at B = 512 (dense / compressed rows), sensitivity, bench lines (driver command, 1-rank RCCL loop from C, 2-rank gloo, habitat)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final4; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/prof.sh r04f > $O/prof.log 2>&1; cp gpurun_out/prof_r04f/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log | cut -c1-200
bash tools/pmc.sh r04f > $O/pmc.log 2>&1; cp gpurun_out/pmc_r04f_*.txt $O/
[ -n "$SKIP_PMC512" ] || bash tools/gpu_call.sh final4 pmc512:dense pmc512:sparse > $O/pmc512.log 2>&1
# the bench lines below cite profiles/r04_* (rocprofv3 averages, PMC traffic): refresh the box's copy from THIS pass first
cp $O/kernel_stats.csv profiles/r04_kernel_stats.csv; python tools/mk_pmc_json.py r04f r04 > /dev/null; cp profiles/r04_pmc_traffic.json profiles/r04_pmc_fetch_size.csv profiles/r04_pmc_write_size.csv $O/
bash tools/gpu_call.sh final4 timeline timeline:10 timeline2 > $O/timelines.log 2>&1
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
( PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err
( PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err
timeout 900 python tools/sensitivity.py > $O/sens.log 2>&1; cp gpurun_out/sensitivity.json $O/
[ -n "$SKIP_TRAIN" ] || bash tools/gpu_call.sh final4 proftrain > $O/proftrain_summary.log 2>&1
python - <<'PY'
import json
for f in ('bench_default','bench_dp1_nccl','bench_n2_gloo','bench_habitat'):
    try:
        d=json.loads([l for l in open('gpurun_out/final4/%s.json'%f) if l.startswith('{')][-1])
        sec = d.get('secondary') or {}
        print(f, d['value'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), d['config'].get('rccl_ranks_seen'), d['config'].get('dp_launch_mode'),
              {k: (v.get('frac'), v.get('ms_per_step')) for k, v in sec.items()})
    except Exception as e: print(f, 'ERR', e)
PY
