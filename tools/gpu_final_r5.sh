#!/bin/bash
# round-5 measurement pass behind profiles/r05_*: GPU suite, rocprofv3 kernel stats (fitting; steady-state train_s2 in both precisions), PMC traffic,
# bench lines (driver command, 1-rank RCCL loop from C, 2-rank gloo, habitat).  SKIP_TESTS=1 / SKIP_TRAIN=1 shorten it.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final5; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { ( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; cp gpurun_out/arbiter/*.json $O/ 2>/dev/null; }
bash tools/prof.sh r05f > $O/prof.log 2>&1; cp gpurun_out/prof_r05f/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log | cut -c1-200
bash tools/pmc.sh r05f > $O/pmc.log 2>&1; cp gpurun_out/pmc_r05f_*.txt $O/
# the bench lines below cite profiles/r05_* (rocprofv3 averages, PMC traffic): refresh the box's copy from THIS pass first
cp $O/kernel_stats.csv profiles/r05_kernel_stats.csv; python tools/mk_pmc_json.py r05f r05 > /dev/null; cp profiles/r05_pmc_traffic.json profiles/r05_pmc_fetch_size.csv profiles/r05_pmc_write_size.csv $O/
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
( PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err
( PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err
if [ -z "$SKIP_TRAIN" ]; then
  bash tools/gpu_call.sh final5 "proftrain:0:--bf16 1" > $O/proftrain_bf16_summary.log 2>&1; cp $O/train_s2_kernel_stats.csv $O/train_s2_bf16_kernel_stats_unfiltered.csv; cp $O/proftrain.log $O/proftrain_bf16.log
  bash tools/gpu_call.sh final5 "proftrain:0:--bf16 0" > $O/proftrain_fp32_summary.log 2>&1; cp $O/train_s2_kernel_stats.csv $O/train_s2_fp32_kernel_stats_unfiltered.csv; cp $O/proftrain.log $O/proftrain_fp32.log
fi
python - <<'PY'
import json
for f in ('bench_default','bench_dp1_nccl','bench_n2_gloo','bench_habitat'):
    try:
        d=json.loads([l for l in open('gpurun_out/final5/%s.json'%f) if l.startswith('{')][-1])
        sec = d.get('secondary') or {}
        print(f, d['value'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), d['config'].get('rccl_ranks_seen'), d['config'].get('dp_launch_mode'),
              {k: (v.get('frac'), v.get('ms_per_step')) for k, v in sec.items()})
    except Exception as e: print(f, 'ERR', e)
PY
