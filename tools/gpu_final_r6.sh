#!/bin/bash
# round-6 measurement pass behind profiles/r06_*: GPU suite, rocprofv3 kernel stats (fitting dense + sparse rows; steady-state train_s2 in both
# precisions), PMC traffic, per-launch distribution and workgroup timeline of fwd_scene, Infinity-Cache probe of the blend stream, sensitivity
# sweep, bench lines (driver command, 1-rank RCCL loop from C, 2-rank gloo, habitat).  SKIP_TESTS=1 / SKIP_TRAIN=1 / SKIP_SENS=1 shorten it.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final6; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { rm -rf gpurun_out/arbiter; ( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log; mkdir -p $O/arbiter; cp gpurun_out/arbiter/*.json $O/arbiter/ 2>/dev/null; }
bash tools/prof.sh r06f > $O/prof.log 2>&1; cp gpurun_out/prof_r06f/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log | cut -c1-200
bash tools/prof.sh r06s --weight-nnz 4 > $O/prof_sparse.log 2>&1; cp gpurun_out/prof_r06s/*kernel_stats*.csv $O/kernel_stats_sparse_rows.csv
bash tools/pmc.sh r06f > $O/pmc.log 2>&1; cp gpurun_out/pmc_r06f_*.txt $O/
cp $O/kernel_stats.csv profiles/r06_kernel_stats.csv; python tools/mk_pmc_json.py r06f r06 > /dev/null; cp profiles/r06_pmc_traffic.json profiles/r06_pmc_fetch_size.csv profiles/r06_pmc_write_size.csv $O/
# per-launch distribution of the dominant kernel + workgroup timeline (the -DPSI_HEAD_STOPS variant, when it travelled with the snapshot)
( cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/launch_hist.py /tmp/kt fwd_scene > $O/launch_hist_fwd_scene.txt 2>&1 )
[ -f tools/_variants/stops.so ] && { PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so PSI_SKIN_STOP=9 python tools/timeline.py > $O/timeline_fwd_scene.txt 2>&1; }
[ -f tools/_variants/stops.so ] && { PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so PSI_SKIN_STOP=11 python tools/timeline_joint.py 2>&1 | grep -v '^{' > $O/timeline_bwd_joint.txt; }
# the blend stream alone in a loop (its 64.5 MB matrix can stay in the 256 MB Infinity Cache) against the same kernel inside the iteration, + L2 counters
( cd /tmp; rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $GRAFT_REPO_ROOT/tools/time_blend_only.py 32 > /dev/null 2>&1; grep blend_fwd $(find /tmp/pb -name '*kernel_stats.csv' | head -1) > $O/blend_alone_kernel_stats.txt )
bash tools/pmc2.sh r06l3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum FETCH_SIZE" blend_fwd_h_kernel python $GRAFT_REPO_ROOT/tools/time_blend_only.py 32 > $O/pmc_blend_alone.txt 2>&1
bash tools/pmc2.sh r06l3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum FETCH_SIZE" blend_fwd_h_kernel python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 > $O/pmc_blend_in_iteration.txt 2>&1
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
( PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err
( PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err
[ -n "$SKIP_SENS" ] || { timeout 900 python tools/sensitivity.py > $O/sens.log 2>&1; cp gpurun_out/sensitivity.json $O/; }
if [ -z "$SKIP_TRAIN" ]; then
  bash tools/gpu_call.sh final6 "proftrain:0:--bf16 1" > $O/proftrain_bf16_summary.log 2>&1; cp $O/train_s2_kernel_stats.csv $O/train_s2_bf16_kernel_stats_unfiltered.csv; cp $O/proftrain.log $O/proftrain_bf16.log
  bash tools/gpu_call.sh final6 "proftrain:0:--bf16 0" > $O/proftrain_fp32_summary.log 2>&1; cp $O/train_s2_kernel_stats.csv $O/train_s2_fp32_kernel_stats_unfiltered.csv; cp $O/proftrain.log $O/proftrain_fp32.log
fi
python - <<'PY'
import json
for f in ('bench_default','bench_dp1_nccl','bench_n2_gloo','bench_habitat'):
    try:
        d=json.loads([l for l in open('gpurun_out/final6/%s.json'%f) if l.startswith('{')][-1])
        sec = d.get('secondary') or {}
        print(f, d['value'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), d['config'].get('rccl_ranks_seen'), d['config'].get('dp_launch_mode'),
              {k: (v.get('frac'), v.get('ms_per_step')) for k, v in sec.items()})
    except Exception as e: print(f, 'ERR', e)
PY
