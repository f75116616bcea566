#!/bin/bash
# round-2 final pass: full GPU suite, kernel stats, PMC traffic + MFMA counters, skinning+SDF counters at B=512, sensitivity, bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
bash tools/prof.sh r02f > $O/prof.log 2>&1; cp gpurun_out/prof_r02f/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log
bash tools/pmc.sh r02f > $O/pmc.log 2>&1
for k in blend_fwd_kernel bwd_joint_kernel; do
  bash tools/pmc2.sh r02f "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" $k python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary 0 > $O/mfma_$k.txt 2>&1
done
bash tools/pmc2.sh r02f "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum" psi_skin_fwd_kernel python $GRAFT_REPO_ROOT/bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 > $O/skin_fwd_sdf_b512_counters.txt 2>&1
timeout 900 python tools/sensitivity.py > $O/sens.log 2>&1; cp gpurun_out/sensitivity.json $O/
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
( time PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
( PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err
timeout 600 python tools/time_files.py > $O/time_files.log 2>&1; cp gpurun_out/files_per_s.json $O/
python - <<'PY'
import json
for f in ('bench_default','bench_n2_gloo','bench_dp1_nccl','bench_habitat'):
    try:
        d=json.loads([l for l in open('gpurun_out/final/%s.json'%f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
