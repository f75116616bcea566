#!/bin/bash
# round-3 measurement pass: kernel stats, PMC traffic, skinning + SDF counters at B = 512, MFMA counters (blend GEMMs, conv kernels),
# sensitivity, bench lines (driver command, 1-rank RCCL loop from C, 2-rank gloo, habitat), train_s2 kernel stats, GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final3; mkdir -p $O
bash tools/prof.sh r03f > $O/prof.log 2>&1; cp gpurun_out/prof_r03f/*kernel_stats*.csv $O/kernel_stats.csv; tail -1 $O/prof.log
bash tools/pmc.sh r03f > $O/pmc.log 2>&1; cp gpurun_out/pmc_r03f_*.txt $O/
bash tools/pmc2.sh r03f "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TD_TC_STALL_sum" psi_skin_fwd_kernel python $GRAFT_REPO_ROOT/bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 > $O/skin_fwd_sdf_b512_counters.txt 2>&1
for k in conv3x3_kernel conv3x3_wrw_kernel; do
  bash tools/pmc2.sh r03f "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" $k python $GRAFT_REPO_ROOT/bench.py --workload train_s2 --steps 5 --warmup 2 > $O/mfma_$k.txt 2>&1
done
timeout 900 python tools/sensitivity.py > $O/sens.log 2>&1; cp gpurun_out/sensitivity.json $O/
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
( PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err
( PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err
bash tools/r3_prof_train.sh > $O/prof_train.log 2>&1; cp gpurun_out/r3_train/kernel_stats.csv $O/train_s2_kernel_stats.csv
timeout 300 python tools/time_linear_bwd.py > $O/linear_bwd.log 2>&1; cp gpurun_out/linear_bwd_times.json $O/
timeout 300 python tools/time_conv.py > $O/conv.log 2>&1; cp gpurun_out/conv_times.json $O/
python - <<'PY'
import json
for f in ('bench_default','bench_dp1_nccl','bench_n2_gloo','bench_habitat'):
    try:
        d=json.loads([l for l in open('gpurun_out/final3/%s.json'%f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['n_gpus'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), (d.get('secondary') or {}).get('train_s2',{}).get('ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
