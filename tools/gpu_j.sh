#!/bin/bash
# round-2 profile pass 2: counters of the fused skinning+SDF kernel at B=512, train_s2 kernel stats, bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/p2; mkdir -p $O
bash tools/pmc2.sh r02 "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TC_STALL_sum" psi_skin_fwd_kernel python $GRAFT_REPO_ROOT/bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 > $O/skin_fwd_sdf_b512_counters.txt 2>&1
cat $O/skin_fwd_sdf_b512_counters.txt
cd /tmp; rm -rf /tmp/prof_s2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2 -o r -- python $GRAFT_REPO_ROOT/bench.py --workload train_s2 --steps 40 --warmup 3 > $O/train_s2_prof.log 2>&1
cp $(find /tmp/prof_s2 -name "*kernel_stats.csv" | head -1) $O/train_s2_kernel_stats.csv
cp $(find /tmp/prof_s2 -name "*kernel_trace.csv" | head -1) /tmp/s2_trace.csv
python - <<'PY'
import csv, os
# when do the multi-millisecond CK bwd_weight launches happen?  (timestamp of each relative to the first kernel of the run)
rows = list(csv.DictReader(open('/tmp/s2_trace.csv')))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
big = [(r['Kernel_Name'][:60], (int(r['Start_Timestamp']) - t0) / 1e9, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows
       if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 1e6]
out = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/p2/train_s2_long_kernels.txt')
with open(out, 'w') as f:
    f.write('kernels longer than 1 ms in a 43-step train_s2 run (run spans %.2f s of GPU timeline): name, start [s], duration [ms]\n' % ((t1 - t0) / 1e9))
    for b in big: f.write('%s  %.3f  %.3f\n' % b)
print(open(out).read()[:1500])
PY
cd $GRAFT_REPO_ROOT
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; grep real $O/bench_default.err
( time PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; tail -c 300 $O/bench_n2_gloo.json
timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 > $O/bench_habitat.json 2> $O/bench_habitat.err; tail -c 200 $O/bench_habitat.json
timeout 300 python tools/time_generation.py 2>&1 | tail -2
