#!/bin/bash
# One GPU call = a list of measurement steps (boxes differ by 2-3 %, so A/B comparisons live inside one call).
# usage: tools/gpu_call.sh <tag> <step>...     steps: tests | bench | ab:<lib,lib,..>[:bench args] | b512[:lib] | pmc512:<dense|sparse> | prof
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
line() {   # bench JSON line -> short summary
  python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('$1', d['ms_per_step'], ' '.join('%s=%.1f/%.3f'%(k.replace('_kernel',''),v.get('us'),v.get('frac_hbm',0) or 0) for k,v in kb.items()))"
}
# a variant is <lib>[@VAR=value[@VAR=value]]: library build (default = the in-tree one) + environment switches for this run only
uselib() {
  local spec=$1 lib=${1%%@*}
  for v in $PSI_CALL_VARS; do unset $v; done; PSI_CALL_VARS=""
  if [ "$lib" = default ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$lib.so; fi
  if [ "$spec" != "$lib" ]; then IFS=@ read -ra kv <<< "${spec#*@}"; for e in "${kv[@]}"; do export "$e"; PSI_CALL_VARS="$PSI_CALL_VARS ${e%%=*}"; done; fi
}
for step in "$@"; do
  IFS=: read -r what a1 a2 <<< "$step"
  case $what in
    tests) ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log ;;
    bench) ( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | line default ;;
    ab) for r in 1 2; do for l in ${a1//,/ }; do uselib $l; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 $a2 2>>$O/ab.err | tail -1 | tee -a "$O/ab_$l.json" | line "$l [$a2]"; done; done; uselib default ;;
    pmc512) args="--batch 512"; [ "$a1" = sparse ] && args="--batch 512 --weight-nnz 4"
      bash tools/pmc2.sh $TAG "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" ${a2:-psi_skin_fwd_kernel} python $GRAFT_REPO_ROOT/bench.py $args --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 > $O/pmc_skin_fwd_sdf_b512_$a1.txt 2>&1; cat $O/pmc_skin_fwd_sdf_b512_$a1.txt ;;
    pmc) # pmc:<kernel substring>:<bench args> — counters of one kernel of the default bench
      bash tools/pmc2.sh $TAG "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES" ${a1:-fwd_scene_kernel} python $GRAFT_REPO_ROOT/bench.py $a2 --steps 10 --warmup 3 --no-cpu-baseline --secondary 0 > "$O/pmc_$a1.txt" 2>&1; cat "$O/pmc_$a1.txt" ;;
    stops) # stops:<list of PSI_SKIN_STOP values>: rocprofv3 average of fwd_scene / skin_bwd_v / bwd_joint with the -DPSI_HEAD_STOPS library (tools/_variants/stops.so)
      for k in ${a1//,/ }; do export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so PSI_SKIN_STOP=$k; rm -rf /tmp/pst; ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pst -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 $a2 > $O/stops_$k.log 2>&1 )
        python - "$(find /tmp/pst -name '*kernel_stats.csv' | head -1)" $k <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r['Calls']) > 1000]
print('stop', sys.argv[2], ' '.join('%s=%.2f' % (r['Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:24], float(r['AverageNs']) / 1e3) for r in sorted(rows, key=lambda r: r['Name'])))
PY
      done; unset PSI_HIP_LIB PSI_SKIN_STOP ;;
    timeline) PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so PSI_SKIN_STOP=${a1:-9} python tools/timeline.py > $O/timeline_${a1:-9}.txt 2>&1; tail -18 $O/timeline_${a1:-9}.txt | cut -c1-400 ;;
    timeline2) PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops2.so python tools/timeline2.py > $O/timeline2.txt 2>&1; tail -12 $O/timeline2.txt | cut -c1-300 ;;
    profab) for l in ${a1//,/ }; do uselib $l; rm -rf /tmp/pab; ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pab -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 $a2 > "$O/profab_$l.log" 2>&1 )
        f=$(find /tmp/pab -name "*kernel_stats.csv" | head -1); cp $f "$O/kernel_stats_$l.csv"
        python - "$f" "$l" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r['Calls']) > 1000]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print(sys.argv[2], 'sum %.1f us |' % sum(float(r['AverageNs']) for r in rows[:7]) ,' '.join('%s=%.2f' % (r['Name'].split('(')[0].split('<')[0][-22:], float(r['AverageNs']) / 1e3) for r in rows[:8]))
PY
      done; uselib default ;;
    proftrain) rm -rf /tmp/ptr; ( cd /tmp; PSI_MIOPEN_FIND=${a1:-0} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr -o p -- python $GRAFT_REPO_ROOT/bench.py --workload train_s2 --steps 40 --warmup 5 $a2 > $O/proftrain.log 2>&1 )
      f=$(find /tmp/ptr -name "*kernel_stats.csv" | head -1); cp $f $O/train_s2_kernel_stats.csv; tail -1 $O/proftrain.log | cut -c1-400
      python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print('total kernel time %.1f ms, %d kernels' % (tot / 1e6, len(rows)))
for r in rows[:40]:
    print('%5.1f%% calls %6s avg %8.1f us  %s' % (100 * float(r['TotalDurationNs']) / tot, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
      ;;
    prof) rm -rf /tmp/prof_$TAG; ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 > $O/prof.log 2>&1 ); cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-150 ;;
    *) echo "unknown step $step" ;;
  esac
done
