#!/bin/bash
cd $GRAFT_REPO_ROOT && python -m pytest tests/test_conv_gpu.py tests/test_bnorm_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cv -o r1 -- python $GRAFT_REPO_ROOT/tools/time_conv.py > $GRAFT_REPO_ROOT/gpurun_out/conv_prof.log 2>&1
grep -v amdgpu.ids $GRAFT_REPO_ROOT/gpurun_out/conv_prof.log | grep "^{" 
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_cv/**/*kernel_stats.csv',recursive=True)[0]
for r in sorted(csv.DictReader(open(f)),key=lambda r:-float(r['TotalDurationNs'])):
    if float(r['AverageNs'])<1e6 and ('conv' in r['Name'] or 'igemm' in r['Name'] or 'copy' in r['Name']): print('%8.1f us x %5s  %s'%(float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:110]))
PY
cd $GRAFT_REPO_ROOT
for cv in 1 0 1; do
PSI_HIP_CONV=$cv python bench.py --workload train_s2 --steps 10 --warmup 3 2>gpurun_out/bench_cv.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PSI_HIP_CONV=$cv', d['ms_per_step'], d['ms_per_step_min'], d.get('roofline',{}).get('frac'))"
done
tail -3 gpurun_out/bench_cv.err
