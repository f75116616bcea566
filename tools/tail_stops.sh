cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/stops.so
for k in 1 21 22 23 24 2 3; do
  PSI_TAIL_STOP=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); kb=d.get('kernel_bandwidth',{}); print('tail stop $k', 'head_bwd_adam', kb['head_bwd_adam_kernel']['us'])"
done
