cd $GRAFT_REPO_ROOT
for v in 1 0; do
for fb in 1 0; do
PSI_DP_SIDE_STREAM=$v PSI_FIT_FUSED_BWD=$fb PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('side_stream=$v fused_bwd=$fb', d['ms_per_step'], d['steady_state']['ms_per_step'], d['config']['dp_launch_mode'])"
done; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('single process', d['ms_per_step'], d['steady_state']['ms_per_step'])"
