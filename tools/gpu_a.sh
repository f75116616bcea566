#!/bin/bash
# round-2 first GPU pass: suite, bench protocol, self-spawned ranks (gloo on one GPU), habitat workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/a; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "n1 rc=$?"
( time timeout 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_n1_k100.json 2> $O/bench_n1_k100.err; echo "n1k100 rc=$?"
( time PSI_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "n2 rc=$?"
( time PSI_FORCE_DP_PATH=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --secondary 0 ) > $O/bench_dp1_nccl.json 2> $O/bench_dp1_nccl.err; echo "dp1 rc=$?"
( time timeout 300 python bench.py --workload fitting_habitat --steps 21 --warmup 7 --cpu-seconds 6 ) > $O/bench_habitat.json 2> $O/bench_habitat.err; echo "hab rc=$?"
tail -c 600 $O/bench_n1.json; echo; tail -c 300 $O/bench_n2_gloo.json; tail -3 $O/bench_n2_gloo.err; tail -c 300 $O/bench_habitat.json; tail -3 $O/bench_habitat.err
