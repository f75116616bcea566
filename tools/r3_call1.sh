#!/bin/bash
# round 3, GPU call 1: new configs[3]/[4] oracle tests, DP loop issued from C (1-rank RCCL), dist tests, DP bench line
mkdir -p gpurun_out/r3b
python -m pytest tests/test_configs_gpu.py tests/test_dist_gpu.py -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
tail -15 gpurun_out/r3b/pytest.log
PSI_FORCE_DP_PATH=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --secondary 0 --no-cpu-baseline > gpurun_out/r3b/bench_dp1.json 2> gpurun_out/r3b/bench_dp1.err
tail -c 1500 gpurun_out/r3b/bench_dp1.json; tail -5 gpurun_out/r3b/bench_dp1.err
PSI_FORCE_DP_PATH=1 PSI_DP_PYTHON_LOOP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --secondary 0 --no-cpu-baseline > gpurun_out/r3b/bench_dp1_pyloop.json 2> gpurun_out/r3b/bench_dp1_pyloop.err
tail -c 400 gpurun_out/r3b/bench_dp1_pyloop.json
python bench.py --secondary 0 --no-cpu-baseline > gpurun_out/r3b/bench_single.json 2>/dev/null; tail -c 300 gpurun_out/r3b/bench_single.json
