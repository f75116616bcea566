#!/bin/bash
# A/B of the multi-block BN finalize: train_s2 step time with the previous library and the new one, plus the BN / trunk parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bnorm_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
for l in base_fin1.so default; do
  if [ "$l" = default ]; then unset PSI_HIP_LIB; else export PSI_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/$l; fi
  timeout 300 python bench.py --workload train_s2 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$l', d['ms_per_step'])"
done; done
