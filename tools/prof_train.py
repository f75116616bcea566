"""dev: torch.profiler breakdown of steady-state train_s2 steps (GPU box)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
args = types.SimpleNamespace(batch=32, m=32768, D=256, nc=2048, warmup=5, steps=5, bf16=int(os.environ.get('BF16', '1')))
# reuse bench_train_s2's setup by monkeypatching the timed loop: simplest is to run it under the profiler after a warm run
bench.bench_train_s2(args)                         # warm (MIOpen find etc. cached in-process)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    args.warmup = 2; args.steps = 10
    bench.bench_train_s2(args)
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=70))
