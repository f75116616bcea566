"""dev: torch.profiler breakdown of steady-state train_s2 steps (GPU box, eager mode): top ops by device time and by launch count."""
import sys, os, types, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
args = types.SimpleNamespace(batch=32, m=32768, D=256, nc=2048, warmup=5, steps=5, bf16=1, graph=0)
with contextlib.redirect_stdout(io.StringIO()):
    bench.bench_train_s2(args)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    args.warmup = 2; args.steps = 10
    with contextlib.redirect_stdout(io.StringIO()):
        bench.bench_train_s2(args)
ka = [e for e in prof.key_averages() if e.self_device_time_total > 0]
tot = sum(e.self_device_time_total for e in ka)
print('total self device ms per step %.3f' % (tot / 1e3 / 12))
print('--- by device time')
for e in sorted(ka, key=lambda e: -e.self_device_time_total)[:22]:
    print('%-70s calls/step %6.1f  ms/step %6.3f' % (e.key[:70], e.count / 12, e.self_device_time_total / 1e3 / 12))
print('--- by launch count')
for e in sorted(ka, key=lambda e: -e.count)[:14]:
    print('%-70s calls/step %6.1f  ms/step %6.3f' % (e.key[:70], e.count / 12, e.self_device_time_total / 1e3 / 12))
