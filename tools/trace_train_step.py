"""dev: attribute the small kernels of one train_s2 step (batch 128, bf16 trunk, eager) to the psi_release_amd source line that issued them.
-> gpurun_out/train_step_ops.txt"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench

ns = argparse.Namespace(batch=128, m=20000, D=256, nc=1121, bf16=1, graph=0, steps=3, warmup=3, repeats=1, min_timed_s=0.0)
captured = {}
import psi_release_amd.training as training
orig = training.TrainOPS2.train_step
steps = []

def hook(self, data, ep):
    captured['op'] = self
    captured['data'] = data
    return orig(self, data, ep)

training.TrainOPS2.train_step = hook
bench.conv_kernel_roofline = lambda *a, **k: {}
try:
    bench.bench_train_s2(ns)
except Exception as e:
    print('bench leg ended with', repr(e))
op, data = captured['op'], captured['data']
training.TrainOPS2.train_step = orig
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        op.train_step(data, ep=9)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    self_dt = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
    if self_dt <= 0:
        continue
    where = '?'
    for fr in (ev.stack or []):
        if 'psi-release_amd' in fr or 'psi_release_amd' in fr:
            where = fr.split('/')[-1]
            break
    k = (ev.name, where)
    agg[k][0] += 1
    agg[k][1] += self_dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/train_step_ops.txt', 'w') as f:
    tot = sum(v[1] for _, v in rows)
    f.write('total self device us over 2 steps: %.0f\n' % tot)
    for (name, where), (n, t) in rows[:150]:
        f.write('%8.0f us %5d  %-40s %s\n' % (t, n, name[:40], where))
print(open('gpurun_out/train_step_ops.txt').read()[:6000])
