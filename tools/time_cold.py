"""dev (GPU box): fwd_scene of the FIRST iteration after psi_fit_set_problem(reset) (cold NN hints) against a warm iteration, HIP-event times."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
args = bench.parse(['--no-cpu-baseline']); args.engine_resolved = 'fused'
op, bodies, _ = bench.make_op(args, 0, torch.device('cuda', 0))
runner = op.make_step_runner(bodies)
runner.steps(30); torch.cuda.synchronize()
eng = runner.eng
cold, warm = [], []
for _ in range(15):
    runner.restart(); torch.cuda.synchronize()
    k = dict(eng.profile(1)); cold.append(k['fwd_scene_kernel'] * 1e3)
    runner.steps(5)
    k = dict(eng.profile(1)); warm.append(k['fwd_scene_kernel'] * 1e3)
print('fwd_scene HIP-event us: cold median %.1f (min %.1f max %.1f)  warm median %.1f' % (statistics.median(cold), min(cold), max(cold), statistics.median(warm)))
