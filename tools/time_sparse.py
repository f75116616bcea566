"""dev: fitting iteration time with SMPL-X-like sparse skinning weights (5 non-zeros per vertex): compressed rows vs dense loop."""
import sys, os, types, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from psi_release_amd import fitting, synth
import bench
def sparsify(data, k=5):
    W = np.array(data.weights, dtype=np.float32)
    keep = np.argsort(-W, axis=1)[:, :k]
    Ws = np.zeros_like(W)
    np.put_along_axis(Ws, keep, np.take_along_axis(W, keep, axis=1), axis=1)
    data.weights = Ws / Ws.sum(axis=1, keepdims=True)
    return data
orig = synth.make_smplx
synth.make_smplx = lambda seed=7, **kw: sparsify(orig(seed, **kw))
for dense in ('1', '0'):
    os.environ['PSI_LBS_DENSE'] = dense
    args = types.SimpleNamespace(batch=32, m=32768, nc=2048, D=256, engine='fused', engine_resolved='fused')
    op, bodies, assets = bench.make_op(args, 0, torch.device('cuda', 0))
    runner = op.make_step_runner(bodies)
    runner.steps(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); runner.steps(200); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    k = runner.eng.profile(20)
    print('PSI_LBS_DENSE=%s  %.4f ms/iter  ' % (dense, dt * 1e3), {n: round(v * 1e3, 1) for n, v in k if 'skin' in n or 'joint' in n})
