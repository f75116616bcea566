"""dev: torch.profiler breakdown of steady-state train_s2 steps by section (GPU box, eager mode)."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity, record_function
from psi_release_amd import training, geometry, ops, body_model, vposer, models


def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function('SEC_' + label):
            return f(*a, **k)
    setattr(obj, name, staticmethod(g) if isinstance(obj, type) and isinstance(obj.__dict__.get(name), staticmethod) else g)


GT = geometry.GeometryTransformer
for n in ('normalize_global_T', 'convert_to_6D_rot', 'convert_to_3D_rot', 'recover_global_T'):
    wrap(GT, n, n)
wrap(geometry.BodyParamParser, 'body_params_encapsulate_batch', 'encapsulate')
wrap(vposer.VPoser, 'decode', 'vposer_decode')
wrap(body_model.SMPLXLayer, 'forward', 'smplx')
wrap(ops, 'chamfer_to_scenes', 'chamfer')
wrap(ops, 'sdf_sample', 'sdf')
wrap(ops, 'penetration_loss', 'pen')
wrap(models.HumanCVAES2, 'forward', 'cvae_forward')
wrap(training.TrainOPS2, '_losses_from_batch', 'FWD_ALL')
args = types.SimpleNamespace(batch=32, m=32768, D=256, nc=2048, warmup=5, steps=5, bf16=1, graph=0)
bench.bench_train_s2(args)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    args.warmup = 2; args.steps = 10
    bench.bench_train_s2(args)
ka = prof.key_averages()
rows = [(e.key, e.count, e.device_time_total / 1e3) for e in ka if e.key.startswith('SEC_') or 'Backward' in e.key or 'Optimizer' in e.key]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:40]:
    print('%-50s calls %5d  cuda_total %8.2f ms  per step %7.3f ms' % (k[:50], c, t, t / 12))
print('total self cuda ms', sum(e.self_device_time_total for e in ka) / 1e3)
