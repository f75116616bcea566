import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ['PSI_HIP_LIB'] = os.path.join(ROOT, 'tools', '_variants', 'kdstats.so')
import numpy as np, torch
from psi_release_amd import ops, hip, synth
L = hip.lib()
scene = synth.make_scene(0, 32768, 16, 2048)
idx = ops.SceneNNIndex(scene.verts, 'cuda')
rs = np.random.RandomState(0)
x = torch.tensor((rs.standard_normal((32, 2048, 3)) * 0.5).astype(np.float32), device='cuda')
hint = torch.full((32, 2048), -1, dtype=torch.int32, device='cuda')
z = (ctypes.c_ulonglong * 4)()
for tag, xx in (('cold', x), ('warm same', x), ('warm moved 1cm', x + 0.01), ('warm moved 5cm', x + 0.05)):
    L.psi_kd_stats(z, 1); torch.cuda.synchronize()
    idx.query(xx, hint=hint); torch.cuda.synchronize()
    L.psi_kd_stats(z, 0)
    nq = 32 * 2048
    print('%-16s nodes/query %.1f  leaves/query %.1f  wave-iterations/wave %.1f' % (tag, z[0] / nq, z[1] / nq, z[2] / (nq / 64)))
