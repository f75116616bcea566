"""dev: time psi_nn_index_query alone on a fitting-like query set (warm hints).  PSI_HIP_LIB selects a variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from psi_release_amd import ops, synth
B, n, m = 32, 2048, 32768
sc = synth.make_scene(0, m, 16, 64)
rs = np.random.RandomState(0)
# queries: near the scene surface, like contact vertices after a few fitting iterations
base = sc.verts[rs.randint(0, m, (B, n))]
q = torch.tensor((base + rs.normal(0, 0.03, base.shape)).astype(np.float32), device='cuda')
ix = ops.SceneNNIndex(sc.verts)
hint = torch.full((B, n), -1, dtype=torch.int32, device='cuda')
d0, i0 = ix.query(q, hint)
q2 = q + 0.002 * torch.randn_like(q)          # next iteration: bodies moved a little, hints from the previous winners
reps = int(os.environ.get('REPS', 50))
for mode in ('cold', 'warm'):
    ts = []
    for r in range(reps):
        h = hint.clone() if mode == 'warm' else None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record(); d, i = ix.query(q2, h); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(mode, 'median us %.1f' % np.median(ts), 'checksum', int(i.sum()), float(d.sum()))
