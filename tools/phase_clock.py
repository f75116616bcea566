"""dev: read the in-kernel phase clocks of the instrumented head/tail variant (tools/_variants/clock.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['PSI_HIP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'clock.so')
import numpy as np, torch, types
import bench
from psi_release_amd import hip
args = types.SimpleNamespace(batch=32, m=32768, nc=2048, D=256, engine='fused')
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '20']
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
buf = (ctypes.c_ulonglong * 64)()
lib = ctypes.CDLL(os.environ['PSI_HIP_LIB'])
lib.psi_dbg_read(buf)
v = np.array(list(buf), dtype=np.float64) * 0.01    # wall_clock64: 100 MHz -> us
names = {0: 'head start (consumer workgroup)', 1: 'x, loss partials, x-only tail work', 2: 'fc1', 3: 'fc2 slice', 4: 'fc3 partial', 8: 'exchange + sum', 5: 'rotations', 40: '  pose_fwd: rodrigues', 41: '  pose_fwd: chain', 6: '  pose_fwd: stores',
         16: 'tail start', 32: '  pose_bwd: loads', 33: '  pose_bwd: level sweep', 34: '  pose_bwd: local grads + gJ',
         35: '  pose_bwd: g_betas', 18: '  pose_bwd: rest (rodrigues bwd, stores)', 19: 'gs_backward/pca bwd', 20: 'W3^T slice', 21: 'W2^T partial', 25: 'exchange + sum', 22: 'W1^T', 23: 'adam'}
order = [0, 1, 2, 3, 4, 8, 5, 40, 41, 6, 16, 32, 33, 34, 35, 18, 19, 20, 21, 25, 22, 23]
prev = None
for i in order:
    if i in (0, 16, 48): prev = v[i]; print(names[i]); continue
    print('%-44s %6.2f us' % (names[i], v[i] - prev)); prev = v[i]
