"""dev: workgroup timeline of the last fit_bwd_joint_kernel launch (needs the -DPSI_HEAD_STOPS build, tools/_variants/stops.so).
usage (GPU box): PSI_HIP_LIB=tools/_variants/stops.so PSI_SKIN_STOP=11 python tools/timeline_joint.py [bench args]"""
import ctypes, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--secondary', '0'])
bench.main()
from psi_release_amd import hip
lib = hip.lib()
N = 8192
buf = (ctypes.c_ulonglong * (4 * N))()
rc = lib.psi_dbg_timeline(buf, N)
a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
print('rc', rc, 'workgroups', len(a), 'span us %.2f' % en.max())
names = {0: 'skin_bwd_A model slices', 1: 'skin_bwd_A contact slices', 2: 'blend_bwd model columns', 3: 'blend_bwd contact columns', 4: 'statistics'}
for k in range(5):
    m = a[:, 3] == k
    if not m.any():
        continue
    d = en[m] - st[m]
    print('%-28s n %4d start min/med/max %5.2f %5.2f %5.2f  dur min/med/p90/max %5.2f %5.2f %5.2f %5.2f  end med/max %5.2f %5.2f'
          % (names[k], int(m.sum()), st[m].min(), np.median(st[m]), st[m].max(), d.min(), np.median(d), np.quantile(d, 0.9), d.max(), np.median(en[m]), en[m].max()))
