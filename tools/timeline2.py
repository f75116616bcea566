"""dev: workgroup timeline of the last bwd_joint launch (needs a -DPSI_HEAD_STOPS build of lbs.hip, tools/_variants/stops2.so).
usage (GPU box): PSI_HIP_LIB=tools/_variants/stops2.so python tools/timeline2.py [bench args]"""
import ctypes, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--secondary', '0'])
bench.main()
from psi_release_amd import hip
lib = hip.lib()
N = 2048
buf = (ctypes.c_ulonglong * (4 * N))()
rc = lib.psi_dbg_timeline2(buf, N)
a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01
print('rc', rc, 'workgroups', len(a), 'span us', en.max())
for k in (1, 0):
    m = a[:, 3] == k
    if not m.any(): continue
    d = en[m] - st[m]
    print('kind', k, '(1 = blend_bwd stream, 0 = skin_bwd_A)', 'n', int(m.sum()), 'start min/med/max %.2f %.2f %.2f' % (st[m].min(), np.median(st[m]), st[m].max()),
          'dur min/med/p90/max %.2f %.2f %.2f %.2f' % (d.min(), np.median(d), np.quantile(d, 0.9), d.max()), 'end min/med/max %.2f %.2f %.2f' % (en[m].min(), np.median(en[m]), en[m].max()))
xcc = (a[:, 2] >> 32) & 0xf
for x in range(8):
    m = (xcc == x) & (a[:, 3] == 1)
    if m.any(): print('xcc', x, 'stream workgroups', int(m.sum()), 'dur med %.2f max %.2f' % (np.median(en[m] - st[m]), (en[m] - st[m]).max()))

mb = (ctypes.c_ulonglong * (2 * N))()
lib.psi_dbg_ska_marks(mb, N)
mk = np.frombuffer(mb, dtype=np.uint64).reshape(N, 2).astype(np.int64)
a0 = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
sel = (a0[:, 1] > 0) & (a0[:, 3] == 0) & (mk[:, 0] > 0) & (mk[:, 1] > 0)
print('skin_bwd_A workgroups: operands staged at %.2f us (median since workgroup start), first body done at %.2f, end %.2f' % (
    np.median((mk[sel, 0] - a0[sel, 0]) * 0.01), np.median((mk[sel, 1] - a0[sel, 0]) * 0.01), np.median((a0[sel, 1] - a0[sel, 0]) * 0.01)))
