"""dev: workgroup timeline of the last fwd_scene launch (needs tools/_variants/stops.so, built with -DPSI_HEAD_STOPS).

usage (GPU box): PSI_HIP_LIB=tools/_variants/stops.so PSI_SKIN_STOP=9 python tools/timeline.py [bench args]
Prints, per kind of workgroup (1 = NN search, 0 = skinning + SDF): count, start-time spread, duration quantiles, and a coarse
histogram of how many workgroups were resident over the launch."""
import ctypes, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--secondary', '0'])
bench.main()

from psi_release_amd import hip
lib = hip.lib() if callable(getattr(hip, 'lib', None)) else hip._lib
N = 8192
buf = (ctypes.c_ulonglong * (4 * N))()
rc = lib.psi_dbg_timeline(buf, N)
sb = (ctypes.c_int * (4 * N))()
rc |= lib.psi_dbg_kd_stat(sb, N)
mb = (ctypes.c_ulonglong * (4 * N))()
rc |= lib.psi_dbg_kd_mark(mb, N)
marks = np.frombuffer(mb, dtype=np.uint64).reshape(N, 4).astype(np.int64)
a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 4).astype(np.int64)
marks = marks[a[:, 1] > 0]
stat = np.frombuffer(sb, dtype=np.int32).reshape(N, 4)[a[:, 1] > 0]
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01          # us
print('rc', rc, 'workgroups', len(a), 'span us', en.max())
for k in (1, 0):
    m = a[:, 3] == k
    if not m.any(): continue
    d = en[m] - st[m]
    print('kind', k, 'n', int(m.sum()), 'start min/med/max %.2f %.2f %.2f' % (st[m].min(), np.median(st[m]), st[m].max()),
          'dur min/med/p90/max %.2f %.2f %.2f %.2f' % (d.min(), np.median(d), np.quantile(d, 0.9), d.max()), 'end max %.2f' % en[m].max())
xcc = (a[:, 2] >> 32) & 0xf
hw = a[:, 2] & 0xffffffff
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
cuid = xcc * 1000 + se * 100 + sh * 50 + cu
print('distinct CUs', len(np.unique(cuid)), 'workgroups per CU min/max', np.bincount(np.unique(cuid, return_inverse=True)[1]).min(),
      np.bincount(np.unique(cuid, return_inverse=True)[1]).max(), 'per XCC', np.bincount(xcc))
T = np.arange(0, en.max(), 1.0)
for k in (1, 0):
    m = a[:, 3] == k
    print('resident kind', k, ' '.join('%d' % int(((st[m] <= t) & (en[m] > t)).sum()) for t in T))

if (a[:, 3] == 1).any():
    m = a[:, 3] == 1
    d = en[m] - st[m]
    S = stat[m]
    print('search workgroups: tree-walk queries per workgroup (of 64) mean %.1f max %d; workgroups with none %d' % (S[:, 0].mean(), S[:, 0].max(), int((S[:, 0] == 0).sum())))
    print('most visits of one query: quantiles', np.quantile(S[:, 1], [0, .5, .9, .99, 1]), ' grid rounds (max per lane)', np.quantile(S[:, 3], [0, .5, .9, .99, 1]))
    order = np.argsort(d)
    for lo_, hi_ in ((0, 0.1), (0.45, 0.55), (0.9, 1.0), (0.99, 1.0)):
        sel = order[int(lo_ * len(d)):max(int(hi_ * len(d)), int(lo_ * len(d)) + 1)]
        print('duration quantile %.2f-%.2f: dur %.2f us, tree queries %.1f, max visits %.1f, visits summed %.1f, grid rounds %.1f' % (
            lo_, hi_, d[sel].mean(), S[sel, 0].mean(), S[sel, 1].mean(), S[sel, 2].mean(), S[sel, 3].mean()))
    print('corr(dur, max visits) %.3f  corr(dur, tree queries) %.3f  corr(dur, grid rounds) %.3f' % (np.corrcoef(d, S[:, 1])[0, 1], np.corrcoef(d, S[:, 0])[0, 1], np.corrcoef(d, S[:, 3])[0, 1]))

    M = (marks[m] - a[m, 0:1]) * 0.01                      # us since the workgroup's start
    okm = (marks[m] > 0).all(axis=1)
    print('search phases (thread 0; us since workgroup start; median / p90): query point %.2f / %.2f, cell ranges in %.2f / %.2f, scan done %.2f / %.2f, '
          'search done %.2f / %.2f, end %.2f / %.2f' % (tuple(x for k in range(4) for x in (np.median(M[okm, k]), np.quantile(M[okm, k], .9))) + (np.median(d), np.quantile(d, .9))))

if hasattr(lib, 'psi_dbg_kd_reason'):
    rb = (ctypes.c_int * 8)()
    lib.psi_dbg_kd_reason(rb)
    print('tree-walk reasons summed over ALL launches of the run: too many candidate columns %d (largest %d), too many surviving columns %d (most %d), tie flag %d, no warm candidate %d' % (rb[0], rb[4], rb[1], rb[5], rb[2], rb[3]))
