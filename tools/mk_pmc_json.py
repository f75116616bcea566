"""dev: gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE}.txt (tools/pmc.sh) -> profiles/<round>_pmc_{fetch_size,write_size}.csv + _pmc_traffic.json"""
import json, sys, shutil
tag, rnd = sys.argv[1], sys.argv[2]
X2 = ('bwd_joint_kernel', 'blend_bwd_kernel', 'blend_fwd_kernel', 'head_bwd_adam_kernel', 'head_fwd_kernel', 'skin_bwd_A_kernel')   # 16 B/lane streams
def rd(c):
    out = {}
    for l in open('gpurun_out/pmc_%s_%s.txt' % (tag, c)).read().splitlines()[1:]:
        k, n, v = l.rsplit(',', 2)
        k = k.split('<')[0].replace('psi_', '').replace('blend_fwd_h_kernel', 'blend_fwd_kernel').replace('fit_bwd_joint_kernel', 'bwd_joint_kernel').replace('fit_reduce_kernel', 'reduce_partials_kernel')
        if k.startswith('skin_fwd_kernel'): k = 'skin_fwd_sdf_kernel' if 'SdfPen' in l else 'skin_fwd_kernel'
        if k.startswith('skin_bwd_v_kernel'): k = 'skin_bwd_v_grad_kernel' if 'FitGrad' in l else 'skin_bwd_v_kernel'
        out[k] = float(v)
    return out
f, w = rd('FETCH_SIZE'), rd('WRITE_SIZE')
doc = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) of `bench.py --steps 20 --warmup 3`, MI355X, B=32 "
       "default shape; KB per launch averaged over all launches.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports exactly "
       "half of a wide (16 B/lane) coalesced stream: kernels listed with fetch_x2=true are doubled; other access widths and WRITE_SIZE are "
       "uncalibrated.  bytes = (fetch_kb * (2 if fetch_x2 else 1) + write_kb) * 1024.")
d = {'_doc': doc}
for k in sorted(f):
    if k.startswith('at::') or k.startswith('__amd'):
        continue
    x2 = k in X2
    d[k] = {'fetch_kb': f[k], 'write_kb': w.get(k, 0.0), 'fetch_x2': x2, 'bytes': int((f[k] * (2 if x2 else 1) + w.get(k, 0.0)) * 1024)}
json.dump(d, open('profiles/%s_pmc_traffic.json' % rnd, 'w'), indent=1)
shutil.copy('gpurun_out/pmc_%s_FETCH_SIZE.txt' % tag, 'profiles/%s_pmc_fetch_size.csv' % rnd)
shutil.copy('gpurun_out/pmc_%s_WRITE_SIZE.txt' % tag, 'profiles/%s_pmc_write_size.csv' % rnd)
print(json.dumps(d, indent=1)[:1500])
