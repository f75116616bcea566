"""Sensitivity sweep of the fitting bench: scene size m, contact count n_c, batch B.  -> gpurun_out/sensitivity.json
(ms_per_iter = the median 100-iteration fresh-start loop, the protocol of round 5; ms_per_iter_steady_state = the blocks without restart, what the sweeps
of rounds 2-4 recorded)"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = [dict(), dict(m=8192), dict(m=131072), dict(nc=512), dict(nc=4096), dict(batch=64), dict(batch=128), dict(batch=256), dict(batch=512),
        dict(batch=4096), {'batch': 512, 'weight-nnz': 4}, {'batch': 4096, 'weight-nnz': 4}]
out = []
for r in runs:
    args = [sys.executable, os.path.join(R, 'bench.py'), '--no-cpu-baseline', '--secondary', '0', '--steps', '20' if r.get('batch', 32) >= 512 else '60', '--warmup', '5']
    for k, v in r.items():
        args += ['--' + k, str(v)]
    p = subprocess.run(args, capture_output=True, text=True)
    try:
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
        B = r.get('batch', 32)
        out.append({'B': B, 'weight_nnz': r.get('weight-nnz', 0), 'm': r.get('m', 32768), 'n_c': r.get('nc', 2048), 'ms_per_iter': d['ms_per_step'], 'ms_per_iter_steady_state': (d.get('steady_state') or {}).get('ms_per_step'), 'iters_per_s': d['value'],
                    'body_iters_per_s': round(d['value'] * B, 1), 'kernels_us': d.get('kernels_us'), 'kernel_bandwidth': d.get('kernel_bandwidth'),
                    'iteration_roofline': d.get('iteration_roofline')})
    except Exception as e:
        out.append({'args': r, 'error': (p.stderr or str(e))[-400:]})
    print(json.dumps(out[-1])[:300], flush=True)
json.dump(out, open(os.path.join(R, 'gpurun_out', 'sensitivity.json'), 'w'), indent=1)
