"""dev: per-kernel event timings of one fused fitting iteration (PSI_HIP_LIB selects a variant)."""
import sys, os, json, subprocess
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), '--no-cpu-baseline'] + sys.argv[1:],
                   capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print(d['ms_per_step'], {k: round(v, 1) for k, v in d['kernels_us'].items()})
