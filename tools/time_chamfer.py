"""Development tool: time psi_chamfer_forward variants on the GPU (A/B of compiler flags), BASELINE shape."""
import ctypes, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psi_release_amd import hip

def load(path):
    l = ctypes.CDLL(path)
    for name, (res, args) in hip.SIGNATURES.items():
        if hasattr(l, name):
            getattr(l, name).restype = res; getattr(l, name).argtypes = args
    return l

def main():
    B, n, m = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 2048, 32768))]
    rs = np.random.RandomState(0)
    x = torch.tensor(rs.uniform(-1.5, 1.5, (B, n, 3)).astype(np.float32), device='cuda')
    y = torch.tensor(rs.uniform(-1.5, 1.5, (B, m, 3)).astype(np.float32), device='cuda')
    d = torch.zeros(B, n, device='cuda'); i = torch.zeros(B, n, dtype=torch.int32, device='cuda')
    libs = {'default': hip.LIB_PATH}
    vdir = os.path.join(ROOT, 'tools', '_variants')
    if os.path.isdir(vdir):
        for f in sorted(os.listdir(vdir)):
            if f.endswith('.so'): libs[f[:-3]] = os.path.join(vdir, f)
    res = {}
    for name, path in libs.items():
        l = load(path)
        ws = torch.empty(l.psi_chamfer_workspace_bytes(B, n, m), dtype=torch.uint8, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(60):                      # long warm-up: the first library timed used to look ~10 % slower (clock ramp)
            l.psi_chamfer_forward(x.data_ptr(), y.data_ptr(), B, n, m, d.data_ptr(), i.data_ptr(), None, None, ws.data_ptr(), st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                l.psi_chamfer_forward(x.data_ptr(), y.data_ptr(), B, n, m, d.data_ptr(), i.data_ptr(), None, None, ws.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = sorted(ts)[len(ts) // 2]
        pairs = B * n * m
        res[name] = (t, i.clone())
        print('%-24s %.3f ms  %.2f Gpairs/s  %.1f TFLOP/s(8 flop/pair)' % (name, t, pairs / t * 1e-6, pairs * 8 / t * 1e-9))
    ref = res['default'][1]
    for name, (t, idx) in res.items():
        print(name, 'idx equal to default:', bool(torch.equal(idx, ref)))

if __name__ == '__main__':
    main()
