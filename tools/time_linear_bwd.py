"""dev: kernel-level time of the hand-written dense-layer BACKWARD (psi_linear_backward: dX + dW + dbias) against the library route the
autograd function used before (mask, cast, two hipBLASLt bf16 GEMMs, column sum), both replayed from a HIP graph of 20 repeats so that
host launch overhead does not hide the kernels.  -> gpurun_out/linear_bwd_times.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psi_release_amd import hip

DEV = 'cuda'
shapes = [('ResBlock 512', 128, 512, 512), ('ResBlock 768', 128, 768, 768), ('decode 128', 128, 128, 128), ('fc 8192->256', 128, 256, 8192),
          ('fc 32768->256', 128, 256, 32768), ('linear_in 80->256', 128, 256, 80)]
L = hip.lib()
out = []
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5
    gy = torch.randn(M, N, device=DEV); a_out = torch.randn(M, N, device=DEV)
    gx = torch.empty(M, K, device=DEV); gw = torch.empty(N, K, device=DEV); gb = torch.empty(N, device=DEV)
    wsb = torch.empty(max(L.psi_linear_backward_workspace_floats(M, N, K), 1), device=DEV)
    def hip_bwd():
        hip.check(L.psi_linear_backward(hip.ptr(gy), hip.ptr(a_out), hip.ptr(x), 0, hip.ptr(W), M, N, K, 0.01, hip.ptr(gx), hip.ptr(gw), hip.ptr(gb),
                                        hip.ptr(wsb), hip.stream()), 'psi_linear_backward')
    def lib_bwd():
        g = torch.where(a_out > 0, gy, gy * 0.01)
        gb16 = g.to(torch.bfloat16)
        return (gb16 @ W.to(torch.bfloat16)).float(), (gb16.t() @ x.to(torch.bfloat16)).float(), g.sum(0)
    res = {}
    for tag, fn in (('hip', hip_bwd), ('library', lib_bwd)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    fn()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                g.replay()
            e1.record(s)
            e1.synchronize()
            res[tag] = e0.elapsed_time(e1) / 200 * 1e3
    out.append({'layer': name, 'M': M, 'N': N, 'K': K, 'hip_bwd_us': round(res['hip'], 2), 'library_bwd_us': round(res['library'], 2)})
    print(json.dumps(out[-1]), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/linear_bwd_times.json', 'w'), indent=1)
