"""dev: forward and backward time of the hand-written dense kernels at the layer shapes of the train_s2 model (batch 128), with the weight bytes
each pass has to move and the bandwidth that corresponds to.  -> gpurun_out/linear_shapes.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psi_release_amd import hip

DEV = 'cuda'
shapes = [('ResBlock 1024 (trans encode)', 128, 1024, 1024, 4), ('ResBlock 1536 (pose encode)', 128, 1536, 1536, 4), ('fc 8192->512', 128, 512, 8192, 1),
          ('fc 32768->512', 128, 512, 32768, 1), ('ResBlock 32 (trans decode)', 128, 32, 32, 4), ('ResBlock 128 (pose decode)', 128, 128, 128, 4),
          ('mean/logvar 1024->32', 128, 32, 1024, 2), ('mean/logvar 1536->32', 128, 32, 1536, 2), ('decode.0 544->32', 128, 32, 544, 1),
          ('decode.0 1056->128', 128, 128, 1056, 1)]
L = hip.lib()
out = []
tot = {'fwd': 0.0, 'bwd': 0.0}
for name, M, N, K, count in shapes:
    x = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    gy = torch.randn(M, N, device=DEV); y = torch.empty(M, N, device=DEV)
    gx = torch.empty(M, K, device=DEV); gw = torch.empty(N, K, device=DEV); gb = torch.empty(N, device=DEV)
    wsb = torch.empty(max(L.psi_linear_backward_workspace_floats(M, N, K), 1), device=DEV)
    nws = L.psi_linear_workspace_floats(M, N, K)
    ws = torch.empty(max(nws, 1), device=DEV)
    def fwd():
        hip.check(L.psi_linear_forward(hip.ptr(x), 0, hip.ptr(W), hip.ptr(b), None, M, N, K, 1, 0.01, hip.ptr(y), None, hip.ptr(ws), hip.stream()), 'fwd')
    def bwd():
        hip.check(L.psi_linear_backward(hip.ptr(gy), hip.ptr(y), hip.ptr(x), 0, hip.ptr(W), M, N, K, 0.01, hip.ptr(gx), hip.ptr(gw), hip.ptr(gb), hip.ptr(wsb), hip.stream()), 'bwd')
    res = {}
    for tag, fn in (('fwd', fwd), ('bwd', bwd)):
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
        tot[tag] += res[tag] * count
    wb = N * K * 4.0
    out.append({'layer': name, 'M': M, 'N': N, 'K': K, 'per_step': count, 'weight_MB': round(wb / 1e6, 2), 'fwd_us': round(res['fwd'], 2),
                'bwd_us': round(res['bwd'], 2), 'fwd_TBps': round(wb / res['fwd'] * 1e-6, 2), 'bwd_TBps_of_2x_weight': round(2 * wb / res['bwd'] * 1e-6, 2)})
    print(json.dumps(out[-1]), flush=True)
print('per step: fwd %.0f us, bwd %.0f us' % (tot['fwd'], tot['bwd']))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/linear_shapes.json', 'w'), indent=1)
