"""dev: HIP-event timing of the hand-written dense-layer kernels (ops.linear_act) at the CVAE shapes, next to torch's bf16 path.
-> gpurun_out/linear_times.json  (achieved weight-stream GB/s and bf16 TFLOP/s per kernel)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from psi_release_amd import ops

DEV = 'cuda'
shapes = [('ResBlock 512 (S1 / trans_vae encode)', 128, 512, 512), ('ResBlock 768 (pose_vae encode)', 128, 768, 768),
          ('fc 8192->256 (S1 / trans_vae scene feature)', 128, 256, 8192), ('fc 32768->256 (pose_vae scene feature)', 128, 256, 32768)]


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


out = []
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    gy = torch.randn(M, N, device=DEV)
    xr, Wr = x.clone().requires_grad_(), W.clone().requires_grad_()
    t_f = timeit(lambda: ops.linear_act(x, W, b, 'leaky_relu', 0.01, residual=None))
    y = ops.linear_act(xr, Wr, b, 'leaky_relu', 0.01)
    t_fb = timeit(lambda: torch.autograd.grad(ops.linear_act(xr, Wr, b, 'leaky_relu', 0.01), (xr, Wr), gy))

    def torch_fwd():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            return F.leaky_relu(F.linear(x, W, b), 0.01).float()
    t_tf = timeit(torch_fwd)

    def torch_fb():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            yy = F.leaky_relu(F.linear(xr, Wr, b), 0.01).float()
        return torch.autograd.grad(yy, (xr, Wr), gy)
    t_tfb = timeit(torch_fb)
    flops = 2.0 * M * N * K
    wbytes = 4.0 * N * K + 4.0 * M * K + 4.0 * M * N
    out.append({'layer': name, 'M': M, 'N': N, 'K': K, 'hip_fwd_us': round(t_f * 1e6, 2), 'hip_fwd_bwd_us': round(t_fb * 1e6, 2),
                'torch_bf16_fwd_us': round(t_tf * 1e6, 2), 'torch_bf16_fwd_bwd_us': round(t_tfb * 1e6, 2),
                'hip_fwd_GBps': round(wbytes / t_f * 1e-9, 1), 'hip_fwd_frac_hbm': round(wbytes / t_f * 1e-9 / 8000, 4),
                'hip_fwd_TFLOPs': round(flops / t_f * 1e-12, 2), 'hip_fwd_frac_bf16_peak': round(flops / t_f * 1e-12 / 2500, 5)})
    print(json.dumps(out[-1]), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/linear_times.json', 'w'), indent=1)
