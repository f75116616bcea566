"""dev: kernel time of ops.conv3x3 (forward, and forward + input gradient) against the library convolution on the trunk's shapes, both
replayed from a HIP graph of 10 repeats.  -> gpurun_out/conv_times.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from psi_release_amd import ops
DEV = 'cuda'
out = []
for name, N, Cin, Cout, H, W in [('layer1 64->64 @32x32', 128, 64, 64, 32, 32), ('layer2 128->128 @16x16', 128, 128, 128, 16, 16)]:
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(N, Cin, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(N, Cout, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = conv.weight.detach().to(torch.bfloat16)
    def hip_f():
        with torch.no_grad():
            return ops.conv3x3(x, conv)
    def lib_f():
        with torch.no_grad():
            return F.conv2d(x, wb, None, 1, 1)
    def hip_fb():
        return torch.autograd.grad(ops.conv3x3(x, conv), x, g)
    def lib_fb():
        return torch.autograd.grad(F.conv2d(x, wb, None, 1, 1), x, g)
    res = {}
    for tag, fn in (('hip_fwd', hip_f), ('lib_fwd', lib_f), ('hip_fwd_dx', hip_fb), ('lib_fwd_dx', lib_fb)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(10):
                    fn()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10):
                gr.replay()
            e1.record(s); e1.synchronize()
            res[tag] = round(e0.elapsed_time(e1) / 100 * 1e3, 2)
    flops = 2.0 * N * H * W * Cout * Cin * 9
    res.update(layer=name, GFLOP=round(flops * 1e-9, 2), hip_fwd_TFLOPs=round(flops / res['hip_fwd'] * 1e-6, 1), lib_fwd_TFLOPs=round(flops / res['lib_fwd'] * 1e-6, 1))
    out.append(res)
    print(json.dumps(res), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/conv_times.json', 'w'), indent=1)
