"""dev: time ops.bn_act (fused HIP BN + ReLU + skip) against the library path (F.batch_norm + relu + add under MIOpen) on the trunk's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from psi_release_amd import ops
dev = 'cuda'
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for shape in [(128, 64, 64, 64), (128, 64, 32, 32), (128, 128, 16, 16)]:
    x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    res = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(shape[1]).to(dev).train()
    mb = x.numel() * 2 / 1e6
    def lib_f():
        return torch.relu(bn(x) + res)
    def hip_f():
        return ops.bn_act(x, bn, relu=True, residual=res)
    def lib_fb():
        y = torch.relu(bn(x) + res); y.backward(g); x.grad = None; res.grad = None
    def hip_fb():
        y = ops.bn_act(x, bn, relu=True, residual=res); y.backward(g); x.grad = None; res.grad = None
    with torch.no_grad():
        tl, th = timeit(lib_f), timeit(hip_f)
    tlb, thb = timeit(lib_fb), timeit(hip_fb)
    print('%s map %.1f MB: fwd lib %.1f us  hip %.1f us (%.0f GB/s over 4 passes) | fwd+bwd lib %.1f us  hip %.1f us' % (shape, mb, tl, th, 4 * mb / th * 1e3, tlb, thb))
