"""dev (GPU box): the general convolution kernels (csrc/conv_gemm.hip) at the trunk's shapes, batch 128: forward, input gradient, weight gradient;
three-term products on fp32 maps and one-term on bf16 maps, against the library (MIOpen) in the same precision.  PSI_CONV_BM=64|128 picks the
pixel tile (read once per process)."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psi_release_amd import ops
DEV = 'cuda'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(2, 64, 7, 2, 3, 128), (64, 64, 3, 1, 1, 32), (64, 128, 3, 2, 1, 32), (64, 128, 1, 2, 0, 32), (128, 128, 3, 1, 1, 16), (128, 32, 3, 1, 1, 16)]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
print('PSI_CONV_BM', os.environ.get('PSI_CONV_BM', 'default'), 'N', N)
for Cin, Cout, K, s, p, H in SHAPES:
    for dt, nterm in ((torch.float32, 3), (torch.bfloat16, 1)):
        conv = torch.nn.Conv2d(Cin, Cout, K, s, p, bias=False).to(DEV).to(memory_format=torch.channels_last)
        x = torch.randn(N, Cin, H, H, device=DEV).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(Cin > 2)
        OH = (H + 2 * p - K) // s + 1
        g = torch.randn(N, Cout, OH, OH, device=DEV).to(dt).contiguous(memory_format=torch.channels_last)
        gflop = 2.0 * N * OH * OH * Cout * Cin * K * K * 1e-9
        with torch.no_grad():
            t_f = timeit(lambda: ops.conv2d_split(x, conv, nterm=nterm, out_bf16=dt == torch.bfloat16))
            wl = conv.weight.to(dt)
            t_fl = timeit(lambda: F.conv2d(x, wl, None, s, p))
        def bwd(mine):
            os.environ['PSI_HIP_CONV2_BWD'] = '1' if mine else '0'
            y = ops.conv2d_split(x, conv, nterm=nterm, out_bf16=dt == torch.bfloat16)
            def f():
                conv.weight.grad = None
                if x.grad is not None: x.grad = None
                y.backward(g, retain_graph=True)
            return timeit(f)
        t_b, t_bl = bwd(True), bwd(False)
        print('%3d->%3d k%d s%d @%3d %s: fwd %6.1f us (%5.0f TF eff; library %6.1f) | dgrad+wgrad %6.1f us (library %6.1f)' % (
            Cin, Cout, K, s, H, 'fp32x3' if nterm == 3 else 'bf16  ', t_f, gflop / t_f * 1e3, t_fl, t_b, t_bl))
