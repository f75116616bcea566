"""dev (GPU box): ONE convolution shape through ops.conv2d_split a few times — the command behind counter collections of the general
convolution kernels (tools/pmc2.sh).  usage: conv_one.py Cin Cout K stride pad H [nterm] [what=fwd|bwd]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psi_release_amd import ops
Cin, Cout, K, s, p, H = [int(a) for a in sys.argv[1:7]]
nterm = int(sys.argv[7]) if len(sys.argv) > 7 else 3
what = sys.argv[8] if len(sys.argv) > 8 else 'fwd'
dt = torch.float32 if nterm == 3 else torch.bfloat16
conv = torch.nn.Conv2d(Cin, Cout, K, s, p, bias=False).cuda().to(memory_format=torch.channels_last)
x = torch.randn(128, Cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(what == 'bwd')
for _ in range(10):
    y = ops.conv2d_split(x, conv, nterm=nterm, out_bf16=dt == torch.bfloat16)
    if what == 'bwd':
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
