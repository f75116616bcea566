"""dev (GPU box): HumanCVAES2 forward / training step at batch 128, fp32 model on the hand-written path (PSI_HIP_PRECISE=1) against the
library path (=0), and the bf16 mode with the general convolution kernel on / off (PSI_HIP_CONV2)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psi_release_amd import models, synth
DEV = 'cuda'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
inp = synth.make_cvae_inputs(13, B)
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)
def run(bf16, env, train, reps=10):
    for k, v in env.items(): os.environ[k] = v
    m = models.HumanCVAES2(latentD_g=256, latentD_l=256, n_dim_body=75, autocast_bf16=bf16).to(DEV)
    m.train(train)
    args = (T(inp['x75']), T(inp['eps32']), T(inp['eps32b']), T(inp['xs']))
    def step():
        if train:
            out = m(*args, use_eps=True)
            (out[0].abs().mean() + out[1].pow(2).mean() + out[3].pow(2).mean()).backward()
        else:
            with torch.no_grad(): m(*args, use_eps=True)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for train in (False, True):
    print('B', B, 'train+backward' if train else 'forward (no_grad, train-mode BN off)',
          'fp32 hand-written %.2f ms' % run(False, {'PSI_HIP_PRECISE': '1'}, train), '| fp32 library %.2f ms' % run(False, {'PSI_HIP_PRECISE': '0'}, train),
          '| bf16 conv2 on %.2f ms' % run(True, {'PSI_HIP_CONV2': '1'}, train), '| bf16 conv2 off %.2f ms' % run(True, {'PSI_HIP_CONV2': '0'}, train))
