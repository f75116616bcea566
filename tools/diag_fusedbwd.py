"""dev (GPU box): first-iteration gradient of the fused engine against the modular (autograd) engine, per block of the 75-D body vector."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests')); sys.path.insert(0, os.path.join(ROOT,'oracle'))
from psi_release_amd import fitting, synth
from test_fitting_gpu import make_op
smplx, vp = synth.make_smplx(7), synth.make_vposer_state(3)
B = int(os.environ.get('DIAG_B', 3))
scene = synth.make_scene(3, 3000, 24, int(os.environ.get('DIAG_NC', 300)))
bodies = synth.make_bodies(21, B); bodies['cam_ext'] = synth.make_cam_ext(7, B)
ops = {e: make_op(smplx, vp, scene, B, e, num_iter=1, lr=0.05) for e in ('modular','fused')}
for e,o in ops.items(): o.use_graph = False
runners = {e: o.make_step_runner(dict(bodies)) for e,o in ops.items()}
for e in runners: runners[e].step()
runners['fused'].finish()
gm = ops['modular'].xhr_rec.grad.cpu().numpy()
mf = ops['fused']._fused.buffer('adam_m', (B,75)).cpu().numpy()/0.1
print('losses m', runners['modular'].last_losses(), 'f', runners['fused'].last_losses())
for nm, lo, hi in (('transl',0,3),('rot6d',3,9),('betas',9,19),('latent',19,51),('lhand',51,63),('rhand',63,75)):
    d = np.abs(gm[:,lo:hi]-mf[:,lo:hi]); print('%-7s max|g| %.3e  max err %.3e  per body %s' % (nm, np.abs(gm[:,lo:hi]).max(), d.max(), d.max(1)))
# ---- intermediate buffers: fused-backward engine against an engine built with PSI_FIT_FUSED_BWD=0
os.environ['PSI_FIT_FUSED_BWD'] = '0'
op0 = make_op(smplx, vp, scene, B, 'fused', num_iter=1, lr=0.05); op0.use_graph = False
r0 = op0.make_step_runner(dict(bodies)); r0.step(); r0.finish()
e1, e0 = ops['fused']._fused, op0._fused
V, Vpad, JP, Kpad = 10475, 10496, 64, 512
for nm, shape in (('gA', (B, JP, 16)), ('gfeat', (B, Kpad)), ('g_transl', (B, 3))):
    a, b = e1.buffer(nm, shape).cpu().numpy(), e0.buffer(nm, shape).cpu().numpy()
    print(nm, 'max|ref| %.3e max err %.3e' % (np.abs(b).max(), np.abs(a - b).max()), 'per body', np.abs(a - b).reshape(B, -1).max(1))
st = e0.buffer('stats', (8,)).cpu().numpy(); N = st[4]; sp = -0.5 / N if N > 0 else 0.0
print('stats', st, 'sp', sp)
gl1, gl0 = e1.buffer('gl', (B, 3 * Vpad)).cpu().numpy(), e0.buffer('gl', (B, 3 * Vpad)).cpu().numpy()
gv1, gv0 = e1.buffer('g_vp', (B, 3 * Vpad)).cpu().numpy(), e0.buffer('g_vp', (B, 3 * Vpad)).cpu().numpy()
nc = int(os.environ.get('DIAG_NC', 300)); ncp = (nc + 255) // 256 * 256
glc, gvc, vpc = (e1.buffer(n_, (B, 3 * ncp)).cpu().numpy() for n_ in ('glc', 'gvpc', 'vpc'))
vid = ops['fused'].contact_vertex_ids().cpu().numpy()
tot_l, tot_v = sp * gl1, sp * gv1
for s_, v in enumerate(vid):
    tot_l[:, 3 * v:3 * v + 3] += glc[:, 3 * s_:3 * s_ + 3]; tot_v[:, 3 * v:3 * v + 3] += gvc[:, 3 * s_:3 * s_ + 3]
print('g_local  total: max|ref| %.3e err %.3e' % (np.abs(gl0).max(), np.abs(tot_l - gl0).max()))
print('g_vposed total: max|ref| %.3e err %.3e' % (np.abs(gv0).max(), np.abs(tot_v - gv0).max()))
vp0 = e0.buffer('v_posed', (B, 3 * Vpad)).cpu().numpy()
print('vpc err %.3e' % max(np.abs(vpc[:, 3 * s_:3 * s_ + 3] - vp0[:, 3 * v:3 * v + 3]).max() for s_, v in enumerate(vid)), 'padding slots zero:', float(np.abs(glc[:, 3 * nc:]).max()))
mask = np.zeros(Vpad, bool); mask[vid] = True
m3 = np.repeat(mask, 3)
print('g_local err: non-contact vertices %.3e (max|ref| %.3e), contact vertices %.3e (max|ref| %.3e)' % (
    np.abs(tot_l - gl0)[:, ~m3].max(), np.abs(gl0[:, ~m3]).max(), np.abs(tot_l - gl0)[:, m3].max(), np.abs(gl0[:, m3]).max()))
w = np.abs(tot_l - gl0); i = np.unravel_index(w.argmax(), w.shape); print('worst entry', i, 'vertex', i[1] // 3, 'is contact', bool(mask[i[1] // 3]), 'ref', gl0[i], 'got', tot_l[i], 'pen part', sp * gl1[i])
gt0, gt1 = e0.buffer('g_transl', (B, 3)).cpu().numpy(), e1.buffer('g_transl', (B, 3)).cpu().numpy()
print('g_transl ref', gt0.ravel(), '\n  sum of ref rows', gl0.reshape(B, -1, 3).sum(1).ravel(), '\n  fused', gt1.ravel(), '\n  sp*sum(pen rows)+sum(contact rows)', (sp * gl1.reshape(B, -1, 3).sum(1) + glc.reshape(B, -1, 3).sum(1)).ravel())
print('dup slots', len(vid) - len(set(vid.tolist())))
