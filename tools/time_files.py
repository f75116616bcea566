"""dev: whole-job throughput of the fitting entry points' file loop at the reference's per-file batch size (1) and iteration count (20,
fitting_proxe.py:235): files/s with K independent files in flight per GPU (FittingOP.fitting_many).  -> gpurun_out/files_per_s.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psi_release_amd import fitting, synth

dev = torch.device('cuda')
scene = synth.make_scene(0, 32768, 256, 2048)
cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None, 'init_lr_h': 0.1, 'num_iter': 20,
       'batch_size': 1, 'device': dev, 'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None, 'verbose': False,
       'smplx_data': synth.make_smplx(7), 'vposer_state': synth.make_vposer_state(3), 'scene': scene, 'engine': 'fused'}
loss = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}
files = []
for i in range(256):
    b = synth.make_bodies(100 + i, 1)
    b['cam_ext'] = synth.make_cam_ext(100 + i, 1)
    files.append(b)
out = {}
op = fitting.FittingOP(cfg, loss)
op.reset_optimizer = True
for K in (1, 2, 4, 8, 16, 32):
    op.fitting_many(files[:2 * K], K)                       # engines + graphs created
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    op.fitting_many(files, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[K] = {'files_per_s': round(len(files) / dt, 1), 'ms_per_file': round(dt / len(files) * 1e3, 3), 'iterations_per_s': round(len(files) * 20 / dt, 1)}
    print(K, out[K], flush=True)
# packed: 32 files per engine run, per-body normalisers (same results as one-by-one fits)
cfg32 = dict(cfg, batch_size=32, independent_bodies=True)
op32 = fitting.FittingOP(cfg32, loss)
many = files * 8                                             # 2048 files
for K in (1, 2, 4):
    op32.fitting_many(many[:64 * K], K)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    op32.fitting_many(many, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['packed32_K%d' % K] = {'files_per_s': round(len(many) / dt, 1), 'ms_per_file': round(dt / len(many) * 1e3, 4), 'iterations_per_s': round(len(many) * 20 / dt, 1)}
    print('packed32', K, out['packed32_K%d' % K], flush=True)
t0 = time.perf_counter()
for f in files[:64]:
    op.fitting(dict(f))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out['sequential_fitting_calls'] = {'files_per_s': round(64 / dt, 1), 'ms_per_file': round(dt / 64 * 1e3, 3)}
print(out['sequential_fitting_calls'])
json.dump(out, open('gpurun_out/files_per_s.json', 'w'), indent=1)
