"""Seeded INPUTS of the section-8(f) fixtures (tests/golden/{generation,plausibility,batchgen}.npz), shared by the script that runs
the reference on them (oracle/make_golden_f.py, this container only) and by the tests that run the product on them (GPU box):
both sides regenerate exactly the same arrays, only the reference's OUTPUTS are stored in the .npz files."""
import numpy as np

from psi_release_amd import synth

GEN_VIEWS = (('wide', (96, 160)), ('tall', (150, 90)))
GEN_SAMPLES = 4
PLAUS_SCENE = dict(seed=4, m=4096, D=32, n_contact=256, radius=1.9)



def gen_views():
    """Synthetic Habitat sensor dumps (cam_*.npy dict, depth_*.npy, seg_*.npy) — regenerated identically by the tests."""
    rs = np.random.RandomState(201)
    views = []
    for i, (tag, (H, W)) in enumerate(GEN_VIEWS):
        depth = rs.uniform(0.3, 9.0, (H, W)).astype(np.float32)
        seg = rs.randint(0, 60, (H, W)).astype(np.float32)
        cam_ext = synth.make_cam_ext(30 + i, 1)[0]
        cam_int = np.array([[500.0 + 10 * i, 0, 320.0], [0, 500.0 + 10 * i, 240.0], [0, 0, 1]], np.float32)
        views.append({'depth': depth, 'seg': seg, 'cam_ext': cam_ext, 'cam_int': cam_int})
    return views



BG_SCENES = ['BasementSittingBooth', 'MPH1Library', 'MPH8', 'MPH11', 'MPH16', 'MPH112', 'N0SittingBooth', 'N0Sofa', 'N3Library',
             'N3Office', 'N3OpenArea', 'Werkraum']          # batch_gen_hdf5.py:103-105 (the scene-id order of the HDF5 files)
BG_M, BG_D = 48, 4


def bg_tables():
    """Two HDF5-shaped tables (placeholder row 0 + samples, utils_convert2hdf5.py:55-61): sample s carries its own id in every
    stream so a batch can be recognised from its contents; one sample of table A violates |z| <= max_d (skip rule)."""
    tabs = []
    for t, n in enumerate((14, 9)):
        rs = np.random.RandomState(300 + t)
        sid = rs.randint(0, len(BG_SCENES), n)
        ids = (100 * (t + 1) + np.arange(n)).astype(np.float32)
        depth = np.tile(ids[:, None, None, None], (1, 1, 128, 128)).astype(np.float32)
        seg = -depth
        body = rs.standard_normal((n, 72)).astype(np.float32)
        body[:, 2] = np.abs(body[:, 2]) + 1.0
        body[:, 0] = ids
        max_d = np.full(n, 6.0, np.float32)
        if t == 0:
            body[5, 2] = 7.5                                     # "wrong prox fitting": |z| > max_d  (batch_gen_hdf5.py:211-214)
        cam_ext = synth.make_cam_ext(40 + t, n)
        cam_int = np.tile(np.array([[1060.0, 0, 951.0], [0, 1060.0, 536.0], [0, 0, 1]], np.float32)[None], (n, 1, 1))
        cam_int[:, 0, 0] = ids
        tab = {'sceneid': sid.astype(np.float32), 'depth': depth, 'seg': seg, 'body': body, 'cam_ext': cam_ext, 'cam_int': cam_int, 'max_d': max_d}
        tabs.append({k: np.concatenate([np.zeros_like(v[:1]), v]) for k, v in tab.items()})
    return tabs


def bg_scenes():
    sc = {}
    for i, name in enumerate(BG_SCENES):
        s = synth.make_scene(100 + i, BG_M, BG_D, 14)
        sc[name] = s
    return sc


def batch_digest(batch):
    """What identifies a batch: the sample ids (stream contents), the scene gather and the shapes of all 12 tensors."""
    if batch is None:
        return None
    d = {'ids': batch[2][:, 0].numpy().copy(), 'depth00': batch[0][:, 0, 0, 0].numpy().copy(), 'seg00': batch[1][:, 0, 0, 0].numpy().copy(),
         'cam_int00': batch[4][:, 0, 0].numpy().copy(), 'cam_ext': batch[3].numpy().copy(), 'max_d': batch[5].numpy().copy(),
         'body': batch[2].numpy().copy(), 's_verts0': batch[6][:, 0, :].numpy().copy(), 's_gmin': batch[8].numpy().copy(),
         's_gmax': batch[9].numpy().copy(), 's_gdim': batch[10].numpy().copy(), 's_sdf000': batch[11][:, 0, 0, :].numpy().copy(),
         'shapes': np.array([list(t.shape) + [0] * (5 - t.dim()) for t in batch], np.int64)}
    return d


