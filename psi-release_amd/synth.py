"""Seeded synthetic assets with the shapes of the licensed PSI inputs.

None of the assets the reference needs ship with it (SMPL-X ``SMPLX_NEUTRAL.npz``,
VPoser ``vposer_v1_0``, PROX-E scenes, ``data/resnet18.pth`` are licensed or
missing blobs; reference README.md:76-111, .MISSING_LARGE_BLOBS:1), so parity
tests, the oracle's golden fixtures and ``bench.py`` all draw from this module.
Everything uses ``numpy.random.RandomState`` (bit-stable across numpy versions
and machines) so the GPU box regenerates exactly what the fixtures were made
from; ``tests/golden/manifest.json`` pins checksums.

Shapes follow SURVEY.md section 8: V=10475 vertices, J=55 joints, 486 pose
blendshape rows, 20 shape components (10 betas + 10 expression), 12 hand PCA
components.
"""
from __future__ import annotations

import hashlib
import json
import os
from dataclasses import dataclass, field

import numpy as np

V_SMPLX = 10475
J_SMPLX = 55
NB_SMPLX = 20
N_HAND_PCA = 12

# SMPL-X kinematic tree (public model topology: pelvis, hips/spine, ..., jaw, eyes,
# 15 left-hand and 15 right-hand joints hanging off the wrists 20 / 21).
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

CONTACT_PARTS = ['back', 'butt', 'L_Hand', 'R_Hand', 'L_Leg', 'R_Leg', 'thighs']


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def checksum(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@dataclass
class SMPLXData:
    """Arrays with the keys/shapes of ``SMPLX_NEUTRAL.npz`` that the path uses."""
    v_template: np.ndarray          # [V,3]
    shapedirs: np.ndarray           # [V,3,NB]  (10 betas | 10 expression)
    posedirs: np.ndarray            # [V,3,486] (npz layout; body model reshapes to [486,3V])
    J_regressor: np.ndarray         # [J,V]
    weights: np.ndarray             # [V,J]
    kintree_table: np.ndarray       # [2,J] row 0 = parents
    hands_componentsl: np.ndarray   # [45,45] (first 12 rows used)
    hands_componentsr: np.ndarray
    hands_meanl: np.ndarray         # [45]
    hands_meanr: np.ndarray
    f: np.ndarray                   # [F,3] faces (unused on the hot path)

    def save_npz(self, path: str) -> None:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, **self.__dict__)


def make_smplx(seed: int = 7, V: int = V_SMPLX, J: int = J_SMPLX, parents: np.ndarray | None = None,
               nb: int = NB_SMPLX, weight_nnz: int = 0) -> SMPLXData:
    """SMPL-X-shaped model: peaky (approximately sparse) regressor / skinning weights.  ``weight_nnz`` > 0: every vertex is bound to
    exactly that many joints — a joint and its neighbours in the kinematic tree, consecutive vertices mostly to the same joint —
    as in the released SMPL-X model, whose skinning rows have at most 4 non-zeros (the default is a DENSE random [V, J] matrix,
    which is what the reference's dense matmul is indifferent to and the harder case for this package's kernels)."""
    rs = np.random.RandomState(seed)
    if parents is None:
        parents = SMPLX_PARENTS if J == J_SMPLX else np.array([-1] + [(i - 1) // 2 for i in range(1, J)])
    v_template = _f32(rs.standard_normal((V, 3)) * 0.3)
    shapedirs = _f32(rs.standard_normal((V, 3, nb)) * 0.01)
    posedirs = _f32(rs.standard_normal((V, 3, (J - 1) * 9)) * 0.001)
    jr = rs.uniform(0, 1, (J, V)) ** 8
    J_regressor = _f32(jr / jr.sum(1, keepdims=True))
    w = rs.uniform(0, 1, (V, J)) ** 8
    if weight_nnz > 0:
        par = np.asarray(parents)
        nbrs = [[j] + ([int(par[j])] if par[j] >= 0 else []) + [int(c) for c in np.nonzero(par == j)[0]] for j in range(J)]
        prim = np.minimum((np.arange(V) * J) // V, J - 1)                      # consecutive vertices share their primary joint
        mask = np.zeros((V, J), bool)
        for v in range(V):
            cand = list(nbrs[int(prim[v])])
            while len(cand) < weight_nnz:                                      # leaves: widen to the neighbours' neighbours
                cand += [c for q in list(cand) for c in nbrs[q] if c not in cand] or [int(rs.randint(J))]
            mask[v, cand[:weight_nnz]] = True
        w = np.where(mask, w + 1e-3, 0.0)
    weights = _f32(w / w.sum(1, keepdims=True))
    kintree = np.stack([parents.astype(np.int64), np.arange(J, dtype=np.int64)])
    kintree[0, 0] = -1
    hcl = _f32(rs.standard_normal((45, 45)) * 0.3)
    hcr = _f32(rs.standard_normal((45, 45)) * 0.3)
    hml = _f32(rs.standard_normal(45) * 0.1)
    hmr = _f32(rs.standard_normal(45) * 0.1)
    faces = rs.randint(0, V, (64, 3)).astype(np.int64)
    return SMPLXData(v_template, shapedirs, posedirs, J_regressor, weights, kintree, hcl, hcr, hml, hmr, faces)


def make_vposer_state(seed: int = 3, num_neurons: int = 512, latentD: int = 32, n_joints: int = 21) -> dict:
    """``VPoser(512, 32, [1,21,3])`` state_dict (numpy), keys as vposer_smpl.py:75-89.

    Linear layers use U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like ``nn.Linear``'s default
    reset; BatchNorm buffers are at their defaults.  Only the ``dec`` half is on the path.
    """
    rs = np.random.RandomState(seed)
    nf = n_joints * 3

    def lin(out_f, in_f, gain=1.0):
        k = gain / np.sqrt(in_f)
        return _f32(rs.uniform(-k, k, (out_f, in_f))), _f32(rs.uniform(-k, k, (out_f,)))

    sd = {}
    for name, n in (('bodyprior_enc_bn1', nf), ('bodyprior_enc_bn2', num_neurons)):
        sd[name + '.weight'] = np.ones(n, np.float32)
        sd[name + '.bias'] = np.zeros(n, np.float32)
        sd[name + '.running_mean'] = np.zeros(n, np.float32)
        sd[name + '.running_var'] = np.ones(n, np.float32)
        sd[name + '.num_batches_tracked'] = np.zeros((), np.int64)
    for name, (o, i) in (('bodyprior_enc_fc1', (num_neurons, nf)), ('bodyprior_enc_fc2', (num_neurons, num_neurons)),
                         ('bodyprior_enc_mu', (latentD, num_neurons)), ('bodyprior_enc_logvar', (latentD, num_neurons)),
                         ('bodyprior_dec_fc1', (num_neurons, latentD)), ('bodyprior_dec_fc2', (num_neurons, num_neurons)),
                         ('bodyprior_dec_out', (n_joints * 6, num_neurons))):
        # decoder gain > 1 so that random latents give well-spread joint rotations
        w, b = lin(o, i, gain=2.0 if 'dec' in name else 1.0)
        sd[name + '.weight'], sd[name + '.bias'] = w, b
    return sd


@dataclass
class SceneData:
    verts: np.ndarray        # [m,3] downsampled scene point cloud (scenes_downsampled/*.ply vertices)
    sdf: np.ndarray          # [D,D,D] indexed [ix][iy][iz] (C order, fitting_proxe.py:85)
    grid_min: np.ndarray     # [3]
    grid_max: np.ndarray     # [3]
    grid_dim: int
    contact_parts: dict = field(default_factory=dict)  # part -> {'verts_ind': [...], 'faces_ind': [...]}

    def write_prox_layout(self, root: str, name: str = 'S') -> dict:
        """Write the on-disk formats the entry points read (fitting_proxe.py:80-96, cvae.py:99-115)."""
        os.makedirs(os.path.join(root, 'scenes_sdf'), exist_ok=True)
        os.makedirs(os.path.join(root, 'scenes_downsampled'), exist_ok=True)
        os.makedirs(os.path.join(root, 'body_segments'), exist_ok=True)
        with open(os.path.join(root, 'scenes_sdf', name + '.json'), 'w') as f:
            json.dump({'min': self.grid_min.tolist(), 'max': self.grid_max.tolist(), 'dim': int(self.grid_dim)}, f)
        np.save(os.path.join(root, 'scenes_sdf', name + '_sdf.npy'), self.sdf.reshape(-1))
        write_ply_vertices(os.path.join(root, 'scenes_downsampled', name + '.ply'), self.verts)
        for part, d in self.contact_parts.items():
            with open(os.path.join(root, 'body_segments', part + '.json'), 'w') as f:
                json.dump(d, f)
        return {'scene_verts_path': os.path.join(root, 'scenes_downsampled', name + '.ply'),
                'scene_sdf_path': os.path.join(root, 'scenes_sdf', name),
                'contact_id_folder': os.path.join(root, 'body_segments')}


def make_scene(seed: int = 0, m: int = 32768, D: int = 256, n_contact: int = 2048, V: int = V_SMPLX,
               extent: float = 2.0, radius: float = 0.8, kind: str = 'room') -> SceneData:
    """One scene: m points U(-1.5,1.5)^3, an analytic SDF on [-extent,extent]^3, contact ids in 7 part files.

    kind='room'  : sdf = radius - |p|  (free space inside a spherical room, walls penetrate) - a
                   minority of body vertices is negative, like a body touching PROX furniture;
    kind='sphere': sdf = |p| - radius  (solid ball at the origin; the survey's planning scene)."""
    rs = np.random.RandomState(seed)
    verts = _f32(rs.uniform(-1.5, 1.5, (m, 3)))
    ax = np.linspace(-extent, extent, D, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
    rr = np.sqrt(X * X + Y * Y + Z * Z)
    sdf = (np.float32(radius) - rr if kind == 'room' else rr - np.float32(radius)).astype(np.float32)
    ids = np.sort(rs.choice(V, n_contact, replace=False))
    parts = {}
    for name, p in zip(CONTACT_PARTS, np.array_split(ids, len(CONTACT_PARTS))):
        parts[name] = {'verts_ind': [int(x) for x in p], 'faces_ind': [0]}
    return SceneData(verts, sdf, np.array([-extent] * 3, np.float32), np.array([extent] * 3, np.float32), D, parts)


def contact_ids_from_parts(parts: dict, order=CONTACT_PARTS) -> np.ndarray:
    """Same expression as GeometryTransformer.get_contact_id (cvae.py:99-115): list(set(.)) per part, concatenated."""
    return np.concatenate([list(set(parts[p]['verts_ind'])) for p in order]).astype(np.int64)


def make_bodies(seed: int = 11, B: int = 32) -> dict:
    """Generated-body pkl contents (keys of cvae.py:320-327) in the scale the survey specifies."""
    rs = np.random.RandomState(seed)
    return {
        'transl': _f32(rs.standard_normal((B, 3)) * 0.3),
        'global_orient': _f32(rs.standard_normal((B, 3)) * 0.5),
        'betas': _f32(rs.standard_normal((B, 10))),
        'body_pose': _f32(rs.standard_normal((B, 32))),
        'left_hand_pose': _f32(rs.standard_normal((B, 12)) * 0.3),
        'right_hand_pose': _f32(rs.standard_normal((B, 12)) * 0.3),
        'cam_ext': _f32(np.tile(np.eye(4)[None], (B, 1, 1))),
        'cam_int': _f32(np.tile(np.array([[1000., 0, 960.], [0, 1000., 540.], [0, 0, 1.]])[None], (B, 1, 1))),
    }


def make_cam_ext(seed: int = 5, B: int = 32) -> np.ndarray:
    """Random rigid camera-to-world transforms [B,4,4] (non-identity cam_ext exercises verts_transform)."""
    rs = np.random.RandomState(seed)
    out = np.tile(np.eye(4, dtype=np.float64)[None], (B, 1, 1))
    for b in range(B):
        q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        out[b, :3, :3] = q
        out[b, :3, 3] = rs.standard_normal(3) * 0.2
    return _f32(out)


def body_vector_72(bodies: dict) -> np.ndarray:
    """[transl|global_orient|betas|vposer latent|lh|rh] = 72-D (cvae.py:312-319)."""
    return _f32(np.concatenate([bodies[k] for k in
                                ('transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose')], -1))


from .scene_io import read_ply_vertices, write_ply_vertices  # noqa: E402,F401


def make_state_like(shapes: dict, seed: int = 0) -> dict:
    """Synthetic state_dict for a module given {key: shape}: per-key generators (order independent), nn-style scales.

    Used for the CVAE models, whose pretrained weights (``data/resnet18.pth``, PSI checkpoints) are licensed / missing."""
    import zlib
    out = {}
    for key, shape in shapes.items():
        rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
        shape = tuple(shape)
        if key.endswith('num_batches_tracked'):
            out[key] = np.zeros(shape, np.int64)
        elif key.endswith('running_var'):
            out[key] = _f32(rs.uniform(0.5, 1.5, shape))
        elif key.endswith('running_mean'):
            out[key] = _f32(rs.standard_normal(shape) * 0.1)
        elif len(shape) == 1 and ('bn' in key or '.1.weight' in key or 'downsample.1' in key) and key.endswith('weight'):
            out[key] = _f32(rs.uniform(0.5, 1.5, shape))
        elif len(shape) == 1:
            out[key] = _f32(rs.uniform(-0.05, 0.05, shape))
        else:
            fan_in = int(np.prod(shape[1:]))
            k = 1.0 / np.sqrt(fan_in)
            out[key] = _f32(rs.uniform(-k, k, shape))
    return out


def make_cvae_inputs(seed: int = 13, B: int = 4) -> dict:
    """Scene images (depth | semantics in [-1,1], [B,2,128,128]), body vectors and reparameterisation noise."""
    rs = np.random.RandomState(seed)
    return {'xs': _f32(rs.uniform(-1, 1, (B, 2, 128, 128))),
            'x75': _f32(rs.standard_normal((B, 75)) * 0.5),
            'eps32': _f32(rs.standard_normal((B, 32))),
            'eps32b': _f32(rs.standard_normal((B, 32)))}
