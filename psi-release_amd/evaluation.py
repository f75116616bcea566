"""Physical-plausibility metrics of generated / fitted bodies.

Reference: utils/utils_eval_collision_habitat.py:91-175 — per body: ``non-collision score`` = #(sdf > 0) / 10475 and
``contact score`` = 1 if any vertex has sdf < 0 else 0 (with the all-outside convention: collision 1.0, contact 0).
Same SMPL-X + SDF path as fitting (HIP operators), Habitat camera flip (:160-165).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from . import ops
from .geometry import BodyParamParser, GeometryTransformer


class PlausibilityEvaluator:
    def __init__(self, fitting_op, flip_camera_yz=True):
        """``fitting_op``: a FittingOP (supplies vposer, body model, scene SDF on the GPU)."""
        self.op = fitting_op
        self.flip = flip_camera_yz

    @torch.no_grad()
    def scores(self, body_param_input):
        """(non-collision score, contact score) of one pkl.  The reference scores ONE body per file (batch_size 1,
        utils_eval_collision_habitat.py:145-175); a pkl that holds B > 1 bodies gives two lists with one entry per body."""
        op = self.op
        xh, cam_ext, _ = BodyParamParser.body_params_parse_fitting(body_param_input)
        B = xh.shape[0]
        cam = cam_ext
        if self.flip:
            T_mat = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=op.device)).unsqueeze(0)
            cam = torch.matmul(cam_ext[:1], T_mat).expand(B, -1, -1).contiguous()
        xh_rec = GeometryTransformer.convert_to_3D_rot(GeometryTransformer.convert_to_6D_rot(xh))
        verts = op.body_verts(xh_rec, cam)
        sdf = ops.sdf_sample(verts, op.s_sdf, op.s_grid_min_batch, op.s_grid_max_batch, align_corners=op.align_corners)
        V = verts.shape[1]
        n_neg = (sdf < 0).sum(dim=1).cpu().tolist()
        n_pos = (sdf > 0).sum(dim=1).cpu().tolist()
        coll, cont = [], []
        for neg, pos in zip(n_neg, n_pos):
            if neg < 1:                                     # utils_eval_collision_habitat.py:131-135: nothing penetrates
                coll.append(10475.0 / 10475.0)
                cont.append(0.0)
            else:
                coll.append(float(pos) / 10475.0)           # :137,139 (the reference hard-codes the SMPL-X vertex count)
                cont.append(1.0)
        return (coll[0], cont[0]) if B == 1 else (coll, cont)

    def eval_folder(self, folder, max_files=8000):
        coll, cont = [], []
        for ii in range(max_files):
            fn = os.path.join(folder, 'body_gen_{:06d}.pkl'.format(ii))
            if not os.path.exists(fn):
                continue
            with open(fn, 'rb') as f:
                c, k = self.scores(pickle.load(f))
            coll.extend(c if isinstance(c, list) else [c])
            cont.extend(k if isinstance(k, list) else [k])
        return coll, cont


def diversity_scores(bodies_72: np.ndarray, n_clusters: int = 20, seed: int = 0):
    """utils/utils_eval_diversity.py:93-104: k-means (k=20) on the generated body vectors; returns the entropy of the
    cluster-size histogram and the mean distance of samples to their cluster centre."""
    from sklearn.cluster import KMeans
    x = np.asarray(bodies_72, dtype=np.float64)
    k = min(n_clusters, len(x))
    km = KMeans(n_clusters=k, random_state=seed, n_init=10).fit(x)
    counts = np.bincount(km.labels_, minlength=k).astype(np.float64)
    p = counts / counts.sum()
    entropy = float(-(p[p > 0] * np.log(p[p > 0])).sum())
    mean_dist = float(np.mean(np.linalg.norm(x - km.cluster_centers_[km.labels_], axis=1)))
    return entropy, mean_dist
