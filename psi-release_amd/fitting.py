"""Scene-geometry-aware fitting of generated bodies: ``FittingOP`` with the reference's config keys and methods.

Reference: source/fitting_proxe.py:40-214 (PROX-E) and source/fitting_habitat.py:40-230 (MP3D-R: batch 1, contact
constant 1.0, camera pre-multiplied by diag(1,-1,-1,1)).  Same ``fittingconfig`` / ``lossconfig`` keys, same
``cal_loss`` / ``fitting`` / ``save_result`` methods, same pkl schema and the same verbose line format.

Two execution modes behind the same class:
* ``engine='fused'``  (default on GPU)  — the whole iteration (forward, losses, hand-derived backward, Adam) runs
  as a fixed sequence of HIP kernels inside libpsi_hip.so (csrc/fit.hip), replayed as a hipGraph;
* ``engine='modular'`` — PyTorch autograd over the individual HIP operators (ops.py, body_model.py); this is
  also what the training entry points use, because the CVAE lives in PyTorch.
Both need the GPU; there is no CPU implementation in this package.

Differences from the reference that do not change results: the scene SDF / point cloud are
kept once per scene instead of ``.repeat(batch_size)`` (fitting_proxe.py:90,96); contact ids are read once instead of
every iteration (cvae.py:105-109); both ``contact_id_folder`` and ``body_segments_folder`` are accepted
(fitting_proxe.py:131 vs :238); no ``.item()`` host sync for the penetration mask (fitting_proxe.py:155).
``align_corners`` of the SDF lookup is explicit: True reproduces the pinned torch 1.2.0 behaviour.
"""
from __future__ import annotations

import ctypes
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from . import body_model, dist as psi_dist, hip, ops, scene_io
from .geometry import BodyParamParser, GeometryTransformer
from .vposer import load_vposer


HAS_FUSED_ENGINE = True


def expand_cam_ext(cam_ext, B):
    """cam_ext as [B,4,4]: the reference's PROX-E pkls carry ONE camera [1,4,4] that broadcasts over the batch
    (verts_transform, cvae.py:141-149); anything else must already have B rows."""
    if cam_ext.dim() != 3 or tuple(cam_ext.shape[1:]) != (4, 4) or cam_ext.shape[0] not in (1, B):
        raise ValueError('cam_ext must be [%d,4,4] or [1,4,4], got %s' % (B, tuple(cam_ext.shape)))
    return cam_ext.expand(B, 4, 4).contiguous() if cam_ext.shape[0] != B else cam_ext


class FittingOP:
    contact_const = 0.01            # fitting_proxe.py:139; fitting_habitat.py:141 uses 1.0
    flip_camera_yz = False          # fitting_habitat.py:179-184

    def __init__(self, fittingconfig, lossconfig):
        self.align_corners = True
        self.engine = 'fused'
        self.use_graph = True
        self.dp_use_graph = False       # data-parallel sequence: half-graphs around the all-reduce instead of plain launches
        self.nn_mode = 'kdtree'          # 'kdtree' (exact index over the static scene cloud) | 'bruteforce'
        self.reset_optimizer = False
        self.independent_bodies = False  # True: the batch is B independent problems (per-body loss normalisers): one engine run over B
                                         # bodies == B runs of the reference's loop at batch size 1 (fused engine only; fitting_many packs files)
        self.concurrent_engines = 1      # engines (of this or other FittingOPs) the caller keeps in flight on this GPU at once: sizes the
                                         # per-body head / tail kernels (psi_fit_config.concurrent_engines); fitting_many sets it itself
        self.data_parallel = None        # None: rows are sharded over the ranks whenever torch.distributed runs with > 1 rank;
                                         # False: this instance fits its own independent batch (file-sharded scripts)
        for key, val in fittingconfig.items():
            setattr(self, key, val)
        for key, val in lossconfig.items():
            setattr(self, key, val)
        self.device = torch.device(self.device)
        if self.device.type != 'cuda':
            raise RuntimeError('FittingOP runs on the GPU through libpsi_hip.so; there is no CPU path')
        BodyParamParser.device = self.device

        vposer_src = getattr(self, 'vposer_state', None) or self.vposer_ckpt_path
        self.vposer, _ = load_vposer(vposer_src, vp_model='snapshot')
        smplx_src = getattr(self, 'smplx_data', None) or self.human_model_path
        self.body_mesh_model = body_model.create(smplx_src, model_type='smplx', gender='neutral', ext='npz',
                                                 num_pca_comps=12, create_global_orient=True, create_body_pose=True,
                                                 create_betas=True, create_left_hand_pose=True,
                                                 create_right_hand_pose=True, create_expression=True,
                                                 create_jaw_pose=True, create_leye_pose=True, create_reye_pose=True,
                                                 create_transl=True, batch_size=self.batch_size, device=self.device)
        self.vposer.to(self.device)

        self.xhr_rec = torch.randn(self.batch_size, 75, device=self.device).requires_grad_(True)
        self.optimizer = optim.Adam([self.xhr_rec], lr=self.init_lr_h)

        # scene: one SDF volume and one point cloud (not replicated per batch row)
        scene = getattr(self, 'scene', None)
        if scene is not None:
            sdf, grid_min, grid_max, scene_verts = scene.sdf, scene.grid_min, scene.grid_max, scene.verts
            self._contact_parts = scene.contact_parts
        else:
            sdf, grid_min, grid_max, _ = scene_io.read_sdf(self.scene_sdf_path)
            scene_verts = scene_io.read_ply_vertices(self.scene_verts_path)
            self._contact_parts = None
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=self.device)
        self.s_grid_min_batch = t(grid_min).unsqueeze(0)
        self.s_grid_max_batch = t(grid_max).unsqueeze(0)
        self.s_sdf = t(sdf).unsqueeze(0).contiguous()                  # [1,D,D,D]
        self.s_verts = t(scene_verts).unsqueeze(0).contiguous()        # [1,m,3]
        self._s_verts_expanded = self.s_verts.expand(self.batch_size, -1, -1).contiguous()
        self._vid = None
        self._chamfer = ops.chamferDist(one_sided=True)

    # ---------------------------------------------------------------------------------------
    def dp_world(self):
        """Number of ranks that share this batch's loss normalisers (1 = an independent batch)."""
        if self.data_parallel is False:
            return 1
        return psi_dist.world_size()

    def contact_vertex_ids(self):
        if self._vid is None:
            if self._contact_parts is not None:
                vid = np.concatenate([list(set(self._contact_parts[p]['verts_ind'])) for p in self.contact_part])
            else:
                folder = getattr(self, 'body_segments_folder', None) or getattr(self, 'contact_id_folder')
                vid, _ = GeometryTransformer.get_contact_id(body_segments_folder=folder, contact_body_parts=self.contact_part)
            self._vid = torch.tensor(np.asarray(vid).astype(np.int64), device=self.device)
        return self._vid

    def body_verts(self, xh_rec, cam_ext):
        body_param_rec = BodyParamParser.body_params_encapsulate_batch(xh_rec)
        joint_rot_batch = self.vposer.decode(body_param_rec['body_pose_vp'], output_type='aa').view(xh_rec.shape[0], -1)
        body_param_ = {k: v for k, v in body_param_rec.items() if k != 'body_pose_vp'}
        # cam_ext is applied inside the skinning kernel (== GeometryTransformer.verts_transform, cvae.py:141-149)
        out = self.body_mesh_model(return_verts=True, body_pose=joint_rot_batch, cam_ext=cam_ext, **body_param_)
        return out.vertices

    def cal_loss(self, xhr, cam_ext):
        """fitting_proxe.py:101-162."""
        loss_rec = self.weight_loss_rec * F.l1_loss(xhr, self.xhr_rec)
        xh_rec = GeometryTransformer.convert_to_3D_rot(self.xhr_rec)
        loss_vposer = self.weight_loss_vposer * torch.mean(xh_rec[:, 16:48] ** 2)

        body_verts_batch = self.body_verts(xh_rec, cam_ext)
        body_verts_contact_batch = body_verts_batch[:, self.contact_vertex_ids(), :]
        if self.nn_mode == 'kdtree':
            if getattr(self, '_nn_index', None) is None:
                self._nn_index = ops.SceneNNIndex(self.s_verts[0], self.device)
            contact_dist = ops.chamfer_to_scene(body_verts_contact_batch.contiguous(), self._nn_index)
        else:
            contact_dist, _ = self._chamfer(body_verts_contact_batch.contiguous(), self._s_verts_expanded)
        s = torch.sqrt(contact_dist + 1e-4)
        loss_contact = self.weight_contact * torch.mean(s / (s + self.contact_const))

        body_sdf_batch = ops.sdf_sample(body_verts_batch, self.s_sdf, self.s_grid_min_batch, self.s_grid_max_batch,
                                        scene_id=None, align_corners=self.align_corners)
        if self.dp_world() > 1:
            # data-parallel batch: global-batch normalisers through ONE all-reduce (dist.py)
            loss_rec, loss_vposer, loss_contact, pen = psi_dist.fitting_loss_reduce(loss_rec, loss_vposer, loss_contact,
                                                                                  body_sdf_batch)
            loss_collision = self.weight_collision * pen
        else:
            loss_collision = self.weight_collision * ops.penetration_loss(body_sdf_batch)
        return loss_rec, loss_vposer, loss_contact, loss_collision

    def _camera(self, cam_ext):
        if not self.flip_camera_yz:
            return cam_ext
        T_mat = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=self.device)).unsqueeze(0)
        return torch.matmul(cam_ext[:1], T_mat).expand(self.batch_size, -1, -1).contiguous()

    def make_step_runner(self, input_data_file):
        """Parse one generated-body record and return an object whose ``step()`` runs ONE fitting iteration
        (fitting_proxe.py:179-189) with every input already resident in HBM; ``fitting`` loops over it."""
        if isinstance(input_data_file, dict):
            body_param_input = input_data_file
        else:
            with open(input_data_file, 'rb') as f:
                body_param_input = pickle.load(f)
        xh, self.cam_ext, self.cam_int = BodyParamParser.body_params_parse_fitting(body_param_input)
        xhr = GeometryTransformer.convert_to_6D_rot(xh)
        self.xhr_rec.data = xhr.clone()
        if self.reset_optimizer:
            self.optimizer = optim.Adam([self.xhr_rec], lr=self.init_lr_h)
        if self.engine == 'fused':
            return _FusedRunner(self, xhr, self._camera(self.cam_ext))
        return _ModularRunner(self, xhr, self._camera(self.cam_ext))

    def fitting(self, input_data_file):
        """fitting_proxe.py:167-195; ``input_data_file`` is a pkl path or the already-loaded dict."""
        runner = self.make_step_runner(input_data_file)
        if not self.verbose:
            runner.steps(self.num_iter)                 # fused engine: the loop is device-resident, 10 iterations per graph launch
        for ii in range(self.num_iter if self.verbose else 0):
            runner.step()
            l = runner.last_losses()
            print('[INFO][fitting] iter={:d}, l_rec={:f}, l_vposer={:f}, l_contact={:f}, l_collision={:f}'.format(
                ii, l[0], l[1], l[2], l[3]))
        runner.finish()
        print('[INFO][fitting] fitting finish, returning optimal value')
        return GeometryTransformer.convert_to_3D_rot(self.xhr_rec)

    def fitting_many(self, inputs, concurrency=4):
        """Fit a list of INDEPENDENT generated-body records (pkl paths or dicts; what the entry points' file loop feeds one at a time,
        fitting_proxe.py:252-263) with up to ``concurrency`` engine runs in flight: every run has its own fused engine and HIP stream, so
        the latency-bound kernels of different runs overlap on the GPU (the reference's batch size per file is 1: a single file leaves
        most of the chip idle).  With ``independent_bodies=True`` the engine normalises every loss per body, and the records are PACKED:
        ``batch_size // bodies_per_record`` files form one engine run that equals fitting them one by one — 32 files per 0.16 ms
        iteration instead of one.  Per-file semantics are unchanged except that every run starts from a fresh Adam state (the reference
        carries one optimizer across the files of a scene).  Returns the fitted 72-D body vectors per file, in input order, and the
        per-file cameras."""
        if self.engine != 'fused':
            raise ValueError('fitting_many needs engine="fused"')
        if self.dp_world() > 1:
            raise ValueError('fitting_many fits independent problems: build the FittingOP with data_parallel=False')
        if not inputs:
            return [], []
        # host-side glue ONCE for the whole list (per file it is ~100 small torch launches = 1.8 ms of host time, more than the
        # 20 iterations themselves take on the GPU): parse and stack the records, one batched 3D->6D conversion in, one 6D->3D out
        recs = []
        for inp in inputs:
            if isinstance(inp, dict):
                recs.append(inp)
            else:
                with open(inp, 'rb') as f:
                    recs.append(pickle.load(f))
        keys = ('transl', 'global_orient', 'betas', 'body_pose', 'left_hand_pose', 'right_hand_pose')
        nb_rec = int(np.asarray(recs[0]['transl']).shape[0])                  # bodies per record
        if any(int(np.asarray(r['transl']).shape[0]) != nb_rec for r in recs):
            raise ValueError('fitting_many: all records must hold the same number of bodies')
        if self.independent_bodies:
            if self.batch_size % nb_rec:
                raise ValueError('independent_bodies: batch_size (%d) must be a multiple of the bodies per record (%d)' % (self.batch_size, nb_rec))
            pack = self.batch_size // nb_rec                                   # records per engine run
        elif nb_rec != self.batch_size:
            raise ValueError('FittingOP was built for batch_size=%d: every record must hold %d bodies' % (self.batch_size, self.batch_size))
        else:
            pack = 1
        n_real = len(recs)
        while len(recs) % pack:                                                # the last run is padded with copies of the last record
            recs.append(recs[-1])
        B = nb_rec
        xh_all = torch.tensor(np.concatenate([np.concatenate([np.asarray(r[k], dtype=np.float32).reshape(B, -1) for k in keys], axis=1) for r in recs]),
                              dtype=torch.float32, device=self.device)
        def cam_rows(r):
            c = np.asarray(r['cam_ext'], dtype=np.float32).reshape(-1, 4, 4)
            if c.shape[0] not in (1, B):
                raise ValueError('cam_ext must hold 1 or %d cameras, got %d' % (B, c.shape[0]))
            return np.broadcast_to(c, (B, 4, 4)) if c.shape[0] != B else c
        cam_all = torch.tensor(np.stack([cam_rows(r) for r in recs]), dtype=torch.float32, device=self.device)     # [N,B,4,4]
        if self.flip_camera_yz:                                                                      # fitting_habitat.py:179-184
            T_mat = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0], device=self.device))
            cam_run = torch.matmul(cam_all[:, :1], T_mat).expand(-1, B, -1, -1).contiguous()
        else:
            cam_run = cam_all
        xhr_all = GeometryTransformer.convert_to_6D_rot(xh_all).contiguous()                         # [N*B,75]
        x_out = torch.empty_like(xhr_all)
        cam_run = cam_run.reshape(-1, 4, 4).contiguous()                                              # [N*B,4,4]
        n_runs = len(recs) // pack
        R = pack * B                                                                                  # rows per engine run (== batch_size)
        K = max(1, min(int(concurrency), n_runs))
        # the engines are built for the number of them that share the GPU (psi_fit_config.concurrent_engines)
        pool = getattr(self, '_fused_pool', [])
        if not pool or pool[0].concurrent_engines != K:
            self.concurrent_engines = K
            pool = []
            self._fused = None
        while len(pool) < K:
            pool.append(FusedEngine(self))
        self._fused_pool = pool
        if self._fused is None:
            self._fused = pool[0]
        engines = pool[:K]
        torch.cuda.current_stream().synchronize()                                                    # inputs are ready for every engine stream
        L = hip.lib()
        for i in range(n_runs):
            eng = engines[i % K]
            xs = xhr_all[i * R:(i + 1) * R]
            hip.check(L.psi_fit_set_problem(eng.handle, hip.ptr(xs), hip.ptr(xs), hip.ptr(cam_run[i * R:(i + 1) * R]), 1, eng.stream.cuda_stream),
                      'psi_fit_set_problem')
            eng.iterate(self.num_iter, self.use_graph)
            hip.check(L.psi_fit_read(eng.handle, hip.ptr(x_out[i * R:(i + 1) * R]), None, 0, None, eng.stream.cuda_stream), 'psi_fit_read')
        import ctypes
        for eng in engines:
            # synchronises the engine's stream AND inspects its error word (sticky until read): a head / tail cluster exchange that gave
            # up waiting in ANY of this engine's runs makes psi_fit_read fail here instead of a silently invalid fit reaching the pkl
            step = ctypes.c_int(0)
            hip.check(L.psi_fit_read(eng.handle, None, None, 0, ctypes.byref(step), eng.stream.cuda_stream), 'psi_fit_read (fitting_many)')
        xh_fit = GeometryTransformer.convert_to_3D_rot(x_out)
        recs = recs[:n_real]
        results = [xh_fit[i * B:(i + 1) * B] for i in range(n_real)]
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)       # host tensors: save_result only writes them back out
        cams = [(t(r['cam_ext']), t(r['cam_int'])) for r in recs]
        return results, cams

    def save_result(self, xh_rec, output_data_file):
        """fitting_proxe.py:199-214 (one pkl per call; with batch>1 the last body wins, as in the reference)."""
        dirname = os.path.dirname(output_data_file)
        if dirname:
            os.makedirs(dirname, exist_ok=True)      # several ranks of a file-sharded run create the same scene directory at once
        body_param_list = BodyParamParser.body_params_encapsulate(xh_rec)
        print('[INFO] save results to: ' + output_data_file)
        if getattr(self, 'save_all_rows', False) and len(body_param_list) > 1:
            # batched fits: one pkl with ALL rows, in the schema of a generated-body pkl ([B,.] arrays); the reference loop below
            # rewrites the same file once per row, so with batch_size > 1 only the last body survives
            out = {k: np.concatenate([bp[k] for bp in body_param_list], axis=0) for k in body_param_list[0]}
            out['cam_ext'] = self.cam_ext.detach().cpu().numpy()
            out['cam_int'] = self.cam_int.detach().cpu().numpy()
            with open(output_data_file, 'wb') as outfile:
                pickle.dump(out, outfile)
            return
        for body_param in body_param_list:
            body_param['cam_ext'] = self.cam_ext.detach().cpu().numpy()
            body_param['cam_int'] = self.cam_int.detach().cpu().numpy()
            with open(output_data_file, 'wb') as outfile:
                pickle.dump(body_param, outfile)


class _ModularRunner:
    """One fitting iteration = zero_grad, cal_loss, backward, Adam step (autograd over the HIP operators)."""

    def __init__(self, op, xhr, cam):
        if tuple(xhr.shape) != (op.batch_size, 75):
            raise ValueError('FittingOP was built for batch_size=%d: expected body vectors of shape (%d, 75), got %s'
                             % (op.batch_size, op.batch_size, tuple(xhr.shape)))
        self.op, self.xhr, self.cam = op, xhr, expand_cam_ext(cam, op.batch_size)
        self._losses = None

    def step(self):
        op = self.op
        op.optimizer.zero_grad()
        losses = op.cal_loss(self.xhr, self.cam)
        self._losses = [l.detach() for l in losses]
        (losses[0] + losses[1] + losses[2] + losses[3]).backward()
        op.optimizer.step()

    def steps(self, n):
        for _ in range(n):
            self.step()

    def last_losses(self):
        """Loss values evaluated at the START of the last step (what the reference prints), as Python floats."""
        return [float(l) for l in self._losses]

    def finish(self):
        pass


class FusedEngine:
    """Owns a ``psi_fit_engine`` (include/psi_hip.h): the whole iteration as HIP kernels, replayed as a hipGraph."""

    def __init__(self, op):
        self.op = op
        dev = op.device
        sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in op.vposer.state_dict().items() if 'dec' in k}
        bm = op.body_mesh_model
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        w1, b1 = f32(sd['bodyprior_dec_fc1.weight']), f32(sd['bodyprior_dec_fc1.bias'])
        w2, b2 = f32(sd['bodyprior_dec_fc2.weight']), f32(sd['bodyprior_dec_fc2.bias'])
        w3, b3 = f32(sd['bodyprior_dec_out.weight']), f32(sd['bodyprior_dec_out.bias'])
        if w1.shape != (512, 32) or w2.shape != (512, 512) or w3.shape != (126, 512):
            raise hip.PsiHipError('the fused engine expects VPoser(512, 32, [1,21,3])')
        lhc, rhc = f32(bm.left_hand_components.cpu().numpy()), f32(bm.right_hand_components.cpu().numpy())
        pm = f32(bm.pose_mean.cpu().numpy())
        vid = np.ascontiguousarray(op.contact_vertex_ids().cpu().numpy(), dtype=np.int32)
        gmin, gmax = f32(op.s_grid_min_batch.cpu().numpy().reshape(3)), f32(op.s_grid_max_batch.cpu().numpy().reshape(3))
        world = op.dp_world() if hasattr(op, 'dp_world') else 1
        self.world = world
        if world > 1:
            psi_dist.assert_equal_across_ranks(op.batch_size, 'per-rank batch size')   # the normalisers are B * world
        cfg = hip.FitConfig(B=op.batch_size, n_contact=len(vid), m_scene=op.s_verts.shape[1], D=op.s_sdf.shape[1],
                            align_corners=int(bool(op.align_corners)), world_size=world, num_pca_comps=lhc.shape[0],
                            max_history=4096, nn_mode=1 if op.nn_mode == 'kdtree' else 0, w_rec=op.weight_loss_rec, w_vposer=op.weight_loss_vposer,
                            w_contact=op.weight_contact, w_collision=op.weight_collision, contact_const=op.contact_const,
                            lr=op.init_lr_h, beta1=0.9, beta2=0.999, eps=1e-8, lr_d=float(op.init_lr_h), beta1_d=0.9, beta2_d=0.999,
                            independent_bodies=int(bool(getattr(op, 'independent_bodies', False))),
                            concurrent_engines=int(getattr(op, 'concurrent_engines', 1)))
        self.concurrent_engines = int(getattr(op, 'concurrent_engines', 1))
        self._keep = (op.s_verts, op.s_sdf)                    # device arrays the engine points into
        h = ctypes.c_void_p()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        with torch.cuda.device(dev):
            hip.check(hip.lib().psi_fit_create(ctypes.byref(h), bm.lbs_model.handle, ctypes.byref(cfg), p(w1), p(b1), p(w2), p(b2),
                                               p(w3), p(b3), p(lhc), p(rhc), p(pm), p(vid), hip.ptr(op.s_verts), hip.ptr(op.s_sdf),
                                               p(gmin), p(gmax)), 'psi_fit_create')
        self.handle = h
        self.stream = torch.cuda.Stream(device=dev)
        self.stats = torch.zeros(8, device=dev)
        self.max_history = 4096

    def set_problem(self, xhr, x_init, cam, reset):
        B = self.op.batch_size
        if tuple(xhr.shape) != (B, 75) or tuple(x_init.shape) != (B, 75):
            raise ValueError('FusedEngine was built for batch_size=%d: expected body vectors of shape (%d, 75), got %s / %s (a generated-body '
                             'pkl must hold batch_size rows)' % (B, B, tuple(xhr.shape), tuple(x_init.shape)))
        cam = expand_cam_ext(cam, B)
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        self._args = (xhr.contiguous(), x_init.contiguous(), cam.contiguous())
        hip.check(hip.lib().psi_fit_set_problem(self.handle, hip.ptr(self._args[0]), hip.ptr(self._args[1]), hip.ptr(self._args[2]),
                                                int(bool(reset)), self.stream.cuda_stream), 'psi_fit_set_problem')

    def iterate(self, n, use_graph=True):
        L = hip.lib()
        # PSI_FORCE_DP_PATH=1 runs the data-parallel sequence (forward half, all-reduce, backward half) even at world size 1:
        # the way to exercise the RCCL leg on a single-GPU box
        if self.world == 1 and not (os.environ.get('PSI_FORCE_DP_PATH') == '1' and torch.distributed.is_initialized()):
            hip.check(L.psi_fit_iterate(self.handle, n, int(bool(use_graph)), self.stream.cuda_stream), 'psi_fit_iterate')
            return
        import torch.distributed as tdist
        if tdist.get_backend() == 'nccl' and os.environ.get('PSI_DP_PYTHON_LOOP') != '1':
            # RCCL: the whole loop is device-resident — psi_fit_iterate_dp issues the all-reduce from C on the engine's stream, and with
            # use_graph the kernels AND the collective of 10 iterations replay as one hipGraph (no Python between iterations)
            hip.check(L.psi_fit_iterate_dp(self.handle, psi_dist.rccl_comm(), n, int(bool(use_graph)), hip.ptr(self.stats), self.stream.cuda_stream),
                      'psi_fit_iterate_dp')
            return
        # gloo (several ranks on one GPU: the single-GPU test boxes) has no device-side collective: forward half, torch.distributed
        # all-reduce, backward half.  Plain launches by default: two half-graphs per iteration pay the ~8 us graph-launch latency twice;
        # op.dp_use_graph / PSI_DP_GRAPH=1 select the half-graphs.
        use_graph = bool(use_graph) and (getattr(self.op, 'dp_use_graph', False) or os.environ.get('PSI_DP_GRAPH') == '1')
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                hip.check(L.psi_fit_forward(self.handle, hip.ptr(self.stats), int(bool(use_graph)), self.stream.cuda_stream), 'psi_fit_forward')
                tdist.all_reduce(self.stats, op=tdist.ReduceOp.SUM)           # the one collective of the data path
                hip.check(L.psi_fit_backward_step(self.handle, hip.ptr(self.stats), int(bool(use_graph)), self.stream.cuda_stream), 'psi_fit_backward_step')

    def dp_mode(self):
        """How the data-parallel iterations of this engine were launched: 0 none yet / python loop, 1 hipGraphs with the collective
        inside, 2 eager launches from C (psi_fit_dp_mode)."""
        return int(hip.lib().psi_fit_dp_mode(self.handle))

    def watchdog(self):
        """Data-parallel runs: wait for the engine's stream with a bound (PSI_DP_WATCHDOG_S seconds, default 60) before a blocking read —
        a collective that can never complete (a rank that died, ranks that issued different numbers of collectives) becomes an error
        with a message instead of a process that hangs in hipStreamSynchronize for ever."""
        if self.world > 1 or os.environ.get('PSI_FORCE_DP_PATH') == '1':
            limit = float(os.environ.get('PSI_DP_WATCHDOG_S', '60'))
            hip.check(hip.lib().psi_stream_wait(self.stream.cuda_stream, int(limit * 1000)), 'psi_stream_wait (data-parallel watchdog)')

    def read(self, n_hist=0):
        op = self.op
        self.watchdog()
        x = torch.empty(op.batch_size, 75, device=op.device)
        hist = torch.empty(max(n_hist, 1), 4, device=op.device)
        step = ctypes.c_int(0)
        hip.check(hip.lib().psi_fit_read(self.handle, hip.ptr(x), hip.ptr(hist), n_hist, ctypes.byref(step), self.stream.cuda_stream),
                  'psi_fit_read')
        torch.cuda.current_stream().wait_stream(self.stream)
        return x, hist[:n_hist], step.value

    def read_losses(self, adam_step):
        """The four loss values of Adam step ``adam_step`` (one 16-byte copy, not the whole history ring)."""
        out = torch.empty(4, device=self.op.device)
        self.watchdog()
        hip.check(hip.lib().psi_fit_read_losses(self.handle, int(adam_step), hip.ptr(out), self.stream.cuda_stream), 'psi_fit_read_losses')
        self.stream.synchronize()
        return out

    def profile(self, n_rep=20):
        """{kernel name: average ms} of one iteration, HIP events on the launch stream (psi_fit_profile)."""
        names = ctypes.create_string_buffer(48 * 48)
        ms = (ctypes.c_float * 48)()
        n = ctypes.c_int(0)
        hip.check(hip.lib().psi_fit_profile(self.handle, n_rep, names, 48, ms, 48, ctypes.byref(n), self.stream.cuda_stream),
                  'psi_fit_profile')
        out = []
        for i in range(n.value):
            out.append((names.raw[i * 48:(i + 1) * 48].split(b'\0')[0].decode(), float(ms[i])))
        return out

    def buffer(self, name, shape):
        """Copy of an engine-owned device buffer (tests / diagnostics)."""
        out = torch.empty(*shape, device=self.op.device)
        hip.check(hip.lib().psi_fit_copy_buffer(self.handle, name.encode(), hip.ptr(out), out.numel(), self.stream.cuda_stream),
                  'psi_fit_copy_buffer')
        self.stream.synchronize()
        return out

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                torch.cuda.synchronize()
                hip.lib().psi_fit_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _FusedRunner:
    """step() = one hipGraph replay (single GPU) or forward / all-reduce / backward (data parallel)."""

    def __init__(self, op, xhr, cam):
        self.op = op
        if getattr(op, '_fused', None) is None:
            op._fused = FusedEngine(op)
            first = True
        else:
            first = False
        self.eng = op._fused
        self.eng.set_problem(xhr, op.xhr_rec.data, cam, reset=op.reset_optimizer and not first)
        _, _, self.step0 = self.eng.read(0)
        self.n = 0

    def step(self):
        self.eng.iterate(1, self.op.use_graph)
        self.n += 1

    def steps(self, n):
        self.eng.iterate(n, self.op.use_graph)
        self.n += n

    def restart(self):
        """Back to the start of the loop (fitting_proxe.py:167-175 with a fresh optimiser): the generated bodies as parameters, zeroed Adam
        state and step count, no nearest-neighbour warm-start hints.  Enqueued on the engine's stream, no host synchronisation."""
        xhr, _, cam = self.eng._args
        self.eng.set_problem(xhr, xhr, cam, reset=True)
        self.step0, self.n = 0, 0

    def last_losses(self):
        idx = self.step0 + self.n                       # Adam step count after the last step
        if idx < 1:
            return [float('nan')] * 4
        return [float(v) for v in self.eng.read_losses(idx).cpu()]

    def finish(self):
        x, _, _ = self.eng.read(0)
        self.op.xhr_rec.data = x


class FittingOPHabitat(FittingOP):
    """fitting_habitat.py: contact constant 1.0 (:141), camera flipped to the Habitat convention (:179-184)."""
    contact_const = 1.0
    flip_camera_yz = True


# ------------------------------------------------------------------------------------------------------------------
# Differentiable body decode for the CVAE training losses
# ------------------------------------------------------------------------------------------------------------------
class BodyDecoder:
    """x75 [B,75] = [transl | 6D global rot | betas | VPoser latent | hand PCA] -> camera-frame SMPL-X vertices [B,V,3] with
    a hand-derived backward (psi_fit_decode_forward / _backward): ONE op instead of the reference's
    ``convert_to_3D_rot -> body_params_encapsulate_batch -> vposer.decode -> body_mesh_model -> verts_transform`` chain
    (train_s1.py:136-157) and the ~1000 elementwise launches autograd spends on it per step.  It is a fused engine used
    only for its head / LBS kernels; the engine's scene-side inputs are one-voxel placeholders that are never evaluated."""

    def __init__(self, vposer, body_mesh_model, batch_size, device):
        import types
        dev = torch.device(device)
        shim = types.SimpleNamespace(
            device=dev, vposer=vposer, body_mesh_model=body_mesh_model, batch_size=batch_size, align_corners=True, nn_mode='bruteforce',
            contact_vertex_ids=lambda: torch.zeros(1, dtype=torch.int64), s_verts=torch.zeros(1, 8, 3, device=dev),
            s_sdf=torch.ones(1, 2, 2, 2, device=dev), s_grid_min_batch=torch.full((1, 3), -1.0), s_grid_max_batch=torch.full((1, 3), 1.0),
            weight_loss_rec=0.0, weight_loss_vposer=0.0, weight_contact=0.0, weight_collision=0.0, contact_const=1.0, init_lr_h=0.0,
            dp_world=lambda: 1)
        self.engine = FusedEngine(shim)
        self.batch_size = batch_size
        self.V = int(body_mesh_model.lbs_model.V) if hasattr(body_mesh_model.lbs_model, 'V') else 10475
        self.version = 0

    def __call__(self, x75, cam_ext):
        return _BodyDecodeFn.apply(x75, cam_ext, self)


class _BodyDecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x75, cam_ext, dec):
        if x75.shape != (dec.batch_size, 75):
            raise ValueError('BodyDecoder was built for x75 of shape (%d, 75), got %s' % (dec.batch_size, tuple(x75.shape)))
        x = x75.detach().contiguous().float()
        cam = expand_cam_ext(cam_ext.detach(), dec.batch_size).contiguous().float()
        verts = torch.empty(dec.batch_size, dec.V, 3, device=x.device)
        hip.check(hip.lib().psi_fit_decode_forward(dec.engine.handle, hip.ptr(x), hip.ptr(cam), hip.ptr(verts), hip.stream()),
                  'psi_fit_decode_forward')
        dec.version += 1
        ctx.dec, ctx.version = dec, dec.version
        return verts

    @staticmethod
    def backward(ctx, gverts):
        dec = ctx.dec
        if dec.version != ctx.version:
            raise RuntimeError('BodyDecoder: another forward ran before this backward (the activations live in the engine)')
        g = gverts.contiguous().float()
        gx = torch.empty(dec.batch_size, 75, device=g.device)
        hip.check(hip.lib().psi_fit_decode_backward(dec.engine.handle, hip.ptr(g), hip.ptr(gx), hip.stream()), 'psi_fit_decode_backward')
        return gx, None, None
