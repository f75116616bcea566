"""CVAE training with the scene losses: ``TrainOP`` for stage-1 and stage-2 models, reference config keys and flags.

Reference: source/train_s1.py:37-338 (HumanCVAES1, one KL term) and source/train_s2.py:37-340 (HumanCVAES2, two KL terms).
Same ``trainconfig`` / ``lossconfig`` keys, same ``cal_loss`` signatures and return order, same checkpoint schema
``{'epoch', 'model_h_state_dict', 'optimizer_h_state_dict'}`` in ``{save_dir}/epoch-{ep+1:06d}.ckp`` every 10 epochs and
every 2 h of wall clock, resume from the newest ``epoch-*.ckp`` by mtime (train_s1.py:220-233,303-321), same verbose
lines.  The body / Chamfer / SDF sub-stack is the HIP operator set (ops.py, body_model.py); the CVAE (models.py) runs on the hand-written
convolution / BatchNorm / dense kernels in both precisions (fp32 = the reference's; optional bf16 autocast), the optimiser on optim.Adam.  GPU only.

Data parallel (new; the reference is single-GPU): wrap-free — gradients of the model parameters are all-reduced
(averaged) across ranks after ``backward`` in one flattened bucket per dtype.  Mean-type losses over equal row shards average to the
full-batch mean; the one data-dependent normaliser, the count of penetrating vertices (train_s1.py:192-202), is made global by a
2-float all-reduce inside the loss (dist.penetration_loss_global), so sharded training equals the full batch in every loss phase.
"""
from __future__ import annotations

import glob
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from . import body_model, dist as psi_dist, ops, optim as psi_optim
from .geometry import BodyParamParser, GeometryTransformer
from .models import HumanCVAES1, HumanCVAES2
from .vposer import load_vposer


class _TrainBase:
    stage = None

    def __init__(self, trainconfig, lossconfig):
        self.align_corners = True
        self.autocast_bf16 = False
        self.resume_training = False
        self.loss_weight_anealing = True
        self.use_cont_rot = True
        self.verbose = False
        self.scene_model_ckpt = None
        self.fused_decode = True        # body decode (6D rot -> VPoser -> SMPL-X -> camera frame) as ONE HIP op (fitting.BodyDecoder)
        self.use_graph = False          # capture the whole optimiser step (forward, backward, Adam) in one HIP graph
        self.fused_glue = True          # the [B,75] body-vector glue and the two scene terms of cal_loss as fused HIP ops (ops.cvae_* / scene_losses)
        self.grad_bucket_mb = 16.0      # data-parallel runs: size of the gradient buckets (dist.GradBuckets)
        self._graphs = {}
        self._fca_t = None
        for key, val in trainconfig.items():
            setattr(self, key, val)
        for key, val in lossconfig.items():
            setattr(self, key, val)
        self.device = torch.device(self.device)
        if self.device.type != 'cuda':
            raise RuntimeError('TrainOP runs on the GPU (HIP operators); there is no CPU path')
        os.makedirs(self.save_dir, exist_ok=True)        # every rank of a torchrun job constructs TrainOP on the same save_dir
        n_dim_body = 72 + 3 if self.use_cont_rot else 72
        self.model_h_latentD = 256
        self.model_h = self._make_model(n_dim_body)
        # train_s1.py:229 optim.Adam defaults; fused=True is the same update as ONE multi-tensor kernel instead of ~10 per step
        # the same optimiser as ONE hand-written multi-tensor launch on the GPU (optim.py / csrc/adam.hip; state layout of torch.optim.Adam)
        self.optimizer_h = psi_optim.Adam(self.model_h.parameters(), lr=self.init_lr_h, fused=bool(getattr(self, 'fused_adam', True)))
        vposer_src = getattr(self, 'vposer_state', None) or self.vposer_ckpt_path
        self.vposer, _ = load_vposer(vposer_src, vp_model='snapshot')
        self.vposer.to(self.device)
        smplx_src = getattr(self, 'smplx_data', None) or self.human_model_path
        self.body_mesh_model = body_model.create(smplx_src, model_type='smplx', gender='neutral', ext='npz', num_pca_comps=12,
                                                 batch_size=self.batch_size, device=self.device)
        self._vid = None
        self._decoder = None
        self._contact_parts = getattr(self, 'contact_parts_data', None)
        self._chamfer = ops.chamferDist(one_sided=True)
        print('--[INFO] device: ' + str(torch.cuda.get_device_name(self.device)))

    # -------------------------------------------------------------------------------------
    def _contact_ids(self):
        if self._vid is None:
            if self._contact_parts is not None:
                vid = np.concatenate([list(set(self._contact_parts[p]['verts_ind'])) for p in self.contact_part])
            else:
                vid, _ = GeometryTransformer.get_contact_id(body_segments_folder=self.contact_id_folder,
                                                            contact_body_parts=self.contact_part)
            self._vid = torch.tensor(np.asarray(vid).astype(np.int64), device=self.device)
            self._vid32 = self._vid.to(torch.int32)
        return self._vid

    def _scene_losses(self, xh_rec, cam_ext, scene_verts, s_grid_min_batch, s_grid_max_batch, s_grid_sdf_batch, ep, loss_vposer=None):
        """Shared tail of cal_loss: VPoser prior, contact (const 1.0) and penetration terms (train_s1.py:136-205)."""
        fused = xh_rec.shape[1] == 75                                            # 75-D: 6D global rotation, decoded by the fused op
        if loss_vposer is None:                                                  # (the fused glue op has it already)
            latent = xh_rec[:, 19:51] if fused else xh_rec[:, 16:48]
            loss_vposer = self.weight_loss_vposer * torch.mean(latent ** 2)
        if not ep > 0.75 * self.epoch and getattr(self, 'skip_gated_losses', True):
            # train_s1.py:171-173,197-199 multiply both scene terms by 0 for the first 75% of the epochs; their value and
            # gradient are exactly 0 there, so the body mesh, NN search and SDF lookup are not evaluated at all.
            zero = xh_rec.new_zeros(())
            return zero, loss_vposer, zero
        if fused:
            if self._decoder is None or self._decoder.batch_size != xh_rec.shape[0]:
                from .fitting import BodyDecoder
                self._decoder = BodyDecoder(self.vposer, self.body_mesh_model, xh_rec.shape[0], self.device)
            body_verts_batch = self._decoder(xh_rec, cam_ext)
        else:
            body_param_rec = BodyParamParser.body_params_encapsulate_batch(xh_rec)
            joint_rot_batch = self.vposer.decode(body_param_rec['body_pose_vp'], output_type='aa').view(xh_rec.shape[0], -1)
            body_param_ = {k: v for k, v in body_param_rec.items() if k != 'body_pose_vp'}
            body_verts_batch = self.body_mesh_model(return_verts=True, body_pose=joint_rot_batch, cam_ext=cam_ext, **body_param_).vertices
        gate = 1.0 if ep > 0.75 * self.epoch else 0.0                           # train_s1.py:171-173,197-199
        if (isinstance(s_grid_sdf_batch, tuple) and len(s_grid_sdf_batch) == 5 and getattr(self, 'use_scene_index', True)
                and not psi_dist.is_dist() and self.fused_glue):
            # both scene terms and their vertex gradient as one op (ops.scene_losses): 6 launches forward, 2 backward
            sdf_t, sid, gmin_t, gmax_t, scenes = s_grid_sdf_batch
            loss_contact, loss_sdf_pene = ops.scene_losses(body_verts_batch, self._contact_ids(), scenes, sid, sdf_t, gmin_t, gmax_t,
                                                           self.align_corners, self.weight_contact, self.weight_collision, gate, vid32=self._vid32)
            return loss_contact, loss_vposer, loss_sdf_pene
        body_verts_contact_batch = body_verts_batch[:, self._contact_ids(), :]
        if isinstance(s_grid_sdf_batch, tuple) and len(s_grid_sdf_batch) == 5 and getattr(self, 'use_scene_index', True):
            contact_dist = ops.chamfer_to_scenes(body_verts_contact_batch.contiguous(), s_grid_sdf_batch[4], s_grid_sdf_batch[1])
        else:
            contact_dist, _ = self._chamfer(body_verts_contact_batch.contiguous(), scene_verts.contiguous())
        s = torch.sqrt(contact_dist + 1e-4)
        loss_contact = gate * self.weight_contact * torch.mean(s / (s + 1.0))
        if isinstance(s_grid_sdf_batch, tuple):                                 # (sdf_table, scene_id, gmin_table, gmax_table)
            sdf_t, sid, gmin_t, gmax_t = s_grid_sdf_batch[:4]
            body_sdf = ops.sdf_sample(body_verts_batch, sdf_t, gmin_t, gmax_t, scene_id=sid, align_corners=self.align_corners)
        else:                                                                   # reference contract: dense [B,D,D,D]
            sid = torch.arange(s_grid_sdf_batch.shape[0], dtype=torch.int32, device=self.device)
            body_sdf = ops.sdf_sample(body_verts_batch, s_grid_sdf_batch, s_grid_min_batch, s_grid_max_batch, scene_id=sid,
                                      align_corners=self.align_corners)
        # data parallel: the mean runs over the penetrating vertices of the GLOBAL batch (one 2-float all-reduce, psi_release_amd/dist.py)
        pen = psi_dist.penetration_loss_global(body_sdf) if psi_dist.is_dist() else ops.penetration_loss(body_sdf)
        loss_sdf_pene = gate * self.weight_collision * pen
        return loss_contact, loss_vposer, loss_sdf_pene

    def _fused_glue(self, xh):
        """The [B,75] body-vector glue of cal_loss as three HIP launches (ops.cvae_target / ops.cvae_losses) instead of ~190 elementwise
        operators: the 75-D (6D rotation) layout on the GPU; ``fused_glue = False`` keeps the operator sequence."""
        return self.fused_decode and self.use_cont_rot and xh.is_cuda and xh.shape[1] == 72 and not xh.requires_grad and self.fused_glue

    def _fca(self, ep):
        if not self.loss_weight_anealing:
            return 1.0
        return min(1.0, max(float(ep) / (self.epoch * 0.75), 0))

    def _kl(self, mu, logsigma2, ep):
        fca = self._fca(ep) if self._fca_t is None else self._fca_t      # graph mode: a device scalar refreshed per step
        return fca ** 2 * self.weight_loss_kl * 0.5 * torch.mean(torch.exp(logsigma2) + mu ** 2 - 1.0 - logsigma2)

    # -------------------------------------------------------------------------------------
    def _resume(self):
        starting_ep = 0
        if self.resume_training:
            ckp_list = sorted(glob.glob(os.path.join(self.save_dir, 'epoch-*.ckp')), key=os.path.getmtime)
            if len(ckp_list) > 0:
                checkpoint = torch.load(ckp_list[-1], map_location=self.device)
                self.model_h.load_state_dict(checkpoint['model_h_state_dict'])
                self.optimizer_h.load_state_dict(checkpoint['optimizer_h_state_dict'])
                starting_ep = checkpoint['epoch']
                print('[INFO] --resuming training from {}'.format(ckp_list[-1]))
        return starting_ep

    def _save(self, ep):
        if psi_dist.rank() != 0:
            return
        torch.save({'epoch': ep + 1, 'model_h_state_dict': self.model_h.state_dict(),
                    'optimizer_h_state_dict': self.optimizer_h.state_dict()},
                   self.save_dir + "/epoch-{:06d}".format(ep + 1) + ".ckp")

    # ---- data-parallel gradient exchange: buckets that the gradients alias, reduced from autograd hooks while the backward pass is still
    # running (psi_release_amd/dist.py: GradBuckets; ``grad_bucket_mb`` sets the bucket size, default 16)
    def _buckets_begin(self):
        """Start of a step's gradient accumulation: zero the gradients.  Data parallel: through the buckets (created on first use)."""
        if not psi_dist.is_dist():
            self.optimizer_h.zero_grad(set_to_none=self.use_graph)
            return
        if getattr(self, '_buckets', None) is None:
            self._buckets = psi_dist.GradBuckets(self.model_h, float(self.grad_bucket_mb))
        self._buckets.begin()

    def _buckets_finish(self):
        if psi_dist.is_dist():
            self._buckets.finish()

    # ---- whole-step HIP graph ------------------------------------------------------------------------------------
    # A train_s{1,2} step is ~2500 small launches (rotation glue, BN, Adam, ...) and is bound by the host issuing them.
    # All shapes are static (fixed batch size; the per-body scene slot is data, not shape), so the step is captured once
    # per loss phase (scene terms gated off / on, train_s1.py:171-173) and replayed; the KL annealing factor is a device
    # scalar.  Replays read the batch from static buffers.
    def _static_batch(self, d):
        keep = lambda x: x.clone() if torch.is_tensor(x) else x
        st = [keep(x) for x in d[:11]]
        last = d[11]
        st.append(tuple(keep(x) for x in last) if isinstance(last, tuple) else keep(last))
        return st

    @staticmethod
    def _load_batch(st, d):
        indexed = isinstance(d[11], tuple) and len(d[11]) == 5
        for i in range(11):
            if i == 6 and indexed:
                continue                                   # the gathered [B,m,3] scene clouds are not read on the indexed path
            if torch.is_tensor(st[i]) and st[i].numel():
                st[i].copy_(d[i])
        if isinstance(d[11], tuple):
            st[11][1].copy_(d[11][1])                      # per-body scene slot; the tables themselves are shared
        else:
            st[11].copy_(d[11])

    def _capture(self, d, ep):
        import copy
        for g in self.optimizer_h.param_groups:
            g['capturable'] = True
        if self._fca_t is None:
            self._fca_t = torch.tensor(float(self._fca(ep)), device=self.device)
        self._fca_t.fill_(float(self._fca(ep)))
        st = self._static_batch(d)
        # warm-up on a side stream (workspace growth, RCCL channel setup) — on a snapshot, so that it does not train
        snap_m = copy.deepcopy(self.model_h.state_dict())
        snap_o = copy.deepcopy(self.optimizer_h.state_dict())
        rng = torch.cuda.get_rng_state(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(3):
                if psi_dist.is_dist():
                    self._buckets_begin()
                else:
                    self.optimizer_h.zero_grad(set_to_none=True)
                sum(self._losses_from_batch(st, ep)).backward()
                self._buckets_finish()
                self.optimizer_h.step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.model_h.load_state_dict(snap_m)
        self.optimizer_h.load_state_dict(snap_o)
        for g in self.optimizer_h.param_groups:
            g['capturable'] = True
        if not self.optimizer_h.state:                     # fresh optimiser: materialise its state outside the graph
            if psi_dist.is_dist():
                self._buckets_begin()                      # (zero gradients that keep aliasing their buckets)
                self._buckets._armed = False
            else:
                self.optimizer_h.zero_grad(set_to_none=True)
            for p in self.model_h.parameters():
                if p.requires_grad and p.grad is None:
                    p.grad = torch.zeros_like(p)
            lr = [g['lr'] for g in self.optimizer_h.param_groups]
            for g in self.optimizer_h.param_groups:
                g['lr'] = 0.0
            self.optimizer_h.step()                        # lr 0: creates exp_avg / exp_avg_sq / step and moves nothing
            for g, v in zip(self.optimizer_h.param_groups, lr):
                g['lr'] = v
            for stt in self.optimizer_h.state.values():
                stt['step'].zero_()
                stt['exp_avg'].zero_()
                stt['exp_avg_sq'].zero_()
        torch.cuda.set_rng_state(rng, self.device)
        graph = torch.cuda.CUDAGraph()
        if not psi_dist.is_dist():
            self.optimizer_h.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            if psi_dist.is_dist():
                self._buckets_begin()                      # (zeroing the buckets is part of the captured step; the collectives are its branches)
            losses = self._losses_from_batch(st, ep)
            sum(losses).backward()
            self._buckets_finish()
            self.optimizer_h.step()
        return {'graph': graph, 'batch': st, 'losses': losses}

    def _train_step_graph(self, train_data, ep):
        key = bool(ep > 0.75 * self.epoch)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = self._capture(train_data, ep)
        self._fca_t.fill_(float(self._fca(ep)))
        self._load_batch(g['batch'], train_data)
        g['graph'].replay()
        return g['losses']

    def train_step(self, train_data, ep):
        """One optimiser step on one batch (the body of the ``while batch_gen.has_next_batch()`` loop)."""
        if self.use_graph and psi_dist.is_dist() and torch.distributed.get_backend() != 'nccl':
            # a host-side collective (gloo: the bucket hooks synchronise the stream and wait for the work) cannot be part of a captured step
            print('[INFO][train] use_graph needs the RCCL backend for data-parallel runs (got %s): running eager steps' % torch.distributed.get_backend())
            self.use_graph = False
        if self.use_graph:
            return self._train_step_graph(train_data, ep)
        if psi_dist.is_dist():
            self._buckets_begin()
        else:
            self.optimizer_h.zero_grad()
        losses = self._losses_from_batch(train_data, ep)
        loss_h = sum(losses)
        loss_h.backward()
        self._buckets_finish()
        self.optimizer_h.step()
        return losses

    def _epoch_validity(self, batch_gen):
        """Data parallel: which of this epoch's batches EVERY rank has (batch_gen_hdf5.py:198-199,211-214 drop a short batch and a batch with a
        wrong PROX fitting) — decided ONCE per epoch from the sharded index (BatchGeneratorWithSceneMesh.epoch_batch_validity: host arrays
        only) with one all-reduce, instead of a host-synchronising collective in front of every step.  None: decide per step."""
        if not psi_dist.is_dist() or not hasattr(batch_gen, 'epoch_batch_validity'):
            return None
        import torch.distributed as tdist
        v = batch_gen.epoch_batch_validity(self.batch_size)
        t = torch.tensor([1.0 if x else 0.0 for x in v], device=self.device if tdist.get_backend() == 'nccl' else 'cpu')
        n = torch.tensor([float(len(v))], device=t.device)
        tdist.all_reduce(n, op=tdist.ReduceOp.MIN)
        if int(n.item()) != len(v):
            raise RuntimeError('ranks disagree on the number of batches of an epoch (%d here, %d elsewhere): shard the index evenly' % (len(v), int(n.item())))
        if len(v):
            tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
        return [bool(x > 0.5) for x in t.tolist()]

    def _all_ranks_have(self, train_data):
        """Skip decision of the batch loop (batch_gen_hdf5.py:198-199,211-214 return None).  It must be COLLECTIVE: train_step
        issues a gradient all-reduce, so a rank that skipped alone would leave the others waiting in it."""
        ok = train_data is not None
        if not psi_dist.is_dist():
            return ok
        import torch.distributed as tdist
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device if tdist.get_backend() == 'nccl' else 'cpu')
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def train(self, batch_gen):
        self.model_h.train()
        self.model_h.to(self.device)
        self.vposer.to(self.device)
        starting_ep = self._resume()
        print('--[INFO] start training')
        start_time = time.time()
        for ep in range(starting_ep, self.epoch):
            valid, bi = self._epoch_validity(batch_gen), 0
            while batch_gen.has_next_batch():
                train_data = batch_gen.next_batch(self.batch_size)
                if valid is not None:
                    ok, bi = (valid[bi] if bi < len(valid) else False), bi + 1
                    assert not ok or train_data is not None, 'epoch_batch_validity and next_batch disagree'
                    if not ok:
                        continue
                elif not self._all_ranks_have(train_data):
                    continue
                losses = self.train_step(train_data, ep)
                if self.verbose:
                    print(self._format(ep, losses))
                if (time.time() - start_time) / 3600.0 >= 2:
                    start_time = time.time()
                    self._save(ep)
            batch_gen.reset()
            if (ep + 1) % 10 == 0:
                self._save(ep)
        if self.verbose:
            print('[INFO]: Training completes!')
            print()


class TrainOP(_TrainBase):
    """Stage-1 trainer, train_s1.py:37-338.  ``cal_loss`` returns [rec_t, rec_p, KL, contact, vposer, sdf_pene]."""
    stage = 's1'

    def _make_model(self, n_dim_body):
        return HumanCVAES1(latentD=self.model_h_latentD, scene_model_ckpt=self.scene_model_ckpt, n_dim_body=n_dim_body,
                           autocast_bf16=self.autocast_bf16).to(self.device)

    def cal_loss(self, xs, xh, cam_ext, cam_int, max_d, scene_verts, scene_face, s_grid_min_batch, s_grid_max_batch,
                 s_grid_sdf_batch, ep, eps=None):
        if self._fused_glue(xh):
            xhnr = ops.cvae_target(xh, cam_int, max_d)
            xhnr_rec, mu, logsigma2 = self.model_h(xhnr, xs, eps=eps)
            xh_rec, L = ops.cvae_losses(xhnr_rec, xhnr, xh, cam_int, max_d, mu, logsigma2, fca=self._fca(ep) if self._fca_t is None else self._fca_t,
                                        w_rec=self.weight_loss_rec_h, w_kl=self.weight_loss_kl, w_vposer=self.weight_loss_vposer)
            loss_rec_t, loss_rec_p, loss_KL, _, loss_vposer = L.unbind(0)
            loss_contact, loss_vposer, loss_sdf_pene = self._scene_losses(xh_rec, cam_ext, scene_verts, s_grid_min_batch, s_grid_max_batch,
                                                                          s_grid_sdf_batch, ep, loss_vposer=loss_vposer)
            return [loss_rec_t, loss_rec_p, loss_KL, loss_contact, loss_vposer, loss_sdf_pene]
        xhn = GeometryTransformer.normalize_global_T(xh, cam_int, max_d)
        xhnr = GeometryTransformer.convert_to_6D_rot(xhn)
        xhnr_rec, mu, logsigma2 = self.model_h(xhnr, xs, eps=eps)
        if self.fused_decode and self.use_cont_rot:
            xh_rec = GeometryTransformer.recover_global_T(xhnr_rec, cam_int, max_d)     # stays 75-D; only the translation changes
        else:
            xhn_rec = GeometryTransformer.convert_to_3D_rot(xhnr_rec)
            xh_rec = GeometryTransformer.recover_global_T(xhn_rec, cam_int, max_d)
        loss_rec_t = self.weight_loss_rec_h * (0.5 * F.l1_loss(xhnr_rec[:, :3], xhnr[:, :3]) + 0.5 * F.l1_loss(xh_rec[:, :3], xh[:, :3]))
        loss_rec_p = self.weight_loss_rec_h * F.l1_loss(xhnr_rec[:, 3:], xhnr[:, 3:])
        loss_KL = self._kl(mu, logsigma2, ep)
        loss_contact, loss_vposer, loss_sdf_pene = self._scene_losses(xh_rec, cam_ext, scene_verts, s_grid_min_batch,
                                                                      s_grid_max_batch, s_grid_sdf_batch, ep)
        return [loss_rec_t, loss_rec_p, loss_KL, loss_contact, loss_vposer, loss_sdf_pene]

    def _losses_from_batch(self, d, ep):
        return self.cal_loss(xs=torch.cat([d[0], d[1]], dim=1), xh=d[2], cam_ext=d[3], cam_int=d[4], max_d=d[5],
                             scene_verts=d[6], scene_face=d[7], s_grid_min_batch=d[8], s_grid_max_batch=d[9],
                             s_grid_sdf_batch=d[11], ep=ep)

    def _format(self, ep, l):
        return "---in [epoch {:d}]: rec_t={:f}, rec_p={:f}, kl={:f}, vp={:f}, contact={:f}, collision={:f}".format(
            ep + 1, l[0].item(), l[1].item(), l[2].item(), l[4].item(), l[3].item(), l[5].item())


class TrainOPS2(_TrainBase):
    """Stage-2 trainer, train_s2.py:37-340.  ``cal_loss`` returns (rec_t, rec_p, KL_g, KL_l, contact, vposer, sdf_pene)."""
    stage = 's2'

    def _make_model(self, n_dim_body):
        return HumanCVAES2(latentD_g=self.model_h_latentD, latentD_l=self.model_h_latentD, scene_model_ckpt=self.scene_model_ckpt,
                           n_dim_body=n_dim_body, autocast_bf16=self.autocast_bf16).to(self.device)

    def cal_loss(self, xs, xh, eps_g, eps_l, cam_ext, cam_int, max_d, scene_verts, scene_face, s_grid_min_batch,
                 s_grid_max_batch, s_grid_sdf_batch, ep, use_eps=False):
        if self._fused_glue(xh):
            xhnr = ops.cvae_target(xh, cam_int, max_d)
            xhnr_rec, mu_g, lv_g, mu_l, lv_l = self.model_h(xhnr, eps_g, eps_l, xs, use_eps=use_eps)
            xh_rec, L = ops.cvae_losses(xhnr_rec, xhnr, xh, cam_int, max_d, mu_g, lv_g, mu_l, lv_l,
                                        fca=self._fca(ep) if self._fca_t is None else self._fca_t,
                                        w_rec=self.weight_loss_rec_h, w_kl=self.weight_loss_kl, w_vposer=self.weight_loss_vposer)
            loss_rec_t, loss_rec_p, loss_KL_g, loss_KL_l, loss_vposer = L.unbind(0)
            loss_contact, loss_vposer, loss_sdf_pene = self._scene_losses(xh_rec, cam_ext, scene_verts, s_grid_min_batch, s_grid_max_batch,
                                                                          s_grid_sdf_batch, ep, loss_vposer=loss_vposer)
            return loss_rec_t, loss_rec_p, loss_KL_g, loss_KL_l, loss_contact, loss_vposer, loss_sdf_pene
        xhn = GeometryTransformer.normalize_global_T(xh, cam_int, max_d)
        xhnr = GeometryTransformer.convert_to_6D_rot(xhn)
        xhnr_rec, mu_g, lv_g, mu_l, lv_l = self.model_h(xhnr, eps_g, eps_l, xs, use_eps=use_eps)
        if self.fused_decode and self.use_cont_rot:
            xh_rec = GeometryTransformer.recover_global_T(xhnr_rec, cam_int, max_d)     # stays 75-D; only the translation changes
        else:
            xhn_rec = GeometryTransformer.convert_to_3D_rot(xhnr_rec)
            xh_rec = GeometryTransformer.recover_global_T(xhn_rec, cam_int, max_d)
        loss_rec_t = self.weight_loss_rec_h * (0.5 * F.l1_loss(xhnr_rec[:, :3], xhnr[:, :3]) + 0.5 * F.l1_loss(xh_rec[:, :3], xh[:, :3]))
        loss_rec_p = self.weight_loss_rec_h * F.l1_loss(xhnr_rec[:, 3:], xhnr[:, 3:])
        loss_KL_g, loss_KL_l = self._kl(mu_g, lv_g, ep), self._kl(mu_l, lv_l, ep)
        loss_contact, loss_vposer, loss_sdf_pene = self._scene_losses(xh_rec, cam_ext, scene_verts, s_grid_min_batch,
                                                                      s_grid_max_batch, s_grid_sdf_batch, ep)
        return loss_rec_t, loss_rec_p, loss_KL_g, loss_KL_l, loss_contact, loss_vposer, loss_sdf_pene

    def _losses_from_batch(self, d, ep):
        B = d[2].shape[0]
        noise_l = torch.randn([B, self.model_h_latentD], dtype=torch.float32, device=self.device)   # train_s2.py:262-267 (unused by the model)
        noise_g = torch.randn([B, self.model_h_latentD], dtype=torch.float32, device=self.device)
        return self.cal_loss(xs=torch.cat([d[0], d[1]], dim=1), xh=d[2], eps_g=noise_g, eps_l=noise_l, cam_ext=d[3], cam_int=d[4],
                             max_d=d[5], scene_verts=d[6], scene_face=d[7], s_grid_min_batch=d[8], s_grid_max_batch=d[9],
                             s_grid_sdf_batch=d[11], ep=ep)

    def _format(self, ep, l):
        return "---in [epoch {:d}]: rec_t={:f}, rec_p={:f}, kl_g={:f}, kl_l={:f}, vp={:f}, contact={:f}, collision={:f}".format(
            ep + 1, l[0].item(), l[1].item(), l[2].item(), l[3].item(), l[5].item(), l[4].item(), l[6].item())
