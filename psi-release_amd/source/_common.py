"""Shared bits of the entry-point scripts (path setup, synthetic stand-in assets for the licensed data)."""
import os
import pickle
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def synthetic_prox_tree(root, scenes, m=8192, D=64, n_contact=512, n_files=2, batch=1, seed=0, write=True):
    """Write a PROX-E-shaped directory tree of synthetic assets (scenes_sdf/, scenes_downsampled/, body_segments/,
    generated bodies) and return (proxe_path, gen_path, smplx_data, vposer_state)."""
    from psi_release_amd import synth
    os.makedirs(root, exist_ok=True)
    proxe = os.path.join(root, 'PROXE')
    gen = os.path.join(root, 'gen')
    for si, name in enumerate(scenes if write else []):
        sc = synth.make_scene(seed + si, m, D, n_contact)
        sc.write_prox_layout(proxe, name)
        os.makedirs(os.path.join(gen, name), exist_ok=True)
        for ii in range(n_files):
            with open(os.path.join(gen, name, 'body_gen_{:06d}.pkl'.format(ii)), 'wb') as f:
                pickle.dump(synth.make_bodies(100 * si + ii, batch), f)
    return proxe, gen, synth.make_smplx(7), synth.make_vposer_state(3)


def dist_setup():
    """Join the torchrun job this script was started under (RANK / WORLD_SIZE / MASTER_* in the environment; backend nccl = RCCL
    on GPUs); a plain `python fitting_proxe.py ...` stays one process.  Returns (rank, world)."""
    from psi_release_amd import dist as psi_dist
    rank, _, world = psi_dist.init_from_env()
    return rank, world


def fit_files(fop_cls, fittingconfig, lossconfig, gen_dir, fit_dir, max_files, shard, rank, world, concurrency=1, pack=1):
    """The per-scene file loop of the fitting entry points (fitting_proxe.py:252-263) on `world` ranks.

    shard='files': rank r fits the pkl files r, r+world, ... — every file is an independent problem (own batch, own loss
    normalisers), no collective on the data path.
    concurrency > 1 (with shard='files'): a rank keeps that many of ITS engine runs in flight at once, each on its own engine and stream.
    pack > 1 (with shard='files'): `pack` files form ONE engine run with per-body loss normalisers (FittingOP.independent_bodies) —
    the same result as fitting them one by one WITH A FRESH ADAM STATE PER FILE (--reset_optimizer semantics: fitting_many resets the
    optimizer for every run, whereas the default file loop carries one optimizer across the files of a scene like the reference,
    fitting_proxe.py:73-74), at the cost of one.
    shard='rows' : every rank opens every file and fits rows [r*B/world, (r+1)*B/world) of its B bodies; the loss normalisers are
    global through the one all-reduce per iteration (psi_release_amd/dist.py), rank 0 gathers the rows and writes the pkl."""
    import os
    import pickle
    import torch
    from psi_release_amd import dist as psi_dist
    B = fittingconfig['batch_size']
    cfg = dict(fittingconfig)
    if shard == 'rows' and world > 1:
        if concurrency > 1 or pack > 1:
            raise SystemExit('--concurrency / --pack fit INDEPENDENT files side by side; they cannot be combined with --shard rows '
                             '(one file at a time, its batch split over the ranks)')
        if B % world:
            raise SystemExit('--shard rows needs batch_size %% world_size == 0 (got %d / %d)' % (B, world))
        cfg['batch_size'] = B // world
        cfg['data_parallel'] = None
    else:
        cfg['data_parallel'] = False
        if pack > 1:
            cfg['batch_size'] = B * pack
            cfg['independent_bodies'] = True
    fop = fop_cls(cfg, lossconfig)
    todo = []
    for ii in range(max_files):
        inp = os.path.join(gen_dir, 'body_gen_{:06d}.pkl'.format(ii))
        outp = os.path.join(fit_dir, 'body_gen_{:06d}.pkl'.format(ii))
        if os.path.exists(inp) and not os.path.exists(outp):
            todo.append((inp, outp))
    if world > 1:                      # all ranks must agree on the work list (a rank may have listed the directory a moment later)
        lst = [todo]
        torch.distributed.broadcast_object_list(lst, src=0)
        todo = lst[0]
    if shard == 'rows' and world > 1:
        per = B // world
        for inp, outp in todo:
            with open(inp, 'rb') as f:
                rec = pickle.load(f)
            lo = rank * per
            rows = {k: (v[lo:lo + per] if getattr(v, 'shape', (0,))[0] == B else v) for k, v in rec.items()}
            xh = fop.fitting(rows)
            xh_all = psi_dist.gather_rows(xh.detach())
            cam_ext, cam_int = psi_dist.gather_rows(fop.cam_ext) if fop.cam_ext.shape[0] == per else fop.cam_ext, \
                psi_dist.gather_rows(fop.cam_int) if fop.cam_int.shape[0] == per else fop.cam_int
            if rank == 0:
                fop.cam_ext, fop.cam_int = cam_ext, cam_int
                fop.save_result(xh_all, outp)
    else:
        mine = todo[rank::world] if world > 1 else todo
        if (concurrency > 1 or pack > 1) and mine:
            # independent files in flight together, one fused engine + stream each (FittingOP.fitting_many)
            results, cams = fop.fitting_many([inp for inp, _ in mine], concurrency)
            for (inp, outp), xh, (cam_ext, cam_int) in zip(mine, results, cams):
                fop.cam_ext, fop.cam_int = cam_ext, cam_int
                fop.save_result(xh, outp)
        else:
            for inp, outp in mine:
                fop.save_result(fop.fitting(inp), outp)
    return len(todo)
