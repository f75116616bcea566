"""Shared bits of the entry-point scripts (path setup, synthetic stand-in assets for the licensed data)."""
import os
import pickle
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def synthetic_prox_tree(root, scenes, m=8192, D=64, n_contact=512, n_files=2, batch=1, seed=0):
    """Write a PROX-E-shaped directory tree of synthetic assets (scenes_sdf/, scenes_downsampled/, body_segments/,
    generated bodies) and return (proxe_path, gen_path, smplx_data, vposer_state)."""
    from psi_release_amd import synth
    os.makedirs(root, exist_ok=True)
    proxe = os.path.join(root, 'PROXE')
    gen = os.path.join(root, 'gen')
    for si, name in enumerate(scenes):
        sc = synth.make_scene(seed + si, m, D, n_contact)
        sc.write_prox_layout(proxe, name)
        os.makedirs(os.path.join(gen, name), exist_ok=True)
        for ii in range(n_files):
            with open(os.path.join(gen, name, 'body_gen_{:06d}.pkl'.format(ii)), 'wb') as f:
                pickle.dump(synth.make_bodies(100 * si + ii, batch), f)
    return proxe, gen, synth.make_smplx(7), synth.make_vposer_state(3)
