"""CPU checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every symbol
that include/psi_hip.h declares (no compute is called: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'psi_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(psi_[a-z0-9_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def built_lib():
    from psi_release_amd import build
    path = build.build()
    assert os.path.exists(path)
    return path


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), 'libpsi_hip.so does not export %s' % s


def test_fma_mode_library_exports_the_same_abi(built_lib):
    """libpsi_hip_fma.so (Chamfer distance in nvcc --fmad=true form) is the same ABI; the mode is a build-time property."""
    from psi_release_amd import build
    assert os.path.exists(build.LIB_FMA)
    lib = ctypes.CDLL(build.LIB_FMA)
    for s in declared_symbols():
        assert hasattr(lib, s), 'libpsi_hip_fma.so does not export %s' % s
    assert lib.psi_chamfer_arith_mode() == 1 and ctypes.CDLL(built_lib).psi_chamfer_arith_mode() == 0


def test_oracle_builds_in_both_arithmetic_modes():
    import numpy as np
    import psi_oracle as O
    rs = np.random.RandomState(0)
    x, y = rs.standard_normal((2, 50, 3)).astype(np.float32), rs.standard_normal((2, 700, 3)).astype(np.float32)
    a, b = O.chamfer_nn_np(x, y, fma=False), O.chamfer_nn_np(x, y, fma=True)
    ref = ((x[:, :, None, :].astype(np.float64) - y[:, None, :, :]) ** 2).sum(-1)
    for d1, i1 in ((a[0], a[1]), (b[0], b[1])):
        assert np.abs(d1 - ref.min(2)).max() < 1e-5 and (i1 == ref.argmin(2)).mean() > 0.98
    assert np.abs(a[0] - b[0]).max() <= 4 * np.finfo(np.float32).eps * np.abs(a[0]).max()
    # the fma form is the exactly-rounded-once-per-step evaluation: reproduce it in float64 arithmetic with explicit roundings
    d = (y[0, b[1][0]] - x[0]).astype(np.float32)
    f32 = np.float32
    acc = (d[:, 0] * d[:, 0]).astype(f32)
    acc = (d[:, 1].astype(np.float64) * d[:, 1].astype(np.float64) + acc.astype(np.float64)).astype(f32)
    acc = (d[:, 2].astype(np.float64) * d[:, 2].astype(np.float64) + acc.astype(np.float64)).astype(f32)
    assert np.array_equal(acc, b[0][0])


def test_binding_table_matches_header(built_lib):
    from psi_release_amd import hip
    assert sorted(hip.SIGNATURES) == declared_symbols()
    assert hip.lib().psi_version() >= 100


def test_code_object_is_gfx950(built_lib):
    data = open(built_lib, 'rb').read()
    assert b'gfx950' in data
    assert b'gfx942' not in data and b'sm_' not in data


def test_ops_refuse_cpu_tensors(built_lib):
    """The product has no CPU fallback: CPU tensors are rejected instead of silently computed elsewhere."""
    import torch
    from psi_release_amd import ops, hip
    with pytest.raises(hip.PsiHipError):
        ops.chamfer_forward_raw(torch.zeros(1, 4, 3), torch.zeros(1, 5, 3))


def test_fit_config_struct_matches_header():
    """`hip.FitConfig` (ctypes) must list the fields of `struct psi_fit_config` (include/psi_hip.h) with the same names, order and
    types — a field added on one side only would shift every later field of the struct the engine is created from."""
    import ctypes
    import re
    from psi_release_amd import hip
    text = open(os.path.join(ROOT, 'include', 'psi_hip.h')).read()
    body = re.search(r'typedef struct psi_fit_config \{(.*?)\} psi_fit_config;', text, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for nm in names.split(','):
            fields.append((nm.strip(), ctype))
    want = {'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double}
    got = [(n, t) for n, t in hip.FitConfig._fields_]
    assert [n for n, _ in got] == [n for n, _ in fields]
    assert all(t is want[c] for (_, t), (_, c) in zip(got, fields))


def test_synthetic_smplx_sparse_weight_rows():
    """`synth.make_smplx(weight_nnz=4)`: exactly four non-zero skinning weights per vertex, rows sum to one, and the four joints are
    a joint with neighbours of its own in the kinematic tree (the structure of the released SMPL-X model's `weights`)."""
    import numpy as np
    from psi_release_amd import synth
    m = synth.make_smplx(7, weight_nnz=4)
    w = np.asarray(m.weights)
    assert w.shape == (synth.V_SMPLX, synth.J_SMPLX)
    assert ((w != 0).sum(1) == 4).all() and np.abs(w.sum(1) - 1).max() < 1e-6 and (w >= 0).all()
    par = np.asarray(synth.SMPLX_PARENTS)
    for v in (0, 777, 5000, synth.V_SMPLX - 1):
        js = set(np.nonzero(w[v])[0].tolist())
        # connected in the tree: every joint of the row has its parent or one of its children in the row
        assert all((par[j] in js) or any(par[c] == j for c in js) for j in js)
    d = np.asarray(synth.make_smplx(7).weights)
    assert (d != 0).sum(1).min() > synth.J_SMPLX - 8               # the default stays dense (more than PSI_WNZ = 8 non-zeros everywhere)


def test_workspace_queries_are_host_functions(built_lib):
    """The *_workspace_floats() / *_blocks() queries run on the host (callers size buffers before any launch).  Their values pin the
    launch shapes of the batch-sized dense layers: the contraction is split until a launch has about one workgroup per compute unit."""
    from psi_release_amd import hip
    L = hip.lib()
    M = 128
    # forward split-K: N / 32 column tiles, S = min(256 / tiles, K / 128) slices of [M, N] partial sums
    assert L.psi_linear_workspace_floats(M, 1024, 1024) == 8 * M * 1024
    assert L.psi_linear_workspace_floats(M, 32, 1024) == 8 * M * 32            # a 32-wide latent head: 1 tile, 8 slices of 128
    assert L.psi_linear_workspace_floats(M, 512, 32768) == 16 * M * 512        # the 32768 -> 512 scene-feature layer: 16 tiles x 16 slices
    assert L.psi_linear_workspace_floats(M, 32, 32) == 0 and L.psi_linear_workspace_floats(M, 128, 128) == 0
    # dX split over n: K / 64 column tiles, S = min(256 / tiles, N / 128) slices of [M, K]
    assert L.psi_linear_backward_workspace_floats(M, 1024, 1024) == 8 * M * 1024
    assert L.psi_linear_backward_workspace_floats(M, 512, 32768) == 0          # 512 column tiles fill the chip on their own
    assert L.psi_linear_backward_workspace_floats(M, 32, 1024) == 0            # one 64-wide tile of n: nothing to split
    assert L.psi_linear_workspace_floats(0, 8, 8) == 0 and L.psi_linear_backward_workspace_floats(M, 0, 8) == 0
    assert L.psi_cvae_losses_workspace_floats() == 64 * 6 and L.psi_scene_losses_workspace_floats() == 128 * 3
    # conv3x3 weight gradient: one 64 x 64 tile of dW for a 64 -> 64 layer, split over 256 pixel-stage groups
    assert L.psi_conv3x3_wrw_workspace_floats(128, 32, 32, 64, 64) == 256 * 64 * 9 * 64
    assert L.psi_conv3x3_wrw_workspace_floats(128, 30, 30, 64, 64) == 0        # shape not covered -> 0 (callers fall back to the library)
    assert L.psi_conv3x3_supported(64, 64, 32, 32) == 1 and L.psi_conv3x3_supported(64, 32, 32, 32) == 0
