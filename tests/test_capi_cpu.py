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
