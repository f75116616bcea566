import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, 'oracle')
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


@pytest.fixture(scope='session')
def smplx_data():
    from psi_release_amd import synth
    return synth.make_smplx(7)


@pytest.fixture(scope='session')
def vposer_sd():
    from psi_release_amd import synth
    return synth.make_vposer_state(3)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
