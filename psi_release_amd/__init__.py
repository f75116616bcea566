"""Importable alias for the ``psi-release_amd/`` package directory.

The product lives in ``psi-release_amd/`` (the name the build contract uses); a
hyphen is not a legal Python identifier, so this stub extends ``__path__`` to
that directory.  ``import psi_release_amd.fitting`` resolves to
``psi-release_amd/fitting.py``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "psi-release_amd")
__path__.append(_real)
REPO_ROOT = _os.path.dirname(_real)
PKG_DIR = _real
