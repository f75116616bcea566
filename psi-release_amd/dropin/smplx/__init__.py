"""Drop-in for ``import smplx`` as PSI uses it (``smplx.create(...)``, fitting_proxe.py:32,55-69): the SMPL-X layer on the
HIP LBS kernels.  Only ``model_type='smplx'`` (the one PSI uses) is provided."""
from psi_release_amd.body_model import SMPLXLayer, create  # noqa: F401
