"""Drop-in for the reference's ``chamfer_pytorch/dist_chamfer.py`` (``import chamfer_pytorch.dist_chamfer as ext``,
fitting_proxe.py:34): put ``psi-release_amd/dropin`` (and the repository root) on sys.path ahead of the reference's own
package.  ``ext.chamferDist()(xyz1, xyz2) -> (dist1, dist2)`` runs the HIP kernels of libpsi_hip.so."""
from psi_release_amd.ops import chamferDist, chamferFunction  # noqa: F401
