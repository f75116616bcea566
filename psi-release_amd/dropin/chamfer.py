"""Drop-in for the reference's compiled ``chamfer`` module (pybind11 over chamfer.cu, chamfer_pytorch/chamfer_cuda.cpp:17-33):
``import chamfer`` with ``psi-release_amd/dropin`` on sys.path, as the reference's own ``chamfer_pytorch/dist_chamfer.py:8``
does.  Same calling convention: the CALLER allocates (zero-filled, contiguous, fp32 / int32, on the GPU) every output
(dist_chamfer.py:19-30,40-45), the functions write in place on the current stream and return 1 for success, 0 for failure
(chamfer.cu:145-152).  ``backward`` accumulates into the zero-filled gradient tensors exactly like NmDistanceGradKernel."""
from psi_release_amd import hip as _hip


def _chk(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError('chamfer: expected contiguous GPU tensors (dist_chamfer.py:19-30 allocates them with .cuda())')


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda.cpp:17-19 -> chamfer_cuda_forward (chamfer.cu:136-154)."""
    _chk(xyz1, xyz2, dist1, dist2, idx1, idx2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = _hip.lib().psi_chamfer_forward(xyz1.data_ptr(), xyz2.data_ptr(), B, n, m, dist1.data_ptr(), idx1.data_ptr(),
                                        dist2.data_ptr(), idx2.data_ptr(), None, _hip.stream())
    return 1 if rc == 0 else 0


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_cuda.cpp:22-27 -> chamfer_cuda_backward (chamfer.cu:176-196)."""
    _chk(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    rc = _hip.lib().psi_chamfer_backward(xyz1.data_ptr(), xyz2.data_ptr(), gradxyz1.data_ptr(), gradxyz2.data_ptr(),
                                         graddist1.data_ptr(), graddist2.data_ptr(), idx1.data_ptr(), idx2.data_ptr(), B, n, m,
                                         _hip.stream())
    return 1 if rc == 0 else 0
