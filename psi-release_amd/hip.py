"""ctypes binding of libpsi_hip.so (include/psi_hip.h).  PyTorch supplies device memory and streams only.

The library is REQUIRED: importing an op that needs it raises if it is missing — there is no
CPU or eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# PSI_CHAMFER_FMA=1 selects the build whose Chamfer distance is contracted the way nvcc's default --fmad=true contracts the
# reference's expression (include/psi_hip.h: psi_chamfer_arith_mode); PSI_HIP_LIB: development A/B builds
LIB_PATH = os.environ.get('PSI_HIP_LIB') or os.path.join(
    _PKG, 'lib', 'libpsi_hip_fma.so' if os.environ.get('PSI_CHAMFER_FMA') == '1' else 'libpsi_hip.so')
_lib = None

c_void_p, c_int, c_long, c_size_t, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); every symbol declared in include/psi_hip.h must appear here
SIGNATURES = {
    'psi_last_error': (ctypes.c_char_p, []),
    'psi_version': (c_int, []),
    'psi_device_info': (c_int, [c_void_p] * 4),
    'psi_chamfer_arith_mode': (c_int, []),
    'psi_chamfer_workspace_bytes': (c_size_t, [c_int] * 3),
    'psi_chamfer_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    'psi_chamfer_backward': (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_void_p]),
    'psi_scratch_release': (c_int, [c_void_p, c_int]),
    'psi_nn_index_create': (c_int, [c_void_p, c_void_p, c_int]),
    'psi_nn_index_destroy': (None, [c_void_p]),
    'psi_nn_index_query': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_nn_index_set_create': (c_int, [c_void_p, c_void_p, c_int]),
    'psi_nn_index_set_destroy': (None, [c_void_p]),
    'psi_nn_index_set_query': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'psi_sdf_sample_forward': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p] * 3),
    'psi_sdf_sample_backward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'psi_sdf_penetration_stats': (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    'psi_lbs_create': (c_int, [c_void_p] * 7 + [c_int] * 3),
    'psi_lbs_destroy': (None, [c_void_p]),
    'psi_lbs_workspace_floats': (c_size_t, [c_void_p, c_int]),
    'psi_lbs_forward': (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 4),
    'psi_lbs_backward': (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 5),
    'psi_linear_workspace_floats': (c_size_t, [c_int] * 3),
    'psi_linear_forward': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    'psi_linear_forward3': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    'psi_linear_backward_workspace_floats': (c_size_t, [c_int, c_int, c_int]),
    'psi_linear_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    'psi_linear_backward3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_adam_step': (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p] + [c_double] * 5 + [c_void_p]),
    'psi_fit_create': (c_int, [c_void_p] * 17),
    'psi_fit_destroy': (None, [c_void_p]),
    'psi_fit_set_problem': (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    'psi_fit_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'psi_fit_backward_step': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'psi_fit_iterate': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'psi_bn_workspace_floats': (c_size_t, [c_long, c_int]),
    'psi_bn_forward': (c_int, [c_void_p] * 7 + [c_long, c_int, c_int, c_float, c_float] + [c_void_p] * 5),
    'psi_bn_backward': (c_int, [c_void_p] * 6 + [c_long, c_int, c_int] + [c_void_p] * 6),
    'psi_bn_forward_t': (c_int, [c_void_p, c_int] + [c_void_p] * 6 + [c_long, c_int, c_int, c_float, c_float] + [c_void_p] * 4 + [c_int, c_void_p]),
    'psi_bn_backward_t': (c_int, [c_void_p, c_int] + [c_void_p] * 6 + [c_long, c_int, c_int] + [c_void_p] * 6),
    'psi_maxpool3x3s2_forward_t': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'psi_maxpool3x3s2_backward_t': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'psi_conv2d_supported': (c_int, [c_int] * 6),
    'psi_conv2d_forward': (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_int, c_int, c_void_p]),
    'psi_conv2d_input_grad': (c_int, [c_void_p, c_int, c_void_p] + [c_int] * 9 + [c_void_p, c_int, c_int, c_void_p]),
    'psi_conv2d_prepared_ok': (c_int, [c_int] * 6),
    'psi_conv2d_prepare_weight': (c_int, [c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, c_void_p]),
    'psi_conv2d_forward_p': (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_int, c_int, c_void_p]),
    'psi_conv2d_input_grad_p': (c_int, [c_void_p, c_int, c_void_p] + [c_int] * 9 + [c_void_p, c_int, c_int, c_void_p]),
    'psi_conv2d_wgrad_workspace_floats': (c_size_t, [c_int] * 9),
    'psi_conv2d_weight_grad': (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 9 + [c_void_p, c_void_p, c_int, c_void_p]),
    'psi_conv3x3_supported': (c_int, [c_int] * 4),
    'psi_conv3x3_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'psi_conv3x3_wrw_workspace_floats': (c_size_t, [c_int] * 5),
    'psi_conv3x3_weight_grad': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'psi_conv3x3_rotate_weight': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'psi_conv3x3_prepare_weight': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'psi_maxpool3x3s2_forward': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'psi_maxpool3x3s2_backward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'psi_cvae_target': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'psi_cvae_losses_workspace_floats': (c_size_t, []),
    'psi_cvae_losses_forward': (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_cvae_losses_backward': (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_scene_losses_workspace_floats': (c_size_t, []),
    'psi_scene_losses_forward': (c_int, [c_void_p, c_long, c_void_p, c_long, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_scene_losses_backward': (c_int, [c_void_p] * 7 + [c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_float,
                                          c_void_p, c_void_p, c_void_p]),
    'psi_contact_slot_chain': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'psi_fit_dp_mode': (c_int, [c_void_p]),
    'psi_stream_wait': (c_int, [c_void_p, c_int]),
    'psi_dp_unique_id': (c_int, [c_void_p]),
    'psi_dp_comm_create': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'psi_dp_comm_destroy': (None, [c_void_p]),
    'psi_dp_comm_info': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_dp_allreduce_sum': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'psi_fit_iterate_dp': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'psi_fit_read': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'psi_fit_read_losses': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'psi_fit_decode_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_fit_decode_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'psi_fit_profile': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'psi_fit_copy_buffer': (c_int, [c_void_p, ctypes.c_char_p, c_void_p, c_long, c_void_p]),
}


class FitConfig(ctypes.Structure):
    """struct psi_fit_config (include/psi_hip.h)."""
    _fields_ = [('B', c_int), ('n_contact', c_int), ('m_scene', c_int), ('D', c_int), ('align_corners', c_int),
                ('world_size', c_int), ('num_pca_comps', c_int), ('max_history', c_int), ('nn_mode', c_int),
                ('w_rec', c_float), ('w_vposer', c_float), ('w_contact', c_float), ('w_collision', c_float),
                ('contact_const', c_float), ('lr', c_float), ('beta1', c_float), ('beta2', c_float), ('eps', c_float),
                ('independent_bodies', c_int), ('concurrent_engines', c_int),
                ('lr_d', ctypes.c_double), ('beta1_d', ctypes.c_double), ('beta2_d', ctypes.c_double)]


class PsiHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises PsiHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PsiHipError('libpsi_hip.so not found at %s — run `python -m psi_release_amd.build` '
                              '(there is no CPU fallback)' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get('PSI_HIP_LIB') and not hasattr(l, name):
                continue                    # partial development build
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    return lib().psi_last_error().decode()


def check(rc: int, what: str):
    if rc != 0:
        raise PsiHipError('%s failed (code %d): %s' % (what, rc, lib().psi_last_error().decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be a contiguous CUDA(HIP) tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PsiHipError('expected a GPU tensor (the HIP kernels are the only implementation)')
    if not t.is_contiguous():
        raise PsiHipError('expected a contiguous tensor')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def device_info():
    cu, ws, clk, is950 = c_int(), c_int(), c_int(), c_int()
    check(lib().psi_device_info(ctypes.byref(cu), ctypes.byref(ws), ctypes.byref(clk), ctypes.byref(is950)),
          'psi_device_info')
    return {'cu_count': cu.value, 'wave_size': ws.value, 'clock_khz': clk.value, 'gfx950': bool(is950.value)}
