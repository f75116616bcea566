"""Co-residency stress of the in-kernel cluster exchange (fit.hip: head / tail kernels spread a body over 8 workgroups that hand partial
sums to the cluster's last workgroup through tagged 64-bit words; the reader assumes its producers were dispatched before it).

The situation of BASELINE configs[4] on one GPU: seven engines (the seven MP3D-R rooms of fitting_habitat.py:238-241) in flight on seven
streams, plus other kernels resident on the chip — here an RCCL all-reduce loop on the library's own communicator and a loop of
brute-force Chamfer launches (a 0.3 ms kernel that fills every CU) on two more streams.  2000 iterations per engine.  Every engine must
(a) finish with its error word clear (a cluster exchange that gave up waiting raises it, psi_fit_read) and (b) reproduce the result of the
same fit run ALONE bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

from psi_release_amd import fitting, hip, ops, synth
from test_fitting_gpu import make_op

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ROOMS, B, ITERS = 7, 4, 2000          # 4 bodies x 7 engines x 8-wide clusters = 224 workgroups per head / tail launch


def _engine_ops(smplx_data, vposer_sd):
    ops_, bodies = [], []
    for r in range(ROOMS):
        op = make_op(smplx_data, vposer_sd, synth.make_scene(10 + r, 3000, 24, 300), B, 'fused', num_iter=ITERS, lr=0.02,
                     cls=fitting.FittingOPHabitat)
        op.concurrent_engines = ROOMS
        b = synth.make_bodies(70 + r, B)
        b['cam_ext'] = synth.make_cam_ext(70 + r, 1)
        ops_.append(op)
        bodies.append(b)
    return ops_, bodies


def test_seven_engines_beside_rccl_and_chamfer_loops(smplx_data, vposer_sd):
    L = hip.lib()
    # alone, one after the other
    ops_, bodies = _engine_ops(smplx_data, vposer_sd)
    alone = []
    for op, b in zip(ops_, bodies):
        alone.append(op.fitting(dict(b)).detach().cpu().numpy().copy())
        assert op._fused.read(0)[2] == ITERS
    del ops_
    torch.cuda.synchronize()
    # together, with two streams of foreign work
    ops_, bodies = _engine_ops(smplx_data, vposer_sd)
    runners = [op.make_step_runner(dict(b)) for op, b in zip(ops_, bodies)]
    idb = ctypes.create_string_buffer(128)
    hip.check(L.psi_dp_unique_id(idb), 'psi_dp_unique_id')
    comm = ctypes.c_void_p()
    hip.check(L.psi_dp_comm_create(ctypes.byref(comm), idb, 0, 1), 'psi_dp_comm_create')
    s_rccl, s_cham = torch.cuda.Stream(), torch.cuda.Stream()
    red_in = torch.ones(8, device=DEV)
    x = torch.randn(8, 2048, 3, device=DEV)
    y = torch.randn(8, 32768, 3, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(s_cham):
        for _ in range(400):
            ops.chamfer_forward_raw(x, y, both=False)
    for k in range(10):                                    # interleave the enqueues: 200 iterations per engine, then 400 collectives
        for r in runners:
            r.steps(ITERS // 10)
        with torch.cuda.stream(s_rccl):
            for _ in range(400):
                hip.check(L.psi_dp_allreduce_sum(comm, hip.ptr(red_in), 8, s_rccl.cuda_stream), 'psi_dp_allreduce_sum')
    got = []
    for op, r in zip(ops_, runners):
        x_e, _, step = op._fused.read(0)                   # synchronises the engine's stream and checks its error word (raises on 902)
        assert step == ITERS
        r.finish()
        got.append(fitting.GeometryTransformer.convert_to_3D_rot(op.xhr_rec).detach().cpu().numpy())
    torch.cuda.synchronize()
    assert float(red_in.sum()) == 8.0                      # world size 1: the sum over ranks is the value itself, 4000 times
    L.psi_dp_comm_destroy(comm)
    for a, g in zip(alone, got):
        assert np.array_equal(a, g)
