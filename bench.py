#!/usr/bin/env python
"""bench.py — fitting iterations/sec of the PSI hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE are read from the environment) or, when no
torchrun environment is present, bench.py starts them ITSELF by re-executing under torch.distributed.run on 127.0.0.1.

A "step" is ONE fitting iteration (fitting_proxe.py:177-189: zero_grad, cal_loss, backward, Adam step) over a batch of
32 bodies per GPU on synthetic PROX-E-shaped inputs (configs[1] of BASELINE.json: V=10475 SMPL-X-shaped model,
VPoser(512,32,[1,21,3]), n_c=2048 contact vertices, m=32768 scene points, 256^3 SDF; all inputs resident in HBM before the
timed region).  Weak scaling: every rank fits its own 32 bodies; the only data-path collective is the one 6-float
all-reduce per iteration of the loss normalisers (psi-release_amd/dist.py).

Protocol (the metric's own: BASELINE configs[1] is quoted on the reference's 100-iteration loop): W untimed warm-up
steps plus an untimed clock-ramp warm-up (>= 0.3 s of iterations), then timed blocks of EXACTLY K steps each, every block bracketed by a
barrier + torch.cuda.synchronize() on both sides and timed as the MAX over ranks.  Consecutive blocks form 100-ITERATION FRESH-START LOOPS
(fitting_proxe.py:177-189: a new batch starts from the initial body vector with a fresh Adam — an untimed psi_fit_set_problem(reset)
before the first block of every loop); at least 5 loops and 0.5 s of timed GPU time.  `ms_per_step` / `value` = the MEDIAN LOOP's time
per iteration (loop total / 100; min / max next to it); `steady_state` carries the median block of iterations 20 .. 100 of the loops
(the number earlier rounds reported), `loop_as_one_call` the same loop issued as ONE psi_fit_iterate call.

Workloads:  --workload fitting (default, configs[1]; configs[3] at --gpus 8) | fitting_habitat (configs[4]: contact
constant 1.0, Habitat camera flip, 64 bodies per GPU, a sweep over 7 synthetic rooms) | train_s2 (configs[2]).
The default line also carries `secondary.train_s2` (a bounded train_s2 measurement, N=1 only) so configs[2] is measured
by the same run, and `secondary.fitting_smplx_sparse_weights`: the headline workload on a body model whose skinning rows have the
released SMPL-X model's 4 non-zeros (the headline keeps the dense random [V, 55] weight matrix of the earlier rounds).

The JSON line also carries
  roofline     — the dominant (longest) kernel of the iteration: the bytes the implementation has to MOVE per launch (DESIGN.md section 3;
                 SURVEY 8(d)'s per-unit figure next to it as survey_8d_*) / its average launch duration measured with HIP events on the
                 launch stream; peak = 8 TB/s HBM3E (MI355X guide); traffic = the PMC bytes of profiles/r06_pmc_traffic.json;
  cpu_baseline — the oracle (oracle/psi_oracle.py + oracle/chamfer_oracle.c, a CPU port of the reference path,
                 validated against the reference's golden vectors) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
ROOMS = 7                     # fitting_habitat.py:238-241: seven MP3D-R rooms
LOOP_ITERS = 100              # fitting_proxe.py / fitting_habitat.py: num_iter of the shipped configuration (BASELINE configs[1]: '100-iter loop')
PMC_FILES = ('r06_pmc_traffic.json', 'r05_pmc_traffic.json')


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='bodies per GPU (fitting: 32 = BASELINE; fitting_habitat: 64)')
    ap.add_argument('--m', type=int, default=32768, help='scene points')
    ap.add_argument('--nc', type=int, default=2048, help='contact vertices')
    ap.add_argument('--D', type=int, default=256, help='SDF grid dimension')
    ap.add_argument('--weight-nnz', type=int, default=0,
                    help='fitting: non-zero skinning weights per vertex of the synthetic body model (0 = dense random rows, the headline since round 1; '
                         '4 = the sparsity of the released SMPL-X model: compressed-row skinning kernels)')
    ap.add_argument('--engine', default=os.environ.get('PSI_ENGINE', 'auto'), choices=['auto', 'fused', 'modular'])
    ap.add_argument('--workload', default='fitting', choices=['fitting', 'fitting_habitat', 'train_s2'],
                    help="'fitting' = BASELINE metric (configs[1]/[3]); 'fitting_habitat' = configs[4]; 'train_s2' = configs[2]")
    ap.add_argument('--repeats', type=int, default=5, help='minimum number of timed K-step blocks (the median is reported)')
    ap.add_argument('--min-timed-s', type=float, default=0.5, help='timed GPU time is kept above this by adding blocks')
    ap.add_argument('--graph', type=int, default=1, help='train_s2: replay the whole optimiser step as one HIP graph')
    ap.add_argument('--bf16', type=int, default=1, help='train_s2: bf16 autocast for the CVAE trunk')
    ap.add_argument('--secondary', type=int, default=1, help='attach the bounded train_s2 measurement (configs[2]) to the default line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='time budget of the CPU baseline sample')
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 64 if args.workload == 'fitting_habitat' else (128 if args.workload == 'train_s2' else 32)
    return args


# ----------------------------------------------------------------------------------------------------------------------
# multi-GPU launch
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def rank_launch_command(n_gpus, argv, port=None, base_env=None):
    """(command line, environment) with which `python bench.py --gpus N` starts its own N ranks: one process per GPU of ONE node under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), the caller's arguments passed through."""
    env = dict(os.environ if base_env is None else base_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(n_gpus, 1) // 2)))
    env['PSI_BENCH_SPAWNED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port() if port is None else port), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one per GPU, RCCL)."""
    cmd, env = rank_launch_command(args.gpus, sys.argv[1:])
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------------------------
# timing protocol
# ----------------------------------------------------------------------------------------------------------------------
def timed_blocks(run_steps, barrier, K, W, world, device, min_repeats=5, min_total_s=0.5, ramp_s=0.3, max_repeats=400, restart=None,
                 restart_every=1):
    """run_steps(n) enqueues n steps.  Returns the per-block wall times (max over ranks) of R blocks of exactly K steps.
    restart (optional) is called, untimed, in front of every `restart_every`-th timed block: it puts the problem back to its initial state
    (the metric is a fitting LOOP from generated bodies, fitting_proxe.py:177-189, not the converged regime of one problem iterated for
    ever); R is then a multiple of restart_every, so that the blocks form whole loops."""
    import torch
    run_steps(max(W, 0))
    barrier()
    # clock ramp: the first milliseconds after idle run at a lower clock (measured: 10 % on a 4 ms region)
    # (with more than one rank the steps contain a collective, so the ranks must agree on how many ramp rounds they run: rank 0's
    # clock decides, the decision is broadcast after every round)
    t0 = time.perf_counter()
    n_ramp = 0
    go = torch.ones(1, device=device, dtype=torch.int32)
    while n_ramp < 10000:
        run_steps(K)
        torch.cuda.synchronize()
        n_ramp += K
        more = time.perf_counter() - t0 < ramp_s
        if world > 1:
            go.fill_(1 if more else 0)
            torch.distributed.broadcast(go, src=0)
            more = bool(go.item())
        if not more:
            break
    est = (time.perf_counter() - t0) / max(n_ramp, 1)       # seconds per step, this rank
    R = max(min_repeats, int(math.ceil(min_total_s / max(est * K, 1e-9))))
    R = min(R, max_repeats)
    if restart is not None and restart_every > 1:
        R = int(math.ceil(R / restart_every)) * restart_every
    if world > 1:                                            # every rank must run the same number of blocks
        t = torch.tensor([R], device=device, dtype=torch.int64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        R = int(t.item())
    times = []
    for i in range(R):
        if restart is not None and i % restart_every == 0:
            restart()
        barrier()
        t0 = time.perf_counter()
        run_steps(K)
        barrier()
        times.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(times, device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        times = [float(x) for x in t.tolist()]
    return times


def fresh_start_blocks(runner, K, blocks=21):
    """The configs[1] workload as the reference runs it: a K-iteration loop from a FRESH start (psi_fit_set_problem with reset: the
    generated bodies as initial parameters, zeroed Adam state, no NN warm-start hints, an unconverged penetration mask) — the headline
    protocol above keeps iterating one problem, i.e. mostly times the converged regime.  Every block = set_problem (untimed) +
    synchronize, then K iterations timed to the next synchronize; the median block is reported."""
    import torch
    eng = runner.eng
    xhr, x0, cam = eng._args
    ts = []
    for _ in range(blocks):
        eng.set_problem(xhr, xhr, cam, reset=True)
        eng.stream.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.iterate(K, True)
        eng.stream.synchronize()
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {'what': '%d-iteration loops from a fresh start (reset Adam state, cold NN hints), median of %d' % (K, blocks),
            'ms_per_step': round(med / K * 1e3, 4), 'ms_per_step_min': round(min(ts) / K * 1e3, 4), 'ms_per_step_max': round(max(ts) / K * 1e3, 4),
            'iters_per_s': round(K / med, 1)}


def summarize(times, K):
    med = statistics.median(times)
    return {'ms_per_step': round(med / K * 1e3, 4), 'ms_per_step_min': round(min(times) / K * 1e3, 4),
            'ms_per_step_max': round(max(times) / K * 1e3, 4), 'repeats': len(times), 'timed_s': round(sum(times), 4)}, med


# ----------------------------------------------------------------------------------------------------------------------
# fitting workloads
# ----------------------------------------------------------------------------------------------------------------------
LOSS = {'weight_loss_rec': 1, 'weight_loss_vposer': 0.01, 'weight_contact': 0.1, 'weight_collision': 0.5}


def make_op(args, rank, device, scene_seed=0, habitat=False, assets=None):
    from psi_release_amd import fitting, synth
    smplx, vposer = assets if assets else (synth.make_smplx(7, weight_nnz=getattr(args, 'weight_nnz', 0)), synth.make_vposer_state(3))
    scene = synth.make_scene(scene_seed, args.m, args.D, args.nc)
    cfg = {'scene_verts_path': None, 'scene_sdf_path': None, 'human_model_path': None, 'vposer_ckpt_path': None,
           'init_lr_h': 0.1, 'num_iter': 1, 'batch_size': args.batch, 'device': device,
           'contact_part': synth.CONTACT_PARTS, 'contact_id_folder': None, 'verbose': False,
           'smplx_data': smplx, 'vposer_state': vposer, 'scene': scene, 'engine': args.engine_resolved,
           'concurrent_engines': ROOMS if habitat else 1}           # the 7 rooms of the sweep are 7 engines on 7 streams
    op = (fitting.FittingOPHabitat if habitat else fitting.FittingOP)(cfg, dict(LOSS))
    bodies = synth.make_bodies(11 + rank + 1000 * scene_seed, args.batch)
    if habitat:
        bodies['cam_ext'] = synth.make_cam_ext(5 + scene_seed, args.batch)[:1].repeat(args.batch, 0)   # one camera per room view
    return op, bodies, (smplx, vposer, scene)


def time_chamfer_kernel(op, args, reps=20):
    """Average duration of one brute-force Chamfer NN launch (body->scene direction, the one PSI consumes) with HIP events on
    the stream the kernel is launched on."""
    import torch
    from psi_release_amd import hip
    B, n, m = args.batch, args.nc, args.m
    x = torch.randn(B, n, 3, device=op.device) * 0.5
    y = op._s_verts_expanded
    d = torch.zeros(B, n, device=op.device)
    i = torch.zeros(B, n, dtype=torch.int32, device=op.device)
    L = hip.lib()
    ws = torch.empty(L.psi_chamfer_workspace_bytes(B, n, m), dtype=torch.uint8, device=op.device)
    st = torch.cuda.current_stream()
    call = lambda: L.psi_chamfer_forward(x.data_ptr(), y.data_ptr(), B, n, m, d.data_ptr(), i.data_ptr(), None, None,
                                         ws.data_ptr(), st.cuda_stream)
    for _ in range(10):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        call()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def kernel_work(args):
    """ALGORITHMIC work per launch of the kernels of one fitting iteration, from SURVEY.md 8(d) only: model tensors at their real
    (unpadded) sizes read once, per-vertex streams, the 96 B / vertex of the SDF lookup, contact queries; no implementation traffic
    (split-contraction partials, padding, weight rows re-read by other workgroups).  -> ('flop'|'byte', amount, note)."""
    B, V, J, nc, m = args.batch, 10475, 55, args.nc, args.m
    K, N = 506, 3 * V                       # feat = [10 betas + 10 expression | 54 x 9 rotation entries]; columns = 3 V
    dirs = K * N * 4.0                      # shapedirs + posedirs as one [K, 3V] matrix: 63.6 MB
    nnz = getattr(args, 'weight_nnz', 0)
    W = J * V * 4.0 if not nnz else nnz * V * 5.0      # skinning weights: dense rows 2.3 MB, or nnz (weight, joint byte) pairs per vertex
    vt = N * 4.0                            # v_template
    w = {
        'nn_partial_kernel': ('flop', 8.0 * B * nc * m, 'brute-force NN: 8 flop per query/target pair, fp32 VALU (no FMA contraction)'),
        'kd_query_kernel': ('byte', B * nc * (12 + 12 + 8) + m * 16.0,
                            'exact NN search (grid ball query / tree walk, latency-bound): queries in, gradients + hints out, scene once'),
        'blend_fwd_kernel': ('byte', dirs + vt + B * N * 4.0 + B * K * 4.0, 'v_posed = v_t + feat @ dirs: dirs streamed once (two fp16 parts per entry: the same 4 bytes) + feat in + v_posed out'),
        'bwd_joint_kernel': ('byte', dirs + B * N * 4.0 + B * K * 4.0 + W + 2 * B * N * 4.0 + B * J * 16 * 4.0,
                             'blend_bwd (dirs streamed once + g_vposed in + g_feat out) and skin_bwd_A (weights + g_local + v_posed in, joint-transform '
                             'gradients out) as one heterogeneous grid'),
        'blend_bwd_kernel': ('byte', dirs + B * N * 4.0 + B * K * 4.0, 'g_feat = g_vposed @ dirs^T: dirs streamed once + g_vposed in + g_feat out'),
        'skin_fwd_kernel': ('byte', W + B * N * 4.0 + B * V * 12.0, 'weights + v_posed in, vertices out (660 FMA per vertex: VALU-heavy)'),
        'skin_fwd_sdf_kernel': ('byte', W + B * N * 4.0 + B * V * (12.0 + 32 + 12),
                                'skinning + SDF lookup fused, SURVEY 8(d) accounting: weights + v_posed in, vertices out, 8 corner values + 12 B masked '
                                'gradient per vertex (round 4 stores only the contact rows of the vertices: `moved_bytes` = what this '
                                'implementation has to move)'),
        'fwd_scene_kernel': ('byte', W + B * N * 4.0 + B * V * (12.0 + 32 + 12) + B * nc * (12 + 12 + 8.0) + m * 16.0
                             + (W + B * V * 12.0 + B * nc * 12.0 + 2 * B * N * 4.0),
                             'ONE launch for both scene terms AND (round 6) the per-vertex skinning backward: skinning + SDF lookup (weights + v_posed in, '
                             'vertices out, 8 gathers + 12 B masked gradient per vertex), the exact NN search of the contact vertices (posed contact vertex in, '
                             'gradient + winner out, scene cloud once), and what SURVEY 8(d) counts for the backward launch it replaces (weights + SDF gradient + '
                             'contact gradients in, g_local + g_vposed out); the weight rows the search lanes re-read to skin their own query are '
                             'implementation traffic and not counted'),
        'skin_bwd_v_kernel': ('byte', W + B * V * 12.0 + 2 * B * N * 4.0, 'weights + grad in, g_local + g_vposed out'),
        'skin_bwd_v_grad_kernel': ('byte', W + B * V * 12.0 + B * nc * 12.0 + 2 * B * N * 4.0,
                                   'loss-gradient assembly + skinning backward fused: weights + SDF gradient + contact gradients in, g_local + g_vposed out'),
        'head_fwd_kernel': ('byte', 1.33e6 + B * 12000.0, 'VPoser decoder weights + per-body state + LBS pose stage'),
        'head_bwd_adam_kernel': ('byte', 1.33e6 + B * 20000.0, 'LBS pose backward + VPoser decoder backward + Adam: a per-body latency chain'),
        'reduce_partials_kernel': ('byte', B * (J * 16 + K) * 4.0, 'sums of the split-contraction partials: pure implementation traffic; only the outputs are algorithmic'),
    }
    return w


def moved_bytes(args, kernel):
    """Bytes the CURRENT implementation of `kernel` has to move for one launch where that is less than the SURVEY 8(d) figure of
    kernel_work (an output of the reference's formulation that is no longer materialised), else None.  Quoted next to the roofline so
    that a fraction computed from 8(d) bytes cannot be mistaken for traffic."""
    B, V, J, nc, m = args.batch, 10475, 55, args.nc, args.m
    N = 3 * V
    nnz = getattr(args, 'weight_nnz', 0)
    W = J * V * 4.0 if not nnz else nnz * V * 5.0
    if kernel == 'skin_fwd_sdf_kernel':          # vertices: only the n_c contact rows are stored (the search reads them)
        return W + B * N * 4.0 + B * V * (32 + 12.0) + B * nc * 12.0
    if kernel == 'fwd_scene_kernel':
        # shared launch, round 6: the search lanes skin their own queries (no vertex is stored), the masked SDF gradient never leaves the lane
        # (the vertex's backward happens on the spot: g_local + g_vposed out), the weights are read once, the contact slots write their own rows
        # (g_local, g_vposed, posed vertex: 36 B) next to the 32 B of the query itself
        return W + B * N * 4.0 + B * V * 32.0 + 2 * B * N * 4.0 + B * nc * (12 + 12 + 8.0 + 36.0) + m * 16.0
    if kernel == 'bwd_joint_kernel':
        # + the contact class of rows (round 6): the contact vertices' blend-shape columns gathered once per engine (3 n_c columns of the 506-row
        # matrix, streamed like the matrix itself), their skinning-weight rows and the slot-ordered gradient rows
        K = 506
        return (K * N * 4.0 + B * N * 4.0 + B * K * 4.0 + W + 2 * B * N * 4.0 + B * J * 16 * 4.0) + K * 3 * nc * 4.0 + J * nc * 4.0 + 3 * B * nc * 12.0
    return None


def load_pmc():
    for f in PMC_FILES:
        p = os.path.join(ROOT, 'profiles', f)
        if os.path.exists(p):
            return json.load(open(p)), 'profiles/' + f
    return None, None


def load_rocprof_stats(kernel):
    """Average duration [us] of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this workload (profiles/), or None.
    HIP-event deltas of single launches (the live measurement below) contain the launch gap — about 3 us on this stack — that the
    profiler's begin/end timestamps exclude; the committed summary is quoted next to the live figure so the two can be compared."""
    import csv
    for f in ('r06_kernel_stats.csv', 'r05_kernel_stats.csv'):
        p = os.path.join(ROOT, 'profiles', f)
        if not os.path.exists(p):
            continue
        try:
            for row in csv.DictReader(open(p)):
                if kernel in row['Name']:
                    return float(row['AverageNs']) * 1e-3, 'profiles/' + f
        except Exception:
            return None, None
    return None, None


def roofline_from_kernels(args, agg, work):
    """Roofline entry of the DOMINANT kernel of the iteration — the one with the largest share of the iteration's TIME — and the
    achieved bandwidth of every kernel (algorithmic bytes of kernel_work / HIP-event launch duration / 8 TB/s)."""
    per = {}
    for k, ms in agg.items():
        w = work.get(k)
        if w is None or w[0] != 'byte':
            continue
        gbs = w[1] / (ms * 1e-3) * 1e-9
        per[k] = {'us': round(ms * 1e3, 2), 'GB/s': round(gbs, 1), 'frac_hbm': round(gbs / PEAK_HBM_GBS, 4)}
        mb = moved_bytes(args, k)
        if mb is not None:                                  # primary = moved bytes; the SURVEY 8(d) figure stays next to it
            per[k]['frac_hbm_survey_8d'] = per[k]['frac_hbm']
            per[k]['GB/s'] = round(mb / (ms * 1e-3) * 1e-9, 1)
            per[k]['frac_hbm'] = round(mb / (ms * 1e-3) * 1e-9 / PEAK_HBM_GBS, 4)
    dom = max(agg, key=agg.get)
    w = work.get(dom)
    roof = None
    if w is not None:
        t_dom = agg[dom] * 1e-3
        if w[0] == 'flop':
            ach = w[1] / t_dom * 1e-12
            roof = {'bound': 'valu (fp32 vector)', 'kernel': dom, 'achieved': round(ach, 2), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_FP32_TFLOPS, 4), 'traffic': None, 'avg_launch_ms': round(agg[dom], 4),
                    'flops_per_launch': w[1], 'note': w[2]}
        else:
            ach = w[1] / t_dom * 1e-9
            roof = {'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': None, 'avg_launch_ms': round(agg[dom], 4),
                    'bytes_per_launch': w[1], 'note': w[2]}
        mb = moved_bytes(args, dom)
        if mb is not None and w[0] == 'byte':
            # SURVEY 8(d) counts the [B,V,3] vertices as an output of this kernel; this implementation no longer stores them (the search
            # lanes skin their own queries).  The PRIMARY figures are the bytes the kernel really has to move; the 8(d) figure is kept
            # next to them, labelled.
            roof['survey_8d_bytes_per_launch'] = w[1]
            roof['survey_8d_achieved'] = roof['achieved']
            roof['survey_8d_frac'] = roof['frac']
            roof['bytes_per_launch'] = mb
            roof['achieved'] = round(mb / t_dom * 1e-9, 1)
            roof['frac'] = round(mb / t_dom * 1e-9 / PEAK_HBM_GBS, 4)
            roof['bytes_note'] = ('`achieved` / `frac` use the bytes this implementation has to move per launch (fwd_scene: no [B,V,3] vertex store, no '
                                  'masked-gradient round trip, weights read once; bwd_joint: + the contact class of rows); survey_8d_* = the same launch '
                                  'time against SURVEY 8(d) bytes of the work the launch does')
        roof['share_of_iteration_time'] = round(agg[dom] / max(sum(agg.values()), 1e-12), 3)
        roof['share_of_iteration_bytes'] = round(w[1] / (131.7e6 + args.batch * 1.76e6), 3) if w[0] == 'byte' else None
        if (args.batch, args.nc, args.m, args.D) == (32, 2048, 32768, 256):
            us, srcp = load_rocprof_stats(dom)
            if us:
                roof['rocprofv3_avg_launch_ms'] = round(us * 1e-3, 4)
                amount = roof.get('bytes_per_launch', w[1]) if w[0] == 'byte' else w[1]
                roof['rocprofv3_frac'] = round(amount / (us * 1e-6) / (PEAK_FP32_TFLOPS * 1e12 if w[0] == 'flop' else PEAK_HBM_GBS * 1e9), 4)
                roof['rocprofv3_source'] = srcp + ' (committed summary of the same command; HIP-event deltas of single launches include the ~3 us launch gap)'
        pmc, src = load_pmc()
        if pmc and (args.batch, args.nc, args.m, args.D) == (32, 2048, 32768, 256) and dom in pmc:
            roof['traffic'] = pmc[dom]['bytes']
            roof['traffic_source'] = '%s (FETCH_SIZE%s + WRITE_SIZE, separate rocprofv3 --pmc passes)' % (
                src, ' x2 (gfx950 wide-load correction)' if pmc[dom].get('fetch_x2') else '')
    return roof, per


def cpu_baseline(args, assets, budget_s, habitat=False):
    """Oracle fitting iterations on the host cores: bounded sample of the SAME workload (same B, m, n_c, D)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import psi_oracle as O
    from psi_release_amd import synth
    smplx, vposer, scene = assets
    ncpu = os.cpu_count() or 1
    kw = {'contact_const': 1.0} if habitat else {}
    fo = O.FittingOracle(O.SMPLXOracle(smplx), vposer, scene.verts, scene.sdf, scene.grid_min, scene.grid_max,
                         synth.contact_ids_from_parts(scene.contact_parts), args.batch, **kw)
    bodies = synth.make_bodies(11, args.batch)
    xh = synth.body_vector_72(bodies)
    # the oracle is memory/NUMA sensitive: all cores of a 256-thread host are ~200x SLOWER than 16 threads.  Report the
    # best of a small sweep so the GPU/CPU ratio is not inflated by a badly configured baseline.
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    best_c, best_t = cands[0], float('inf')
    for c in cands:
        O.set_threads(c)
        fo.fitting(xh, bodies['cam_ext'], 1)                # untimed warm-up at this thread count
        t0 = time.time()
        fo.fitting(xh, bodies['cam_ext'], 1)
        dt1 = time.time() - t0
        if dt1 < best_t:
            best_c, best_t = c, dt1
        if dt1 > 4 * best_t:                                # clearly past the knee: skip larger counts
            break
    cores = best_c
    O.set_threads(cores)
    n, t0 = 0, time.time()
    while True:
        fo.fitting(xh, bodies['cam_ext'], 1)
        n += 1
        el = time.time() - t0
        if (el >= budget_s and n >= 2) or n >= 2000:
            break
    res = {'value': round(n / el, 4), 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
           'sample': '%d fitting iterations (B=%d, n_c=%d, m=%d, D=%d) of oracle/psi_oracle.py FittingOracle '
                     '(torch-CPU fp32 + C/OpenMP AVX Chamfer restatement) in %.1f s at the best of %s threads on a %d-thread host'
                     % (n, args.batch, args.nc, args.m, args.D, el, cands, ncpu)}
    # second Chamfer variant of BASELINE.md section 2: the reference's own pure-PyTorch formulation (chamfer_python.py:4-9,18-28:
    # expanded form through matrix products + min), what a CPU run of the reference without the CUDA extension would execute
    try:
        fo2 = O.FittingOracle(O.SMPLXOracle(smplx), vposer, scene.verts, scene.sdf, scene.grid_min, scene.grid_max,
                              synth.contact_ids_from_parts(scene.contact_parts), args.batch, chamfer='expanded', **kw)
        best2 = None
        for c in sorted({cores, min(ncpu, 64)}):
            O.set_threads(c)
            n2, t0 = 0, time.time()
            while True:
                fo2.fitting(xh, bodies['cam_ext'], 1)
                n2 += 1
                el2 = time.time() - t0
                if el2 >= budget_s / 3 or n2 >= 50:
                    break
            if best2 is None or n2 / el2 > best2[0]:
                best2 = (n2 / el2, c, n2, el2)
        res['variant_torch_expanded_form_chamfer'] = {
            'value': round(best2[0], 4), 'unit': 'iters/s', 'cores': best2[1],
            'sample': '%d iterations in %.1f s; Chamfer as chamfer_pytorch/chamfer_python.py:4-9 (|x|^2 + |y|^2 - 2 x.y^T via torch.mm, min)' % (best2[2], best2[3])}
    except Exception as e:
        res['variant_torch_expanded_form_chamfer'] = {'error': repr(e)}
    O.set_threads(cores)
    return res


def bench_fitting(args):
    import torch
    from psi_release_amd import dist as pd
    habitat = args.workload == 'fitting_habitat'
    # RCCL (backend 'nccl') over xGMI; PSI_DIST_BACKEND=gloo lets a single-GPU box exercise the N>1 code path
    backend = os.environ.get('PSI_DIST_BACKEND', 'nccl')
    rank, local_rank, world = pd.init_from_env(backend)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path is the only implementation')
    ndev = torch.cuda.device_count()
    if world > ndev and backend == 'nccl':
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU over RCCL)' % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    from psi_release_amd import fitting
    args.engine_resolved = ('fused' if getattr(fitting, 'HAS_FUSED_ENGINE', False) else 'modular') if args.engine == 'auto' else args.engine

    if habitat:
        ops, runners = [], []
        assets0 = None
        for r in range(ROOMS):
            op, bodies, assets = make_op(args, rank, device, scene_seed=r, habitat=True, assets=assets0[:2] if assets0 else None)
            assets0 = assets0 or assets
            ops.append(op)
            runners.append(op.make_step_runner(bodies))
        op, runner, assets = ops[0], runners[0], assets0

        def run_steps(n):                                     # K steps of the sweep: the rooms take turns, K/7 iterations each
            base, extra = divmod(n, ROOMS)
            for r in range(ROOMS):
                k = base + (1 if r < extra else 0)
                if k:
                    runners[r].steps(k)
    else:
        op, bodies, assets = make_op(args, rank, device)
        runner = op.make_step_runner(bodies)                  # everything resident in HBM from here on
        runners = [runner]
        # steps(n) = n fitting iterations, exactly what FittingOP.fitting runs for num_iter = n (single process: device-resident
        # graphs of 10 iterations + single-iteration graphs for the remainder; data parallel: forward / all-reduce / backward)
        run_steps = runner.steps

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The metric is a fitting LOOP from generated bodies (fitting_proxe.py:177-189: 100 iterations per file), so every timed block starts
    # where that loop starts: psi_fit_set_problem(reset) — the generated bodies as initial parameters, zeroed Adam state, no NN warm-start
    # hints, an unconverged penetration mask — issued UNTIMED in front of the block's barrier.  `steady_state` (extra key) keeps iterating
    # one problem, i.e. times the converged regime (rounds 1-4 reported that as the headline).
    can_restart = args.engine_resolved == 'fused' and all(hasattr(r, 'restart') for r in runners)
    restart = (lambda: [r.restart() for r in runners]) if can_restart else None
    # blocks of exactly K steps; LOOP_ITERS / K consecutive blocks (rounded up) form one loop of the reference and share one restart
    bpl = max(1, int(math.ceil(LOOP_ITERS / args.steps))) if restart else 1
    times = timed_blocks(run_steps, barrier, args.steps, args.warmup, world, device, max(args.repeats, bpl), args.min_timed_s, restart=restart,
                         restart_every=bpl)
    summ, med = summarize(times, args.steps)
    if restart and bpl > 1:
        # headline = the median LOOP: its blocks' times added up / its iterations (a median over single blocks would pick a warm block and
        # hide the loop's cold first one)
        loops = [sum(times[i:i + bpl]) for i in range(0, len(times), bpl)]
        lmed = statistics.median(loops)
        summ = dict(summ, ms_per_step=round(lmed / (bpl * args.steps) * 1e3, 4), ms_per_step_min=round(min(loops) / (bpl * args.steps) * 1e3, 4),
                    ms_per_step_max=round(max(loops) / (bpl * args.steps) * 1e3, 4), loops=len(loops), blocks_per_loop=bpl,
                    per_block_ms_per_step={'first_block_of_a_loop': round(statistics.median(times[0::bpl]) / args.steps * 1e3, 4),
                                           'later_blocks': round(statistics.median([t for i, t in enumerate(times) if i % bpl]) / args.steps * 1e3, 4)})
        med = lmed / bpl                                      # seconds per block of K steps, loop average
    steady = None
    if restart is not None:
        ts = timed_blocks(run_steps, barrier, args.steps, 0, world, device, 5, 0.2, ramp_s=0.05)
        steady, _ = summarize(ts, args.steps)
        steady['what'] = 'the same blocks WITHOUT the restart: one problem iterated on and on (converged regime, warm NN hints)'
    fresh = None
    if world == 1 and not habitat and args.engine_resolved == 'fused':
        fresh = fresh_start_blocks(runner, LOOP_ITERS)     # the same loop as ONE call of 100 iterations (no per-block synchronisation)
    losses = runners[-1].last_losses() if habitat else runner.last_losses()
    rccl_world = torch.distributed.get_world_size() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
    per_rank_ms = None
    if world > 1:                                             # per-rank local view of the median block (no max-reduction)
        t = torch.zeros(world, device=device, dtype=torch.float64)
        t[rank] = med / args.steps * 1e3
        torch.distributed.all_reduce(t)
        per_rank_ms = [round(float(x), 4) for x in t.tolist()]
    # what RCCL ITSELF reports on every rank (ncclCommCount / ncclGetVersion through psi_dp_comm_info, all-gathered) and how each rank's
    # loop was launched (psi_fit_dp_mode): a line produced on N GPUs shows N ranks that each saw N ranks, all on graphs — or says what else
    # happened.  Single process: one entry, RCCL not loaded.
    rccl_seen, rccl_version, dp_modes = [1], None, ['single process']
    uses_rccl = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl' \
        and args.engine_resolved == 'fused' and (world > 1 or os.environ.get('PSI_FORCE_DP_PATH') == '1')
    if uses_rccl:
        from psi_release_amd import dist as _pd
        _, seen, ver = _pd.rccl_comm_info()
        eng = (runners[0].eng if hasattr(runners[0], 'eng') else None)
        mode = eng.dp_mode() if eng is not None else 0
        t = torch.zeros(world, 3, device=device, dtype=torch.int64)
        t[rank, 0], t[rank, 1], t[rank, 2] = seen, ver, mode
        if world > 1:
            torch.distributed.all_reduce(t)
        rccl_seen = [int(x) for x in t[:, 0].tolist()]
        rccl_version = int(t[0, 1])
        dp_modes = [{0: 'python loop', 1: 'hipGraph (collective captured)', 2: 'eager from C (capture refused)'}.get(int(x), str(int(x))) for x in t[:, 2].tolist()]
    elif world > 1:
        dp_modes = ['python loop (%s collective)' % backend] * world
        rccl_seen = [0] * world

    out = None
    if rank == 0:
        if habitat:
            metric = 'fitting iters/sec (SMPL-X+SDF+Chamfer), MP3D-R 7-room sweep batch=%d per GPU' % args.batch
            wl = ('fitting_habitat sweep over %d synthetic rooms (contact constant 1.0, Habitat camera flip): one Adam fitting iteration per step, '
                  'batch=%d bodies per GPU, V=10475, n_c=%d, m=%d, SDF %d^3 per room (BASELINE configs[4])' % (ROOMS, args.batch, args.nc, args.m, args.D))
        else:
            metric = 'fitting iters/sec (SMPL-X+SDF+Chamfer), PROX-E batch=32'
            wl = ('fitting_proxe 100-iter loop: one Adam fitting iteration per step, batch=%d bodies per GPU, V=10475, n_c=%d, m=%d, SDF %d^3, '
                  'synthetic SMPL-X/VPoser/scene (BASELINE configs[%d])' % (args.batch, args.nc, args.m, args.D, 3 if world >= 8 else 1))
        out = {
            'metric': metric, 'value': round(world / (med / args.steps), 3), 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'dtype_note': 'fp32 storage and arithmetic; the two blend-shape products (v_posed = v_t + feat.dirs and its backward, B <= 128) run on the fp16 matrix pipe as three split products of two fp16 parts per operand (22 mantissa bits, 2^-22 per product): held to the fp32 accuracy class against fp64 by tests/test_lbs_gpu.py and tests/test_fitting_gpu.py (*_accuracy_class)', 'data': 'synthetic',
            'config': {'workload': wl, 'skinning_weight_nnz': args.weight_nnz or 'dense', 'per_gpu_batch': args.batch, 'global_batch': args.batch * world, 'engine': args.engine_resolved,
                       'nn': op.nn_mode, 'parallelism': 'dp%d (rows sharded, one 6-float all-reduce per iteration)' % world,
                       'rccl_world_size': rccl_world, 'rccl_ranks_seen': rccl_seen, 'rccl_version': rccl_version, 'dp_launch_mode': dp_modes,
                       'backend': backend if world > 1 else 'none (single process)',
                       'launcher': 'bench.py self-spawn' if os.environ.get('PSI_BENCH_SPAWNED') == '1' else ('torchrun' if world > 1 else 'direct'),
                       'protocol': '%d blocks of %d steps (barrier + synchronize around each, max over ranks)%s' % (
                           summ['repeats'], args.steps, ('; %d consecutive blocks = ONE %d-iteration fitting loop from a FRESH problem (untimed '
                           'psi_fit_set_problem with reset in front of its first block: generated bodies, zeroed Adam state, cold NN hints); '
                           'ms_per_step = median loop / its iterations' % (bpl, bpl * args.steps)) if restart else '; median block'),
                       'final_losses': [round(float(x), 6) for x in losses]},
        }
        out.update(summ)
        if steady:
            out['steady_state'] = steady
        if fresh:
            out['loop_as_one_call'] = fresh
        if per_rank_ms:
            out['per_rank_ms_per_step'] = per_rank_ms
        # ---- per-kernel times of one iteration: HIP events recorded on the launch stream after every kernel launch
        work = kernel_work(args)
        roof = None
        if args.engine_resolved == 'fused' and world == 1:
            kernels = runner.eng.profile(30)
            agg = {}
            for nm, ms in kernels:
                agg[nm] = agg.get(nm, 0.0) + ms
            out['kernels_us'] = {k: round(v * 1e3, 2) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}
            roof, per = roofline_from_kernels(args, agg, work)
            out['kernel_bandwidth'] = per
            alg = 131.7e6 + args.batch * 1.76e6                    # algorithmic bytes of the whole iteration
            out['iteration_roofline'] = {'bound': 'hbm', 'algorithmic_bytes': alg, 'achieved': round(alg / (med / args.steps) * 1e-9, 1),
                                         'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(alg / (med / args.steps) * 1e-9 / PEAK_HBM_GBS, 4)}
        if world == 1 and not habitat:
            t_ch = time_chamfer_kernel(op, args)
            flops = 8.0 * args.batch * args.nc * args.m
            out['roofline_bruteforce_nn'] = {
                'bound': 'valu (fp32 vector)', 'kernel': 'nn_partial_kernel + nn_resolve_kernel (brute-force Chamfer NN op, fp32 VALU)',
                'achieved': round(flops / t_ch * 1e-12, 2), 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(flops / t_ch * 1e-12 / PEAK_FP32_TFLOPS, 4), 'avg_launch_ms': round(t_ch * 1e3, 4), 'flops_per_launch': flops,
                'note': 'fp32 vector peak == fp32 MFMA peak on gfx950; 8 flop/pair without FMA contraction caps the fraction near 8/18'}
            if roof is None:
                roof = dict(out['roofline_bruteforce_nn'], traffic=None)
        if roof is not None:
            out['roofline'] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args, assets, args.cpu_seconds, habitat)
            except Exception as e:  # the oracle needs gcc; report rather than fail the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'iters/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
    # free the fitting state before the secondary measurement
    del runners, runner, op
    if habitat:
        del ops
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not habitat and args.secondary and (args.batch, args.nc, args.m, args.D) == (32, 2048, 32768, 256):
            out['secondary'] = {}
            try:
                # the same workload on a body model with the RELEASED SMPL-X model's skinning sparsity (4 non-zero weights per vertex,
                # tree-local): the headline line above keeps the dense random [V, J] weight matrix of rounds 1-2
                from psi_release_amd import synth
                sp_assets = (synth.make_smplx(7, weight_nnz=4), assets[1])
                op_s, bodies_s, _ = make_op(args, rank, device, assets=sp_assets)
                run_s = op_s.make_step_runner(bodies_s)
                ts = timed_blocks(run_s.steps, lambda: torch.cuda.synchronize(), args.steps, args.warmup, 1, device, min_repeats=5, min_total_s=0.2)
                run_s.finish()
                med = statistics.median(ts) / args.steps
                out['secondary']['fitting_smplx_sparse_weights'] = {
                    'metric': out['metric'], 'value': round(1.0 / med, 1), 'ms_per_step': round(med * 1e3, 4), 'blocks': len(ts),
                    'body_model': 'synthetic SMPL-X, 4 non-zero skinning weights per vertex (as the released model)'}
                del run_s, op_s
                torch.cuda.empty_cache()
            except Exception as e:
                out['secondary']['fitting_smplx_sparse_weights'] = {'error': repr(e)}
            # the skinning + SDF kernel at a batch size where it is throughput-bound (B = 512): the kernel the north star asks >= 40 % of the
            # HBM peak of, measured live (HIP events around the launch, psi_fit_profile) for the released model's 4-non-zero skinning rows
            # (compressed-row kernels: the realistic case) and for the dense random rows of the headline
            for key, nnz in (('skin_fwd_sdf_asymptote_sparse_rows', 4), ('skin_fwd_sdf_asymptote_dense_rows', 0)):
                try:
                    from psi_release_amd import synth
                    a512 = argparse.Namespace(**dict(vars(args), batch=512, weight_nnz=nnz))
                    as_assets = (synth.make_smplx(7, weight_nnz=nnz) if nnz else assets[0], assets[1])
                    op_a, bodies_a, _ = make_op(a512, rank, device, assets=as_assets)
                    run_a = op_a.make_step_runner(bodies_a)
                    run_a.steps(20)
                    torch.cuda.synchronize()
                    agg_a = {}
                    for nm, ms in run_a.eng.profile(20):
                        agg_a[nm] = agg_a.get(nm, 0.0) + ms
                    wk = kernel_work(a512)['skin_fwd_sdf_kernel']
                    t_k = agg_a['skin_fwd_sdf_kernel'] * 1e-3
                    mb = moved_bytes(a512, 'skin_fwd_sdf_kernel')
                    out['secondary'][key] = {
                        'kernel': 'psi_skin_fwd_kernel<SdfPenEpilogue> (skin_fwd_sdf_kernel)', 'batch': 512, 'skinning_weight_nnz': nnz or 'dense',
                        'bound': 'hbm', 'avg_launch_ms': round(agg_a['skin_fwd_sdf_kernel'], 4),
                        'achieved': round(mb / t_k * 1e-9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(mb / t_k * 1e-9 / PEAK_HBM_GBS, 4),
                        'bytes_per_launch': mb, 'survey_8d_bytes_per_launch': wk[1], 'survey_8d_frac': round(wk[1] / t_k * 1e-9 / PEAK_HBM_GBS, 4),
                        'iteration_ms': round(sum(agg_a.values()), 4),
                        'note': 'bytes this implementation moves (contact rows of the vertices only) / HIP-event launch time / 8 TB/s; survey_8d_* '
                                'counts all [B,V,3] vertices as an output (SURVEY 8(d))'}
                    run_a.finish()
                    del run_a, op_a, bodies_a
                    torch.cuda.empty_cache()
                except Exception as e:
                    out['secondary'][key] = {'error': repr(e)}
            try:
                import contextlib
                import io
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    sec = bench_train_s2(argparse.Namespace(**dict(vars(args), batch=128, steps=10, warmup=3, repeats=5, min_timed_s=0.3)))
                sys.stderr.write(buf.getvalue())
                out['secondary']['train_s2'] = sec
            except Exception as e:
                out['secondary']['train_s2'] = {'error': repr(e)}
            # the same step at the REFERENCE's precision (fp32 model, cvae.py:427-455): what `train_s1/s2.py` run out of the box (`--bf16 0` is
            # their default: a drop-in must not change a checkpoint's arithmetic silently) — since round 5 on hand-written kernels as well:
            # fp32 maps, three-term split products on the bf16 matrix cores with fp32 accumulation (conv_gemm.hip, linear.hip, bnorm.hip)
            try:
                import contextlib
                import io
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    sec = bench_train_s2(argparse.Namespace(**dict(vars(args), batch=128, steps=5, warmup=3, repeats=3, min_timed_s=0.2, bf16=0)))
                sys.stderr.write(buf.getvalue())
                out['secondary']['train_s2_fp32'] = sec
            except Exception as e:
                out['secondary']['train_s2_fp32'] = {'error': repr(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        pd.rccl_comm_release()
        torch.distributed.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------
# train_s2 (BASELINE configs[2])
# ----------------------------------------------------------------------------------------------------------------------
def bench_train_s2(args):
    """train_s2.py optimiser steps at batch 128 on synthetic PROX-shaped data (two scenes with 256^3 SDFs held once in HBM,
    indirect scene ids), CVAE trunk on the matrix cores.  Returns the result dict (same keys as the main line)."""
    import tempfile
    import numpy as np
    import torch
    from psi_release_amd import batch_gen, synth, training
    dev = torch.device('cuda', torch.cuda.current_device())
    B = args.batch
    names = ['SynA', 'SynB']
    sd = {n: synth.make_scene(i, args.m, args.D, args.nc) for i, n in enumerate(names)}
    scenes = {n: {'verts': s.verts, 'sdf': s.sdf, 'grid_min': s.grid_min, 'grid_max': s.grid_max, 'grid_dim': s.grid_dim} for n, s in sd.items()}
    rs = np.random.RandomState(0)
    n = B * 4
    body = synth.body_vector_72(synth.make_bodies(0, n))
    body[:, 2] = np.abs(body[:, 2]) + 2.0
    t = {'depth': rs.uniform(-1, 1, (n, 1, 128, 128)), 'seg': rs.uniform(-1, 1, (n, 1, 128, 128)), 'body': body,
         'cam_ext': synth.make_cam_ext(0, n), 'cam_int': synth.make_bodies(0, n)['cam_int'], 'max_d': np.full(n, 6.0),
         'sceneid': rs.randint(0, 2, n).astype(np.float32)}
    table = {k: np.concatenate([np.zeros_like(np.asarray(v)[:1]), np.asarray(v)]).astype(np.float32) for k, v in t.items()}
    bg = batch_gen.BatchGeneratorWithSceneMesh.from_arrays(table, scenes, dev, indirect_sdf=True)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = {'human_model_path': None, 'vposer_ckpt_path': None, 'scene_model_ckpt': None, 'init_lr_h': 1e-4, 'batch_size': B, 'epoch': 10,
               'loss_weight_anealing': True, 'device': dev, 'save_dir': tmp, 'contact_id_folder': None, 'contact_part': synth.CONTACT_PARTS,
               'verbose': False, 'use_cont_rot': True, 'resume_training': False, 'smplx_data': synth.make_smplx(7),
               'vposer_state': synth.make_vposer_state(3), 'contact_parts_data': sd['SynA'].contact_parts, 'autocast_bf16': bool(args.bf16), 'use_graph': bool(args.graph)}
        lw = {'weight_loss_rec_s': 1.0, 'weight_loss_rec_h': 1.0, 'weight_loss_vposer': 1e-3, 'weight_loss_kl': 1e-1, 'weight_contact': 1e-1,
              'weight_collision': 1e-1}
        op = training.TrainOPS2(cfg, lw)
        op.model_h.train()
        batches = []
        while bg.has_next_batch():
            d = bg.next_batch(B)
            if d is not None:
                batches.append(d)
        # matrix-core flops of one step (forward + backward of the PyTorch trunk; the HIP body/scene operators are not GEMMs)
        # (counted with the hand-written convolution / dense / BN ops switched off: the flop counter only sees aten operators, and the
        # arithmetic of the model is the same on either path)
        flops = None
        import psi_release_amd.models as _models
        routing = {k: getattr(_models, k) for k in ('_precise', '_use_hip_bn', '_use_hip_linear', '_conv')}
        try:
            from torch.utils.flop_counter import FlopCounterMode
            # (for the count only: every layer as the aten operator it is, so that the counter sees it — the product path below is untouched)
            _models._precise = lambda x: False
            _models._use_hip_bn = lambda bn, x: False
            _models._use_hip_linear = lambda m, x: False
            _models._conv = lambda conv, x: conv(x)
            running = {k: v.clone() for k, v in op.model_h.state_dict().items() if 'running_' in k or 'num_batches' in k}
            with FlopCounterMode(display=False) as fc:
                op.optimizer_h.zero_grad(set_to_none=True)
                sum(op._losses_from_batch(batches[0], 9)).backward()
            flops = float(fc.get_total_flops())
            op.optimizer_h.zero_grad(set_to_none=True)
            op.model_h.load_state_dict(running, strict=False)
        except Exception:
            pass
        finally:
            for k, v in routing.items():
                setattr(_models, k, v)
        cnt = [0]

        def run_steps(k):
            for _ in range(k):
                op.train_step(batches[cnt[0] % len(batches)], ep=9)       # ep > 0.75*epoch: scene losses active
                cnt[0] += 1

        times = timed_blocks(run_steps, torch.cuda.synchronize, args.steps, args.warmup, 1, dev, args.repeats, args.min_timed_s, ramp_s=0.2)
        summ, med = summarize(times, args.steps)
    spp = med / args.steps
    res = {'metric': 'train_s2 optimiser steps/sec (HumanCVAES2 + SMPL-X + Chamfer + SDF), batch=%d' % B,
           'value': round(1.0 / spp, 3), 'unit': 'steps/s', 'samples_per_s': round(B / spp, 1), 'n_gpus': 1,
           'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'bf16 trunk / f32 losses' if args.bf16 else 'f32 storage, bf16x3 split products (2^-16 per product), f32 accumulate', 'data': 'synthetic',
           'config': {'workload': 'train_s2.py step, batch=%d, 2 scenes (m=%d, SDF %d^3, indirect scene ids), n_c=%d (BASELINE configs[2])'
                                  % (B, args.m, args.D, args.nc), 'hip_graph': bool(args.graph)}}
    res.update(summ)
    if flops:
        # the fp32 model issues THREE bf16 products per useful multiply-add (hi*hi + hi*lo + lo*hi) on the bf16 matrix pipe: the fraction is
        # quoted against the pipe the step runs on (2.5 PFLOP/s bf16), with the issued flops; the useful-flop rate is next to it
        issued = flops * (1.0 if args.bf16 else 3.0)
        ach = issued / spp * 1e-12
        res['roofline'] = {'bound': 'mfma', 'kernel': 'whole optimiser step (CVAE trunk GEMMs/convs, forward + backward)', 'achieved': round(ach, 2),
                           'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_BF16_TFLOPS, 5), 'flops_per_step': issued,
                           'useful_flops_per_step': flops, 'useful_TFLOP/s': round(flops / spp * 1e-12, 2), 'traffic': None,
                           'note': 'matrix-core flops counted by torch.utils.flop_counter over one forward+backward (library path; the hand-written '
                                   'conv / dense kernels do the same arithmetic); a step is a chain of small (128-row) GEMMs/convs and elementwise '
                                   'passes, bound by launch count and activation traffic, not by the MFMA pipe',
                           'hand_written': ('bf16 mode: stride-1 3x3 convolutions forward + input gradient + weight gradient (conv.hip), every other '
                                            'convolution (7x7 stem, strided 3x3, 1x1 downsample, 128 -> 32 head) forward + both gradients on the general '
                                            'implicit-GEMM kernel with bf16 products (conv_gemm.hip), dense layers forward + backward (linear.hip), BatchNorm + '
                                            'ReLU + skip (bnorm.hip), stem max-pool' if args.bf16 else
                                            'fp32 mode (the reference\'s precision): EVERY convolution forward + input gradient + weight gradient on the general '
                                            'implicit-GEMM kernel with three-term split products (conv_gemm.hip), every dense layer forward + backward likewise '
                                            '(linear.hip: psi_linear_forward3 / _backward3), BatchNorm + ReLU + skip and max-pool on fp32 maps (bnorm.hip) — no '
                                            'MIOpen / hipBLASLt kernel is left in the step (profiles/r05_train_s2_fp32_kernel_stats.csv)') +
                                           '; the loss glue of cal_loss (cvae_loss.hip, scene_loss.hip), body decode / NN / SDF operators, Adam over all tensors in two launches '
                                           '(adam.hip); aten: gradient accumulation adds, layout copies of inputs / gradients, concatenations'}
    try:
        res['conv_kernel_roofline'] = conv_kernel_roofline(dev, B)
    except Exception as e:
        res['conv_kernel_roofline'] = {'error': repr(e)}
    return res


def conv_kernel_roofline(dev, N):
    """The matrix-core kernel that carries most of the trunk's flops — conv3x3_kernel (csrc/conv.hip) at the layer1 shape of the step
    (N x 32 x 32 x 64 -> 64, cvae.py:427-435) — timed alone with HIP events on its launch stream: achieved bf16 TFLOP/s against the dense
    MFMA peak."""
    import torch
    from psi_release_amd import hip
    H = W = 32
    C = 64
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
    L = hip.lib()
    st = torch.cuda.current_stream()
    call = lambda: hip.check(L.psi_conv3x3_forward(x.data_ptr(), w.data_ptr(), None, N, H, W, C, C, y.data_ptr(), st.cuda_stream), 'psi_conv3x3_forward')
    for _ in range(10):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record(st)
    for _ in range(reps):
        call()
    e1.record(st)
    e1.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    flops = 2.0 * N * H * W * C * C * 9
    return {'bound': 'mfma', 'kernel': 'conv3x3_kernel<64> (layer1: %d x 32 x 32 x 64 -> 64, bf16 MFMA 32x32x16, fp32 accumulate)' % N,
            'achieved': round(flops / t * 1e-12, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(flops / t * 1e-12 / PEAK_BF16_TFLOPS, 4),
            'avg_launch_ms': round(t * 1e3, 4), 'flops_per_launch': flops,
            'traffic': None, 'counters': 'profiles/r03_pmc_mfma_conv3x3_kernel.txt (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA per launch)'}


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.workload == 'train_s2':
        import contextlib
        import io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):                 # the trainer's [INFO] lines go to stderr: stdout carries ONE JSON line
            res = bench_train_s2(args)
        sys.stderr.write(buf.getvalue())
        print(json.dumps(res))
        return
    bench_fitting(args)


if __name__ == '__main__':
    main()
