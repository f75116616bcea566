"""CPU checks of bench.py's host-side helpers (the GPU legs are exercised by the driver): argument contract, block statistics,
the algorithmic-work table behind `roofline`, and the rank launcher's command line."""
import os
import sys
import types

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_default_arguments_follow_the_driver_contract():
    a = bench.parse([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.workload == 'fitting'
    assert (a.batch, a.nc, a.m, a.D) == (32, 2048, 32768, 256)                # BASELINE configs[1]
    b = bench.parse(['--gpus', '4', '--steps', '20', '--warmup', '5'])
    assert (b.gpus, b.steps, b.warmup) == (4, 20, 5)


def test_block_statistics_report_the_median_block():
    s, med = bench.summarize([0.004, 0.002, 0.003, 0.010, 0.0025], K=20)
    assert med == 0.003
    assert s['ms_per_step'] == 0.15 and s['ms_per_step_min'] == 0.1 and s['ms_per_step_max'] == 0.5 and s['repeats'] == 5


def test_work_table_and_roofline_of_the_dominant_kernel():
    args = types.SimpleNamespace(batch=32, nc=2048, m=32768, D=256)
    work = bench.kernel_work(args)
    for k in ('head_fwd_kernel', 'blend_fwd_kernel', 'fwd_scene_kernel', 'bwd_joint_kernel', 'reduce_partials_kernel', 'head_bwd_adam_kernel'):   # the six launches of an iteration
        assert k in work and work[k][0] == 'byte' and work[k][1] > 0, k
    # SURVEY 8(d) bytes only, unpadded model dimensions: dirs [506, 31425] streamed once + g_vposed in + g_feat out, skinning weights
    # + g_local + v_posed in + joint-transform gradients out
    V, K = 10475, 506
    want = K * 3 * V * 4 + 32 * 3 * V * 4 + 32 * K * 4 + 55 * V * 4 + 2 * 32 * 3 * V * 4 + 32 * 55 * 16 * 4
    assert work['bwd_joint_kernel'][1] == want and abs(want - 78.2e6) < 0.2e6
    # the search lanes' weight-row re-reads are implementation traffic: the scene kernel counts 32 B per contact query, not 276; since round 6
    # the launch also does the per-vertex skinning backward: SURVEY 8(d)'s bytes of the launch it replaced are part of its algorithmic work
    fwd_part = 55 * V * 4 + 32 * 3 * V * 4 + 32 * V * 56 + 32 * 2048 * 32 + 32768 * 16
    bwd_part = 55 * V * 4 + 32 * V * 12 + 32 * 2048 * 12 + 2 * 32 * 3 * V * 4
    assert work['fwd_scene_kernel'][1] == fwd_part + bwd_part
    agg = {k: 0.02 for k in work if work[k][0] == 'byte'}                    # 20 us each
    agg['fwd_scene_kernel'] = 0.0264                                           # the LONGEST kernel is the dominant one
    roof, per = bench.roofline_from_kernels(args, agg, work)
    assert roof['kernel'] == 'fwd_scene_kernel' and roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and roof['unit'] == 'GB/s'
    # primary figures = the bytes the implementation has to MOVE (no [B,V,3] vertex store, no masked-gradient round trip, weights once, + the
    # contact slots' own rows); SURVEY 8(d)'s figure next to it
    moved = bench.moved_bytes(args, 'fwd_scene_kernel')
    assert moved == 55 * V * 4 + 32 * 3 * V * 4 + 32 * V * 32 + 2 * 32 * 3 * V * 4 + 32 * 2048 * 68 + 32768 * 16 and roof['bytes_per_launch'] == moved
    assert moved < work['fwd_scene_kernel'][1]
    assert abs(roof['achieved'] - moved / 26.4e-6 * 1e-9) < 5 and abs(roof['frac'] - roof['achieved'] / 8000.0) < 1e-3
    assert abs(roof['survey_8d_achieved'] - work['fwd_scene_kernel'][1] / 26.4e-6 * 1e-9) < 5 and roof['survey_8d_frac'] > roof['frac']
    assert per['fwd_scene_kernel']['frac_hbm'] == roof['frac'] and per['fwd_scene_kernel']['frac_hbm_survey_8d'] == roof['survey_8d_frac']
    assert abs(roof['share_of_iteration_time'] - 0.0264 / sum(agg.values())) < 1e-3
    assert set(per) == set(agg)


def test_rank_launcher_command_line(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment re-executes itself under torch.distributed.run on 127.0.0.1."""
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(bench.subprocess, 'call', fake_call)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '20', '--warmup', '5'])
    rc = bench.spawn_ranks(bench.parse(['--gpus', '4', '--steps', '20', '--warmup', '5']))
    cmd = ' '.join(seen['cmd'])
    assert rc == 0 and 'torch.distributed.run' in cmd and '--nproc-per-node 4' in cmd.replace('=', ' ') and '127.0.0.1' in cmd
    assert cmd.rstrip().endswith('--gpus 4 --steps 20 --warmup 5')


def test_rank_launch_command_for_an_eight_gpu_node():
    """The command line and environment with which `bench.py --gpus 8` starts its ranks (configs[3]: 8 x 32 bodies): eight processes on
    one node under torch.distributed.run, loopback rendezvous, the caller's arguments passed through unchanged, dmabuf IPC for RCCL."""
    argv = ['--gpus', '8', '--steps', '20', '--warmup', '5']
    cmd, env = bench.rank_launch_command(8, argv, port=29555, base_env={'PATH': '/usr/bin'})
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29555'
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == argv                                             # bench.py then joins the job: RANK / WORLD_SIZE from the env
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env['PSI_BENCH_SPAWNED'] == '1' and int(env['OMP_NUM_THREADS']) >= 1
    assert env['PATH'] == '/usr/bin'
