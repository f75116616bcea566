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
    for k in ('head_fwd_kernel', 'blend_fwd_kernel', 'fwd_scene_kernel', 'skin_bwd_v_grad_kernel', 'bwd_joint_kernel',
              'reduce_partials_kernel', 'head_bwd_adam_kernel'):
        assert k in work and work[k][0] == 'byte' and work[k][1] > 0, k
    assert abs(work['bwd_joint_kernel'][1] - 79.1e6) < 0.2e6
    agg = {k: 0.02 for k in work if work[k][0] == 'byte'}                    # 20 us each
    agg['bwd_joint_kernel'] = 0.0264
    roof, per = bench.roofline_from_kernels(args, agg, work)
    assert roof['kernel'] == 'bwd_joint_kernel' and roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and roof['unit'] == 'GB/s'
    assert abs(roof['achieved'] - 79.08e6 / 26.4e-6 * 1e-9) < 5 and abs(roof['frac'] - roof['achieved'] / 8000.0) < 1e-3
    assert set(per) == set(agg)
    if os.path.exists(os.path.join(ROOT, 'profiles', 'r02_kernel_stats.csv')):
        assert 0.2 < roof['rocprofv3_frac'] < 1.0 and roof['rocprofv3_avg_launch_ms'] > 0


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
