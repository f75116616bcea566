"""Several ranks of a file-sharded run create the same fresh output directory at the same moment (`--shard files`, every rank of
`torchrun train_s2.py` on one `save_dir`): the reference's single-process idiom `if not exists: makedirs` (fitting_proxe.py:201-204)
loses that race.  Two processes are released together, 50 rounds, each round into directories that do not exist yet."""
import os
import subprocess
import sys
import time

from conftest import ROOT


def test_two_processes_write_into_fresh_directories(tmp_path):
    n = 50
    worker = os.path.join(ROOT, 'tests', 'race_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path), str(i), str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for i in range(2)]
    try:
        for r in range(n):
            t0 = time.time()
            while not all(os.path.exists(os.path.join(str(tmp_path), 'ready_%d_%d' % (r, i))) for i in range(2)):
                dead = [p for p in procs if p.poll() not in (None, 0)]
                if dead:
                    raise AssertionError('a worker lost the directory race in round %d:\n%s' % (r, dead[0].stderr.read()[-3000:]))
                assert time.time() - t0 < 120
                time.sleep(0.001)
            open(os.path.join(str(tmp_path), 'go_%d' % r), 'w').close()
        for p in procs:
            out, err = p.communicate(timeout=120)
            assert p.returncode == 0, err[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r in range(n):
        for sub in ('fit_%d' % r, 'gen_%d' % r):
            assert sorted(os.listdir(os.path.join(str(tmp_path), sub, 'scene'))) == ['body_gen_000000.pkl', 'body_gen_000001.pkl']


def test_no_check_then_create_left_in_the_package():
    """The idiom itself must not come back on any path several ranks run."""
    pkg = os.path.join(ROOT, 'psi-release_amd')
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith('.py'):
                for line in open(os.path.join(dp, fn)):
                    if 'os.makedirs(' in line and 'exist_ok=True' not in line:
                        bad.append((fn, line.strip()))
    assert not bad, bad
