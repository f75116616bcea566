"""Build libpsi_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m psi_release_amd.build [--force]

Sources: psi-release_amd/csrc/*.hip -> psi-release_amd/lib/libpsi_hip.so (in-tree, so the built
library travels to the GPU box with the repository snapshot).  chamfer.hip is compiled with
-ffp-contract=off (the Chamfer distance expression is part of the parity contract); the other
kernels use the default contraction.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libpsi_hip.so')
LIB_FMA = os.path.join(LIBDIR, 'libpsi_hip_fma.so')     # same library with the Chamfer distance in nvcc --fmad=true form (-DPSI_CHAMFER_FMA)
FMA_FILES = ('chamfer.hip', 'nnindex.hip')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# (no -munsafe-fp-atomics: nothing on the fitting or training path accumulates with floating-point atomics any more; the one kernel that
# does — the target-side gradient of the general chamfer.backward op, which the reference itself scatters with atomicAdd,
# chamfer.cu:180-195 — asks for the hardware instruction explicitly)
COMMON = ['-O3', '-std=c++17', '--offload-arch=' + ARCH, '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']
# -fno-slp-vectorize: hipcc's SLP pass packs the distance arithmetic into a v_pk/SGPR-shuffle mix that measured
# 21% slower on MI355X than the plain stream (gpurun t1: 0.393 vs 0.324 ms at B=32, n=2048, m=32768)
PER_FILE = {'chamfer.hip': ['-ffp-contract=off', '-fno-slp-vectorize'], 'nnindex.hip': ['-ffp-contract=off'],
            'cvae_loss.hip': ['-ffp-contract=off']}     # the operator sequence of geometry.py, association for association


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.join(LIBDIR, 'obj'), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG), 'include', 'psi_hip.h'))
    jobs = []
    objs = []
    objs_fma = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(LIBDIR, 'obj', f[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + COMMON + PER_FILE.get(f, []) + ['-c', src, '-o', obj])
        if f in FMA_FILES:
            obj2 = os.path.join(LIBDIR, 'obj', f[:-4] + '_fma.o')
            if force or _stale(obj2, [src] + headers):
                jobs.append([HIPCC] + COMMON + PER_FILE.get(f, []) + ['-DPSI_CHAMFER_FMA', '-c', src, '-o', obj2])
            objs_fma.append(obj2)
        else:
            objs_fma.append(obj)

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs)
    if force or jobs or _stale(LIB_FMA, objs_fma):
        run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB_FMA] + objs_fma)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
