"""Build libcodeformer_hip.so (gfx950) in-tree with hipcc.

`python -m codeformer_amd.build` or `codeformer_amd.build.build()`.  hipcc cross-compiles for gfx950 without a GPU.
The shared object lands next to this file so that it travels with the source tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, 'libcodeformer_hip.so')
SOURCES = ['cf_igemm.hip', 'cf_winograd.hip', 'cf_split.hip', 'cf_norm.hip', 'cf_attention.hip', 'cf_misc.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (set HIPCC)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, 'include', 'codeformer_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-o', LIB + '.tmp'] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
