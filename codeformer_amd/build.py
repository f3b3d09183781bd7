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
SOURCES = ['cf_igemm.hip', 'cf_winograd.hip', 'cf_split.hip', 'cf_wsplit.hip', 'cf_wf43.hip', 'cf_gemm_split.hip', 'cf_norm.hip', 'cf_attention.hip', 'cf_misc.hip', 'cf_paste.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (set HIPCC)')


def source_hash():
    """sha256 (16 hex digits) over every file the library is compiled from, in a fixed order."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h'))) + \
        [os.path.join(ROOT, 'include', 'codeformer_hip.h')]
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_id():
    """Build id of the shared object on disk (what its cf_build_id() returns), read from the file itself -- loading it here
    would pin the old image in this process across a rebuild.  None when the file is absent or predates the marker."""
    if not os.path.exists(LIB):
        return None
    with open(LIB, 'rb') as fh:
        blob = fh.read()
    i = blob.find(b'CF_BUILD_ID=')
    if i < 0:
        return None
    return blob[i + 12:blob.index(b'\0', i)].decode('ascii', 'replace')


def needs_build():
    """True when the .so is missing or was compiled from other sources than the tree holds (content hash, not mtime: a shipped
    .so that is older or newer than an edited source is rebuilt either way)."""
    return built_id() != source_hash()


FIXED_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c']


def _cache_dir():
    """Object cache: $CF_OBJ_CACHE, else build/objcache inside this checkout (git- and gpurun-ignored), else -- a read-only tree --
    ~/.cache/codeformer_amd/obj; created with mode 0700 and never a shared world-writable directory: whatever sits in the cache under
    the right name is linked into the library."""
    for d in (os.environ.get('CF_OBJ_CACHE'), os.path.join(ROOT, 'build', 'objcache'), os.path.join(os.path.expanduser('~'), '.cache', 'codeformer_amd', 'obj')):
        if not d:
            continue
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            continue
    raise RuntimeError('no writable object cache directory (set CF_OBJ_CACHE)')


def _toolchain_id(hipcc):
    """What else decides an object's bytes besides source, headers and -D flags: the compiler (its --version text) and the fixed flags."""
    try:
        ver = subprocess.run([hipcc, '--version'], capture_output=True, text=True, timeout=60).stdout
    except Exception:   # noqa: BLE001
        ver = 'unknown'
    return (os.path.realpath(hipcc) + '\n' + ver + '\n' + ' '.join(FIXED_FLAGS)).encode()


def _compile_one(args):
    hipcc, src, obj, flags = args
    if os.path.exists(obj):
        return obj, ''
    tmp = obj + f'.{os.getpid()}.tmp'
    r = subprocess.run([hipcc] + FIXED_FLAGS + flags + ['-o', tmp, src], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {os.path.basename(src)}:\n' + r.stdout + r.stderr)
    os.replace(tmp, obj)
    return obj, r.stderr


def build(force=False, verbose=False, out=None, defines=(), extra_sources=()):
    """Compile every source of SOURCES for gfx950 (one hipcc process per file, in parallel; objects are cached under $CF_OBJ_CACHE or
    build/objcache by a hash of source, headers, flags and the compiler's identity, so an edit recompiles one file and a
    ROCm upgrade all of them) and link libcodeformer_hip.so in-tree.
    out / defines / extra_sources: experiment variants (tools/*): another output path, extra -D flags, extra .hip files."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    target = out or LIB
    if out is None and not defines and not extra_sources and not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    cache = _cache_dir()
    tool = _toolchain_id(hipcc)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(ROOT, 'include', 'codeformer_hip.h')]
    hh = hashlib.sha256()
    for f in headers:
        with open(f, 'rb') as fh:
            hh.update(fh.read())
    build_id = source_hash()
    jobs = []
    for src in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.abspath(s) for s in extra_sources]:
        flags = ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + [f'-D{d}' for d in defines]
        if os.path.basename(src) == 'cf_misc.hip':
            flags.append(f'-DCF_BUILD_ID="{build_id}"')    # (only this file reads it: the others stay cached across unrelated edits)
        h = hashlib.sha256(hh.digest())
        h.update(tool)
        with open(src, 'rb') as fh:
            h.update(fh.read())
        h.update(' '.join(flags).encode())
        jobs.append((hipcc, src, os.path.join(cache, f'{os.path.basename(src)}.{h.hexdigest()[:20]}.o'), flags))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        objs = [o for o, _ in ex.map(_compile_one, jobs)]
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', target + '.tmp'] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc link failed:\n' + r.stdout + r.stderr)
    os.replace(target + '.tmp', target)
    return target


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
