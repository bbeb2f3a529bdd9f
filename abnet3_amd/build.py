"""Builds libabnet3_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m abnet3_amd.build [--force] [--verbose]

Cross-compiles without a GPU.  The shared object lands in abnet3_amd/lib/ so
that it travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'lib', 'obj')
LIB = os.path.join(HERE, 'lib', 'libabnet3_hip.so')
SOURCES = ['tower.hip', 'loss.hip', 'ops.hip', 'dtw.hip', 'fbank.hip', 'oneshot.hip']
HEADERS = sorted(h for h in os.listdir(CSRC) if h.endswith('.h')) + [os.path.join('..', '..', 'include', 'abnet3_hip.h')]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall',
         '-Wno-unused-function']
# translation units whose float arithmetic must be identical on CPU and GPU
# (DTW distances: no fused-multiply-add contraction)
STRICT_FP = {'dtw.hip': ['-ffp-contract=off']}


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace('.hip', '.o'))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [cc] + FLAGS + STRICT_FP.get(src, []) + ['-c', path, '-o', obj]
            if verbose:
                cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
                print(' '.join(cmd), flush=True)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0 or verbose:
                sys.stderr.write(r.stdout)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed on %s' % src)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    # (the library's digest is kept beside it: a file that something else wrote over it -- a diagnostic build -- is not ours)
    if force or any(c for _, c in results) or _stale(LIB, objs) or _digest(LIB) != _recorded_digest():
        # Linked beside the target and renamed into place: several ranks may find the library stale at the same moment (the
        # first start after an upgrade), and none of them may dlopen a file another one is still writing.
        tmp = '%s.%d.tmp' % (LIB, os.getpid())
        cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError('link failed')
        digest = _digest(tmp)
        os.replace(tmp, LIB)
        with open(tmp, 'w') as f:
            f.write(digest)
        os.replace(tmp, LIB + '.sha1')
    return LIB


def _digest(path):
    import hashlib
    h = hashlib.sha1()
    with open(path, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()


def _recorded_digest():
    try:
        with open(LIB + '.sha1') as f:
            return f.read().strip()
    except OSError:
        return None


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv))
