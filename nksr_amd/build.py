"""In-tree build of the HIP extension (libnksr_hip.so) for gfx950.

No torch types cross the boundary, so the library is built with plain hipcc and loaded
through ctypes (nksr_amd/_lib.py).  The .so stays in-tree (git-ignored) so it travels to
the GPU box with the snapshot.
"""
import fcntl
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libnksr_hip.so')
SOURCES = ['prims.hip', 'hierarchy.hip', 'kfield.hip', 'rows.hip', 'evalf.hip', 'assemble.hip', 'pcg.hip', 'fused.hip', 'meshing.hip', 'nn.hip', 'knn.hip', 'chunks.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result'] + os.environ.get('NKSR_EXTRA_HIPCC_FLAGS', '').split()


def _deps():
    files = [os.path.join(CSRC, s) for s in SOURCES]
    files += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'nksr_hip.h'))
    return files


STAMP = LIB + '.srchash'        # hash of the sources + flags the library was built from (travels with the .so)


def source_hash():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in sorted(_deps()):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


KERNEL_SOURCES = {'fused': ['fused.hip', 'pcg_core.h', 'common.h'],      # k_fz_sweep / k_fz_gather / k_fz_cellsum
                  'spmv': ['pcg.hip', 'pcg_core.h', 'common.h']}         # k_spmv / k_spmv_fixup


def kernel_hash(kind):
    """Hash of the sources (+ flags) that define one of the roofline kernels: the counter records under profiles/ carry it, and
    bench.py attaches a record to a line only when it was taken on the same kernel code."""
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in KERNEL_SOURCES[kind]:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    """Stale when the library is missing or was built from other sources / flags.  Content hash, not mtimes: a
    snapshot copied to another box (gpurun) does not preserve them."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def _compile(src):
    extra = os.environ.get('NKSR_EXTRA_HIPCC_FLAGS', '')
    tag = '.' + hashlib.sha1(extra.encode()).hexdigest()[:8] if extra else ''       # (objects of a probe build do not pass for the product's)
    obj = os.path.join(CSRC, src.replace('.hip', tag + '.o'))
    hdr_t = max(os.path.getmtime(f) for f in _deps() if f.endswith('.h'))
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(os.path.join(CSRC, src)), hdr_t):
        return obj
    cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj


def build_library(force=False, verbose=False):
    """Compile + link under an exclusive file lock (one process per GPU imports this package at the same time under
    torchrun); the library is linked to a temporary name and renamed into place, so a concurrent loader never sees a
    truncated file."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not needs_build():
        return LIB
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():      # another process built it while we waited
                return LIB
            if not os.path.exists(HIPCC):
                raise RuntimeError('hipcc not found at %s and %s is missing/stale' % (HIPCC, LIB))
            stamp = source_hash()
            with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
                objs = list(ex.map(_compile, srcs))
            tmp = '%s.tmp.%d' % (LIB, os.getpid())
            cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
            os.replace(tmp, LIB)
            with open(STAMP + '.tmp', 'w') as fh:
                fh.write(stamp + '\n')
            os.replace(STAMP + '.tmp', STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
