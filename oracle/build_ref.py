"""Builds the ONE piece of the reference that ships as source and touches this path's "next" rows: the training ground-truth
generator ``ext.sdfgen.sdf_from_points`` (/root/reference/ext/sdfgen/{bind.cpp,sdf_from_points.cu} + ext/common/kdtree_cuda.cu),
as a PyTorch-ROCm extension for gfx950 -- the reference's own load() recipe (ext/__init__.py:18-23: name 'nksr_sdfgen', -O2)
with torch's CUDA->HIP source translation.  TEST INFRASTRUCTURE: the result ``oracle/_ref/nksr_sdfgen.so`` is loaded only by
tests/test_gpu_sdfgen.py to check csrc/knn.hip's k_sdf_from_points against the reference's OWN kernel.

    python -m oracle.build_ref            (in the dev container: needs /root/reference; hipcc cross-compiles without a GPU)

Nothing of the reference is copied into the repository: the sources are staged in a temporary directory OUTSIDE the repo (torch's
hipify writes its translated files next to its inputs, and /root/reference is read-only), compiled there, and only the shared
object lands in oracle/_ref/ (git-ignored; it travels to the GPU box with the snapshot like the product's own .so).
The hot path itself (the `nksr` wheel) has no source in the reference tree, so nothing else can be built."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/ext'
OUT = os.path.join(ROOT, 'oracle', '_ref')
SOURCES = ['sdfgen/bind.cpp', 'common/kdtree_cuda.cu', 'sdfgen/sdf_from_points.cu']
HEADERS = ['common/kdtree_cuda.cuh', 'common/cutil_math.h']


def build(verbose=False):
    if not os.path.isdir(REF):
        return None                      # the GPU box has no /root/reference: it uses the prebuilt file
    os.environ.setdefault('PYTORCH_ROCM_ARCH', 'gfx950')
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    stage = tempfile.mkdtemp(prefix='nksr_ref_')
    try:
        for rel in SOURCES + HEADERS:
            os.makedirs(os.path.dirname(os.path.join(stage, rel)), exist_ok=True)
            shutil.copy(os.path.join(REF, rel), os.path.join(stage, rel))
        bdir = os.path.join(stage, 'build')
        os.makedirs(bdir)
        load(name='nksr_sdfgen', sources=[os.path.join(stage, s) for s in SOURCES], extra_cflags=['-O2'], extra_cuda_cflags=['-O2'],
             build_directory=bdir, verbose=verbose, is_python_module=False)
        so = os.path.join(bdir, 'nksr_sdfgen.so')
        shutil.copy(so, os.path.join(OUT, 'nksr_sdfgen.so'))
        return os.path.join(OUT, 'nksr_sdfgen.so')
    finally:
        shutil.rmtree(stage, ignore_errors=True)


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
