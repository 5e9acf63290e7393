"""Builds the ONE piece of the reference that ships as source and touches this path's "next" rows: the training ground-truth
generator ``ext.sdfgen.sdf_from_points`` (/root/reference/ext/sdfgen/{bind.cpp,sdf_from_points.cu} + ext/common/kdtree_cuda.cu,
cutil_math.h), compiled FROM THE REFERENCE'S OWN SOURCES as a PyTorch-ROCm extension for gfx950.  TEST INFRASTRUCTURE: the result
``oracle/_ref/nksr_sdfgen.so`` is loaded only by tests/test_gpu_sdfgen.py, to check csrc/knn.hip's k_sdf_from_points and the
restatement oracle/sdfgen.py against the reference's OWN kernels (kd-tree kNN + estimator) on the GPU box.

    python -m oracle.build_ref [-v]       (in the dev container: needs /root/reference; hipcc cross-compiles without a GPU, ~90 s)

Nothing of the reference is copied into the repository: the sources are staged in a temporary directory OUTSIDE the repo,
translated and compiled there, and only the shared object lands in oracle/_ref/ (git-ignored; it travels to the GPU box with
the snapshot like the product's own .so).  The recipe, step by step (the reference's ext/__init__.py:18-23 JIT recipe -- name
'nksr_sdfgen', -O2 -- does not run here: torch's hipify never returns on kdtree_cuda.cu):
  1. CUDA -> HIP names: torch.utils.hipify for sdf_from_points.cu (it knows the ATen / c10 names), /opt/rocm/bin/hipify-perl for
     kdtree_cuda.cu, kdtree_cuda.cuh, cutil_math.h;
  2. two source-level patches of the STAGED copies, neither touches arithmetic: (a) cutil_math.h defines component-wise
     operators for float2/3/4, int2/3/4 ... -- HIP's vector types (HIP_vector_type) define the same operators themselves and the
     two sets are ambiguous: the 155 free ``operator`` functions are dropped, HIP's take their place (same component-wise
     semantics); (b) ``<<<grid, block >> >`` (kdtree_cuda.cu:1188ff, a formatter's split of the closing chevrons) -> ``>>>``;
  3. hipcc -O2 --offload-arch=gfx950 per file, g++ for bind.cpp, linked against this image's torch libraries.
The hot path itself (the `nksr` wheel) has no source in the reference tree, so nothing else can be built."""
import os
import re
import shutil
import subprocess
import sys
import sysconfig
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/ext'
OUT = os.path.join(ROOT, 'oracle', '_ref')
LIB = os.path.join(OUT, 'nksr_sdfgen.so')
FILES = ['sdfgen/bind.cpp', 'sdfgen/sdf_from_points.cu', 'common/kdtree_cuda.cu', 'common/kdtree_cuda.cuh', 'common/cutil_math.h']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
HIPIFY_PERL = '/opt/rocm/bin/hipify-perl'


def _strip_operators(src):
    """Drops every free function named operator... (signature line ``inline __host__ __device__ T operator..(``, body up to the
    closing brace in column 0).  Returns (text, number of functions dropped)."""
    lines = src.split('\n')
    out, i, n = [], 0, 0
    while i < len(lines):
        if re.match(r'^inline\s+(__host__|__device__)\s+(__host__|__device__)\s+\S+\s+operator\S*\s*\(', lines[i]):
            while not lines[i].startswith('}'):
                i += 1
            i += 1
            n += 1
            continue
        out.append(lines[i])
        i += 1
    return '\n'.join(out), n


def _run(cmd, cwd, verbose):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if verbose or r.returncode:
        sys.stderr.write(' '.join(cmd)[:400] + '\n' + r.stdout[-2000:] + r.stderr[-4000:] + '\n')
    if r.returncode:
        raise RuntimeError('oracle/build_ref: %s failed' % cmd[0])
    return r.stdout


def loadable():
    return os.path.exists(LIB)


def load():
    """The reference's extension module (``.sdf_from_points(queries, ref_xyz, ref_normal, nb_points, stdv, compute_grad, imls,
    adaptive_knn)``, ext/sdfgen/bind.cpp:10-15) or None when oracle/_ref/ holds no build."""
    if not loadable():
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location('nksr_sdfgen', LIB)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(verbose=False, force=False):
    if not os.path.isdir(REF):
        return LIB if loadable() else None      # the GPU box has no /root/reference: it uses the prebuilt file
    if loadable() and not force:
        return LIB
    from torch.utils import cpp_extension as ce
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT, exist_ok=True)
    stage = tempfile.mkdtemp(prefix='nksr_ref_')
    try:
        for rel in FILES:
            os.makedirs(os.path.dirname(os.path.join(stage, rel)), exist_ok=True)
            shutil.copy(os.path.join(REF, rel), os.path.join(stage, rel))
            os.chmod(os.path.join(stage, rel), 0o644)
        # 1. names
        src = os.path.join(stage, 'sdfgen/sdf_from_points.cu')
        hipify_python.hipify(project_directory=stage, output_directory=stage, includes=[src], extra_files=[src], show_detailed=False,
                             is_pytorch_extension=True, hipify_extra_files_only=True)
        for rel in ('common/kdtree_cuda.cu', 'common/kdtree_cuda.cuh', 'common/cutil_math.h'):
            p = os.path.join(stage, rel)
            text = _run([HIPIFY_PERL, p], stage, False)
            open(p[:-3] + '.hip' if rel.endswith('.cu') else p, 'w').write(text)
        # 2. patches of the staged copies
        p = os.path.join(stage, 'common/cutil_math.h')
        text, n = _strip_operators(open(p).read())
        if n < 100:
            raise RuntimeError('oracle/build_ref: expected ~155 operator overloads in cutil_math.h, found %d' % n)
        open(p, 'w').write(text)
        p = os.path.join(stage, 'common/kdtree_cuda.hip')
        text = open(p).read().replace('>> >', '>>>').replace('<< <', '<<<')
        open(p, 'w').write(text)
        # 3. compile + link
        inc = ['-I' + i for i in ce.include_paths(device_type='cuda') + [sysconfig.get_paths()['include']]]
        common = inc + ['-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-DTORCH_EXTENSION_NAME=nksr_sdfgen', '-DTORCH_API_INCLUDE_EXTENSION_H',
                        '-D_GLIBCXX_USE_CXX11_ABI=1', '-fPIC', '-std=c++17', '-O2']
        dev = ['--offload-arch=gfx950', '-fno-gpu-rdc']
        _run([HIPCC] + common + dev + ['-c', 'common/kdtree_cuda.hip', '-o', 'kdtree.o'], stage, verbose)
        _run([HIPCC] + common + dev + ['-c', 'sdfgen/sdf_from_points.hip', '-o', 'sdf.o'], stage, verbose)
        _run(['g++'] + common + ['-c', 'sdfgen/bind.cpp', '-o', 'bind.o'], stage, verbose)
        tl = ce.library_paths(device_type='cuda')
        _run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', 'nksr_sdfgen.so', 'bind.o', 'sdf.o', 'kdtree.o'] +
             ['-L' + d for d in tl] + ['-Wl,-rpath,' + tl[0], '-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch', '-ltorch_python', '-lamdhip64'],
             stage, verbose)
        shutil.copy(os.path.join(stage, 'nksr_sdfgen.so'), LIB + '.tmp')
        os.replace(LIB + '.tmp', LIB)
        return LIB
    finally:
        if not os.environ.get('NKSR_REF_KEEP_STAGE'):
            shutil.rmtree(stage, ignore_errors=True)
        else:
            sys.stderr.write('stage kept: %s\n' % stage)


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv, force='--force' in sys.argv))
