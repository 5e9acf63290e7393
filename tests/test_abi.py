"""The C-ABI library loads and exports every symbol include/nksr_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'nksr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nksr_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    from nksr_amd import _lib
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), 'libnksr_hip.so does not export %s' % n
    # and the ctypes table binds exactly the declared entry points
    assert set(_lib.EXPORTED) == set(names), set(_lib.EXPORTED) ^ set(names)


def test_struct_layout_matches_header():
    from nksr_amd import _lib
    assert ctypes.sizeof(_lib.LevelT) == 80
    assert ctypes.sizeof(_lib.HierT) == 16 + 6 * 80
    assert ctypes.sizeof(_lib.SiteSetT) == 32 + 2 * 6 * 8 + 16 + 16 + 8
    assert ctypes.sizeof(_lib.FusedOpT) == 16 + 8 + 8 * 8 + 8 + 5 * 8 + 3 * 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(_lib.CoarsePrecondT) == 16 + 8 + 14 * 8
    assert ctypes.sizeof(_lib.SegmentsT) == 8 + 3 * 8
    assert _lib.lib.nksr_version() >= 100
    assert _lib.lib.nksr_pcg_workspace_bytes(1000, 50000) >= 4 * 4000


def test_product_refuses_cpu_and_never_imports_oracle():
    import sys
    import pytest
    import torch
    import nksr
    with pytest.raises(RuntimeError):
        nksr.Reconstructor(torch.device('cpu'))
    with pytest.raises(RuntimeError):
        nksr.SparseFeatureHierarchy(0.1, 4, torch.device('cpu'))
    for root, _, files in os.walk(os.path.join(ROOT, 'nksr_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), '%s imports the oracle' % f


def test_api_surface():
    """Names the reference's call sites use."""
    import nksr
    from nksr import Reconstructor, fields, utils  # noqa: F401  (recons_colored_mesh.py:12)
    from nksr.configs import load_checkpoint_from_url  # noqa: F401  (models/nksr_net.py:17)
    from nksr.fields import KernelField, LayerField, NeuralField  # noqa: F401  (models/nksr_net.py:16)
    from nksr.svh import SparseFeatureHierarchy  # noqa: F401  (models/loss.py:12)
    assert callable(nksr.get_estimate_normal_preprocess_fn(64, 85.0))
    import inspect
    sig = inspect.signature(nksr.Reconstructor.reconstruct)
    for kw in ('normal', 'sensor', 'detail_level', 'voxel_size', 'chunk_size', 'preprocess_fn', 'approx_kernel_grad',
               'solver_tol', 'fused_mode'):
        assert kw in sig.parameters
    sig = inspect.signature(fields.BaseField.extract_dual_mesh)
    for kw in ('mise_iter', 'grid_upsample', 'max_points'):
        assert kw in sig.parameters


def test_csr_physical_layouts_roundtrip_on_cpu():
    """include/nksr_hip.h col_format 0 / 1: the host-side decoder (solver.csr_logical) inverts the documented
    tile interleave and the 21-bit column packing."""
    import numpy as np
    import torch
    from nksr_amd import solver
    rng = np.random.default_rng(0)
    nnz = 5000
    cols = rng.integers(0, 1 << 21, nnz).astype(np.int64)
    vals = rng.standard_normal(nnz).astype(np.float32)
    rowptr = torch.tensor([0, nnz], dtype=torch.int32)
    k = np.arange(nnz)
    # format 0: 256-entry tiles, entry m of a tile at 4 (m % 64) + m / 64, int32 columns
    m = k & 255
    phys = (k & ~255) + 4 * (m & 63) + (m >> 6)
    npad = (nnz + 4095) // 4096 * 4096
    c0, v0 = np.zeros(npad, np.int32), np.zeros(npad, np.float32)
    c0[phys], v0[phys] = cols, vals
    lc, lv = solver.csr_logical(rowptr, torch.from_numpy(c0), torch.from_numpy(v0))
    assert np.array_equal(lc.numpy(), cols) and np.array_equal(lv.numpy(), vals)
    # format 1: 192-entry tiles, entry m at 3 (m % 64) + m / 64, three 21-bit columns per 64-bit word
    t, m = k // 192, k % 192
    phys = t * 192 + 3 * (m & 63) + (m >> 6)
    npad = (nnz + 4607) // 4608 * 4608
    c32, v1 = np.zeros(npad, np.int64), np.zeros(npad, np.float32)
    c32[phys], v1[phys] = cols, vals
    packed = c32[0::3] | (c32[1::3] << 21) | (c32[2::3] << 42)
    lc, lv = solver.csr_logical(rowptr, torch.from_numpy(packed), torch.from_numpy(v1))
    assert lc.dtype == torch.int32 and np.array_equal(lc.numpy(), cols) and np.array_equal(lv.numpy(), vals)
    assert solver.col_format(torch.from_numpy(packed)) == 1 and solver.col_format(torch.from_numpy(c0)) == 0


def test_c_abi_argument_errors_without_a_gpu():
    """Argument validation happens before any launch: these calls fail cleanly (error code + message) on a
    machine without a GPU, never abort."""
    import ctypes as C
    from nksr_amd import _lib
    lib = _lib.lib
    lib.nksr_last_error.restype = C.c_char_p
    null = C.c_void_p(0)

    def err():
        return lib.nksr_last_error().decode()

    assert lib.nksr_pack_cols21(null, C.c_int64(100), null, null) != 0 and '192' in err()
    assert lib.nksr_spmv_csr(null, null, null, C.c_int32(10), C.c_int64(100), C.c_int(0), null, null, null, null) != 0
    assert 'workspace' in err()
    assert lib.nksr_spmv_plan(null, C.c_int32(10), C.c_int64(100), C.c_int(7), null, null) != 0 and 'col_format' in err()
    assert lib.nksr_spmv_plan(null, C.c_int32(3 << 20), C.c_int64(100), C.c_int(1), null, null) != 0 and '2^21' in err()
    assert lib.nksr_splat_trilinear(null, null, C.c_int(9), null, null, null, null, C.c_int32(1), C.c_float(1.0), null, null, null) != 0
    assert 'channels' in err()
    h = _lib.HierT()
    h.depth = 4
    assert lib.nksr_kernel_rows(C.byref(h), null, C.c_int64(5), C.c_int(0), C.c_float(1.0), null, C.c_int64(0), null, null, null, null, null) != 0 and 'NULL' in err()
    assert lib.nksr_fused_block_counts(C.c_int32(9), C.c_int32(10), C.c_int64(5), null, null, null, null, null) != 0 and 'depth' in err()
    assert lib.nksr_fused_block_counts(C.c_int32(4), C.c_int32(10), C.c_int64(5), null, null, null, null, null) != 0 and 'NULL' in err()
    op = _lib.FusedOpT()
    op.depth, op.M = 4, 10
    assert lib.nksr_fused_apply(C.byref(op), C.c_float(1.0), null, null, null) != 0 and 'NULL' in err()
    assert lib.nksr_hash_build(null, C.c_int32(2), null, null, C.c_int32(4), null) != 0 and 'power of two' in err()
    assert lib.nksr_hash_query(null, C.c_int64(2), null, null, C.c_int32(12), null, null) != 0 and 'power of two' in err()
    # round-2 entry points: argument errors are reported before anything is launched
    hh = _lib.HierT()
    hh.depth = 4
    hh.lv[3].n = 5
    assert lib.nksr_fused_tables(C.byref(hh), C.c_int64(0), null, null, null, null, null, null, null) != 0 and 'NULL' in err()
    assert lib.nksr_coarse_lambda_max(null, null, null, null, C.c_int32(5), C.c_int(8), null, null, None, C.c_int32(0), null) != 0 and 'NULL' in err()
    one = (C.c_float * 8)()
    assert lib.nksr_sdf_from_points(one, one, null, null, null, null, null, C.c_int32(0), C.c_float(1.0), C.c_float(1.0), one, C.c_int64(1), C.c_int(0),
                                    C.c_int(4), C.c_float(0.02), C.c_int(0), one, null, one, null) != 0 and 'nb_points' in err()
    assert lib.nksr_sdf_from_points(one, one, null, null, null, null, null, C.c_int32(0), C.c_float(1.0), C.c_float(1.0), one, C.c_int64(1), C.c_int(8),
                                    C.c_int(4), C.c_float(0.0), C.c_int(0), one, null, one, null) != 0 and 'stdv' in err()
    assert lib.nksr_assemble_split_bytes(C.byref(h), C.c_int64(10 ** 6)) == 0           # no voxels: nothing to split
    with __import__('pytest').raises(RuntimeError):
        _lib.call('nksr_pack_cols21', null, 100, null, null)
