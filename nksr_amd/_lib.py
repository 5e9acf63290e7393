"""ctypes binding of the C-ABI declared in include/nksr_hip.h.

The product path has NO CPU fallback: if libnksr_hip.so is missing (and cannot be built)
importing this module raises, and every op raises RuntimeError on a non-GPU tensor.
"""
import ctypes as C
import os

import torch

from . import build as _build

MAX_DEPTH = 6
_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class LevelT(C.Structure):
    _fields_ = [('n', _i32), ('offset', _i32), ('keys', _vp), ('ijk', _vp), ('nbr', _vp), ('hkeys', _vp),
                ('hvals', _vp), ('hcap', _i32), ('feat', _vp), ('psi', _vp), ('mlp', _vp)]


class HierT(C.Structure):
    _fields_ = [('depth', _i32), ('kdim', _i32), ('hidden', _i32), ('inv_w0', _f32), ('lv', LevelT * MAX_DEPTH)]


class ThetaGradT(C.Structure):
    _fields_ = [('gfeat', _vp * MAX_DEPTH), ('gpsi', _vp * MAX_DEPTH), ('gmlp', _vp * MAX_DEPTH)]


KNN_LEVELS = 12


class KnnPyramidT(C.Structure):
    _fields_ = [('xyz_sorted', _vp), ('start', _vp * KNN_LEVELS), ('end', _vp * KNN_LEVELS), ('child', _vp * KNN_LEVELS),
                ('cmask', _vp * KNN_LEVELS), ('hkeys', _vp * KNN_LEVELS), ('hvals', _vp * KNN_LEVELS), ('hcap', _i32 * KNN_LEVELS),
                ('levels', _i32), ('leaf', _i32), ('cell', _f32), ('inv_cell', _f32)]


CELL_SIZES = 12


class CellTableT(C.Structure):
    _fields_ = [('nlev', _i32), ('lam', _i32 * CELL_SIZES), ('offset', _i32 * CELL_SIZES), ('hcap', _i32 * CELL_SIZES),
                ('hkeys', _vp * CELL_SIZES), ('hvals', _vp * CELL_SIZES)]


class FusedOpT(C.Structure):
    _fields_ = [('depth', _i32), ('M', _i32), ('n_multi', _i32), ('n_big', _i32), ('rows_total', _i64), ('rows_all', _vp),
                ('targets_all', _vp), ('row_cells', _vp), ('nbr32', _vp), ('nbrT', _vp), ('item_begin', _vp), ('offsets', _vp), ('multi', _vp), ('nblocks', _i64),
                ('nnz_counter', _vp), ('workspace', _vp), ('cell_sums', _vp), ('item_seg', _vp), ('unknown_seg', _vp),
                ('fac_vec', _vp), ('fac_pos', _vp), ('psi_all', _vp), ('inv_w0', _f32), ('dense_from', _i32), ('dense_out', _vp), ('compact', _i32), ('rows_words', _i64)]


class ChunkGridT(C.Structure):
    _fields_ = [('grid', _i32 * 3), ('reach', _i32), ('origin', _f32 * 3), ('inv_cs', _f32), ('inv_2ov', _f32),
                ('lo_sel', _vp * 3), ('hi_sel', _vp * 3), ('lo_w', _vp * 3), ('hi_w', _vp * 3), ('shift', _vp)]


class SegmentsT(C.Structure):
    _fields_ = [('nseg', _i32), ('nranges', _i32), ('lo', _vp), ('hi', _vp), ('info', _vp)]


class CoarsePrecondT(C.Structure):
    _fields_ = [('first', _i32), ('n', _i32), ('steps', _i32), ('format', _i32), ('lambda_scale', _f32), ('ratio', _f32),
                ('lambda_', _vp), ('row_seg', _vp), ('rowptr', _vp), ('cols', _vp), ('vals', _vp), ('diag', _vp), ('work', _vp), ('coef', _vp),
                ('packed', _vp), ('packed_rowptr', _vp), ('dis', _vp), ('old_of_new', _vp), ('seg_base', _vp), ('gersh', _vp)]


PC_MAX_STEPS = 16


class SiteSetT(C.Structure):
    _fields_ = [('n', _i64), ('ncomp', _i32), ('weight', _f32), ('val', _vp), ('target', _vp),
                ('start', _vp * MAX_DEPTH), ('end', _vp * MAX_DEPTH), ('level_stride', _i64), ('row_index', _vp), ('compact_cells', _vp), ('compact_nbr32', _vp), ('level_base', _i64)]


def _load():
    path = _build.LIB
    if _build.needs_build():
        try:
            _build.build_library()
        except Exception as e:  # stale/missing library and no working compiler: fail loudly, never load a stale build silently
            if not os.path.exists(path):
                raise RuntimeError('libnksr_hip.so is missing and could not be built: %s' % e)
            if not os.environ.get('NKSR_ALLOW_STALE_LIB'):
                raise RuntimeError('libnksr_hip.so is older than its sources and the rebuild failed (%s); set '
                                   'NKSR_ALLOW_STALE_LIB=1 to load it anyway' % e)
    return C.CDLL(path)


lib = _load()
lib.nksr_last_error.restype = C.c_char_p
lib.nksr_pcg_workspace_bytes.restype = _sz
lib.nksr_bbox_work_floats.restype = _i64
lib.nksr_bbox_work_floats.argtypes = []
lib.nksr_pcg_workspace_bytes.argtypes = [_i32, _i64]
lib.nksr_assemble_workspace_bytes.restype = _sz
lib.nksr_assemble_workspace_bytes.argtypes = [C.POINTER(HierT)]
lib.nksr_assemble_split_bytes.restype = _sz
lib.nksr_assemble_split_bytes.argtypes = [C.POINTER(HierT), _i64]
lib.nksr_spmv_workspace_bytes.restype = _sz
lib.nksr_spmv_workspace_bytes.argtypes = [_i64]
lib.nksr_fused_item_entries.restype = _i64
lib.nksr_fused_item_entries.argtypes = [_i64]
lib.nksr_fused_workspace_bytes.restype = _sz
lib.nksr_fused_workspace_bytes.argtypes = [_i64, _i32]
lib.nksr_conv3_wgrad_chunks.restype = _i64
lib.nksr_conv3_wgrad_chunks.argtypes = [_i32]
lib.nksr_pcg_vector_workspace_bytes.restype = _sz
lib.nksr_pcg_vector_workspace_bytes.argtypes = [_i32]
lib.nksr_pcg_profile_survey_bytes.restype = C.c_double
lib.nksr_pcg_profile_survey_bytes.argtypes = []
lib.nksr_pcg_profile_samples.restype = _i64
lib.nksr_pcg_profile_samples.argtypes = [_vp, _i64]
lib.nksr_pcg_vector_workspace_bytes_seg.restype = _sz
lib.nksr_pcg_vector_workspace_bytes_seg.argtypes = [_i32, _i32, _i32]

_P = C.POINTER
_PROTOS = {
    'nksr_sort_keys_u64': [_vp, _P(_sz), _vp, _vp, _i64, C.c_int, C.c_int, _vp],
    'nksr_sort_pairs_u64_u32': [_vp, _P(_sz), _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp],
    'nksr_unique_u64': [_vp, _P(_sz), _vp, _vp, _vp, _i64, _vp],
    'nksr_exclusive_sum_i32': [_vp, _P(_sz), _vp, _vp, _i64, _vp],
    'nksr_exclusive_sum_i64': [_vp, _P(_sz), _vp, _vp, _i64, _vp],
    'nksr_splat_keys': [_vp, _i64, _f32, C.c_int, C.c_int, _vp, _vp],
    'nksr_cell_footprint_keys': [_vp, _i64, C.c_int, C.c_int, _vp, _vp],
    'nksr_bbox': [_vp, _i64, _vp, _vp, _vp],
    'nksr_footprint_keys_dedup': [_vp, _vp, _i64, C.c_float, C.c_int, C.c_int, _vp, _vp, _vp],
    'nksr_point_keys': [_vp, _i64, _f32, _vp, _vp],
    'nksr_decode_keys': [_vp, _i64, C.c_int, _vp, _vp],
    'nksr_encode_keys': [_vp, _i64, C.c_int, _vp, _vp],
    'nksr_hash_build': [_vp, _i32, _vp, _vp, _i32, _vp],
    'nksr_hash_query': [_vp, _i64, _vp, _vp, _i32, _vp, _vp],
    'nksr_build_nbr': [_vp, _i32, C.c_int, _vp, _vp, _i32, _vp, _vp],
    'nksr_build_nbr_from_parent': [_vp, _vp, _i32, C.c_int, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp],
    'nksr_site_ranges': [_vp, _i64, _vp, _i32, C.c_int, _vp, _vp, _vp],
    'nksr_sorted_lookup': [_vp, _i64, _vp, _i64, _vp, _vp],
    'nksr_rank_sorted': [_vp, _i64, _vp, _i64, C.c_int, _vp, _vp],
    'nksr_splat_trilinear': [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _vp],
    'nksr_point_mlp': [_vp, _vp, _i64, _f32, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    'nksr_splat_mean': [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp],
    'nksr_sparse_conv3': [_vp, _vp, _i32, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp],
    'nksr_pool_children': [_vp, _vp, _vp, _i32, C.c_int, _vp, _vp],
    'nksr_conv3_wgrad': [_vp, _vp, _i32, C.c_int, _vp, _vp, _vp],
    'nksr_gather_rows': [_vp, _vp, _i64, C.c_int, _vp, _vp, _vp],
    'nksr_linear': [_vp, _i64, C.c_int, _vp, _vp, C.c_int, _vp, _vp],
    'nksr_splat_plane': [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp],
    'nksr_udf_decode': [_P(LevelT), C.c_int, _vp, _vp, _i64, _f32, _f32, C.c_int, _vp, _vp],
    'nksr_voxel_psi': [_vp, _i32, C.c_int, C.c_int, _vp, _vp, _vp],
    'nksr_kernel_rows': [_P(HierT), _vp, _i64, C.c_int, _f32, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    'nksr_evaluate_f': [_P(HierT), _vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp],
    'nksr_kernel_rows_vjp': [_P(HierT), _vp, _i64, C.c_int, C.c_int, _f32, _vp, _vp, _vp, _vp, _P(ThetaGradT), _vp],
    'nksr_voxel_psi_vjp': [_vp, _i32, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp],
    'nksr_assemble_count': [_P(HierT), _vp, _vp, _vp, _vp, _vp, _vp],
    'nksr_assemble': [_P(HierT), _P(SiteSetT), C.c_int, _f32, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    'nksr_place_mirrors': [_vp, _vp, _i64, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp],
    'nksr_spmv_set_variant': [C.c_int],
    'nksr_spmv_plan': [_vp, _i32, _i64, C.c_int, _vp, _vp],
    'nksr_pack_cols21': [_vp, _i64, _vp, _vp],
    'nksr_spmv_csr': [_vp, _vp, _vp, _i32, _i64, C.c_int, _vp, _vp, _vp, _vp],
    'nksr_pcg_solve': [_vp, _vp, _vp, _vp, _i32, _i64, C.c_int, _vp, _vp, _f32, C.c_int, C.c_int, _vp, _P(CoarsePrecondT), _P(C.c_double), _vp],
    'nksr_fused_block_counts': [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp],
    'nksr_fused_tables': [_P(HierT), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'nksr_fused_row_sizes': [_P(HierT), _vp, _vp, _vp],
    'nksr_row_cells_merged': [_P(HierT), _vp, _vp, _vp, _i64, _vp, _vp],
    'nksr_fused_rhs_diag': [_P(FusedOpT), _f32, _vp, _vp, _vp],
    'nksr_fused_apply': [_P(FusedOpT), _f32, _vp, _vp, _vp],
    'nksr_fused_expand_rows': [_P(FusedOpT), _vp],
    'nksr_kernel_rows_merged': [_P(HierT), _vp, _vp, _f32, _vp, _vp, _f32, C.c_int, _vp, _i64, _vp, C.c_int, _vp, _vp, _vp],
    'nksr_row_sources': [_vp, _i64, C.c_int, C.c_int, _vp, _vp],
    'nksr_kernel_factors': [_P(HierT), _vp, _i64, C.c_int, C.c_int, _f32, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    'nksr_pcg_solve_fused': [_P(FusedOpT), _f32, _vp, _vp, _vp, _f32, C.c_int, C.c_int, _vp, _P(CoarsePrecondT), _P(SegmentsT), _P(C.c_double), _vp],
    'nksr_coarse_lambda_max': [_vp, _vp, _vp, _vp, _i32, C.c_int, _vp, _vp, _P(SegmentsT), _i32, _vp],
    'nksr_coarse_pack_count': [_vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp],
    'nksr_coarse_pack': [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp],
    'nksr_coarse_lambda_max_packed': [_P(CoarsePrecondT), _i32, C.c_int, _vp, _vp, _vp],
    'nksr_coarse_gershgorin': [_P(CoarsePrecondT), _i32, _vp, _vp, _vp],
    'nksr_pcg_profile': [C.c_int, _P(C.c_double), _P(_i64)],
    'nksr_pcg_profile_bytes': [_P(C.c_double), _P(C.c_double)],
    'nksr_chunk_pair_counts': [_P(ChunkGridT), C.c_int, _vp, _i64, _vp, _vp, _vp],
    'nksr_chunk_pair_fill': [_P(ChunkGridT), C.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'nksr_chunk_blend': [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'nksr_edge_seam_flags': [_P(ChunkGridT), _vp, _vp, _i64, _i32, _f32, _vp, _i32, _vp, _vp],
    'nksr_points_owner_flags': [_P(ChunkGridT), _vp, _i64, _f32, _vp, _i32, _vp, _vp],
    'nksr_halo_band_flags': [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _f32, _vp, _vp, _vp],
    'nksr_knn_pca_normals': [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _f32, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp],
    'nksr_nearest_index': [_vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _i64, C.c_int, _vp, _vp],
    'nksr_sdf_from_points': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp, _i64, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp, _vp, _vp],
    'nksr_knn_mean_dist': [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _f32, C.c_int, C.c_int, _vp, _vp, _vp],
    'nksr_knn_pyramid_level': [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    'nksr_sdf_from_points_pyramid': [_P(KnnPyramidT), _vp, _vp, _vp, _i64, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp, _vp, _vp],
    'nksr_knn_mean_dist_pyramid': [_P(KnnPyramidT), _i64, C.c_int, C.c_int, _vp, _vp, _vp],
    'nksr_base_cell_flags': [_vp, _i32, _vp, _vp],
    'nksr_base_cell_keys': [_vp, _vp, _i64, C.c_int, _vp, _vp],
    'nksr_level_cell_keys': [_vp, _vp, _i64, C.c_int, C.c_int, _vp, _vp],
    'nksr_cell_corner_keys': [_vp, _i64, _vp, _vp],
    'nksr_lattice_positions': [_vp, _i64, _f32, _f32, _vp, _vp],
    'nksr_cell_config': [_vp, _vp, _i64, _vp, _vp, _vp],
    'nksr_cell_active_flags': [_vp, _i64, _vp, _vp],
    'nksr_compact_block_counts': [_vp, _i64, _vp, _vp],
    'nksr_compact_scatter': [_vp, _i64, _vp, _vp, _vp],
    'nksr_mise_constrain': [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    'nksr_cell_children': [_vp, _vp, _i64, _vp, _vp],
    'nksr_mc_emit': [_vp, _vp, _vp, _i64, _vp, _vp],
    'nksr_mc_vertices': [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _f32, _vp, _vp],
    'nksr_adaptive_corner_keys': [_vp, _i64, C.c_int, _vp, _vp],
    'nksr_adaptive_dual_cells': [_vp, _i64, _P(CellTableT), _vp, _vp],
    'nksr_adaptive_positions': [_vp, _vp, _i64, _f32, _vp, _vp],
    'nksr_mc_emit_pairs': [_vp, _vp, _vp, _i64, _vp, _vp],
    'nksr_pair_vertices': [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp],
}
for _name, _args in _PROTOS.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = C.c_int

EXPORTED = ['nksr_points_owner_flags', 'nksr_edge_seam_flags', 'nksr_halo_band_flags', 'nksr_conv3_wgrad_chunks', 'nksr_pcg_profile_samples', 'nksr_last_error', 'nksr_version', 'nksr_pcg_workspace_bytes', 'nksr_spmv_workspace_bytes', 'nksr_assemble_workspace_bytes',
            'nksr_assemble_split_bytes',
            'nksr_fused_workspace_bytes', 'nksr_fused_item_entries', 'nksr_pcg_vector_workspace_bytes', 'nksr_pcg_vector_workspace_bytes_seg', 'nksr_pcg_profile_survey_bytes', 'nksr_bbox_work_floats'] + sorted(_PROTOS)


def check(rc):
    if rc != 0:
        msg = lib.nksr_last_error().decode(errors='replace')
        if 'out of memory' in msg.lower():
            raise MemoryError(msg)
        raise RuntimeError('nksr_hip: %s (code %d)' % (msg, rc))


def ptr(t):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('nksr_amd is MI355X-only: expected a GPU tensor, got device %s' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('expected a contiguous tensor')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(device):
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError("nksr_amd is MI355X-native: device must be 'cuda' (got %s). The CPU restatement of "
                           "this path lives in oracle/ and is test infrastructure only." % device)
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible to PyTorch-ROCm')
    return device


def call(name, *args):
    check(getattr(lib, name)(*args))


def with_tmp(name, device, *args_after_tmp):
    """Run a rocPRIM-backed primitive: size query, allocate, run."""
    nbytes = _sz(0)
    fn = getattr(lib, name)
    check(fn(None, C.byref(nbytes), *args_after_tmp))
    tmp = torch.empty(max(int(nbytes.value), 16), dtype=torch.uint8, device=device)
    check(fn(tmp.data_ptr(), C.byref(nbytes), *args_after_tmp))
    return tmp
