"""Reconstructor -- host-side mirror of ``nksr.Reconstructor``.

Reference interface (call sites): ``nksr.Reconstructor(device)`` examples/recons_simple.py:25;
attributes ``network`` / ``chunk_tmp_device`` recons_by_chunk.py:26-27, NKSR-USAGE.md:164;
``reconstruct(xyz, normal=None, sensor=None, detail_level=, voxel_size=, chunk_size=,
preprocess_fn=, approx_kernel_grad=, solver_tol=, fused_mode=)`` recons_simple.py:26,
recons_by_chunk.py:29, recons_scannet.py:28, recons_waymo.py:30-37, gis_app.py:38-42;
semantics of detail_level / voxel_size NKSR-USAGE.md:129-137.  Solver weights follow
models/nksr_net.py:103-112.
"""
import time

import torch

from . import _lib, configs
from .fields import KernelField, LayerField, NeuralField
from .nn.network import NKSRNetwork
from .svh import SparseFeatureHierarchy


class Reconstructor:
    def __init__(self, device, config='ks', hparams=None):
        self.device = _lib.require_gpu(device)
        self.hparams = hparams if hparams is not None else configs.get_hparams(config)
        self.network = NKSRNetwork(self.hparams).to(self.device)
        self.network.eval()
        self.chunk_tmp_device = self.device
        self.timing = {}
        self.sync_timing = False   # insert stream syncs so that per-stage wall times are exact
        # chunk mode: ALL chunks of a rank are solved as one block-diagonal system (nksr_amd/chunking.py); chunk_batch_points caps the
        # points (band included) of one such batch.  None = automatic: the batches follow the FREE device memory (~4.5 KB of HBM per solved
        # point at tree_depth 5; 70 % of what is free + what torch's allocator holds unused -- NKSR_FREE_HBM_GB overrides the free figure --,
        # at most 2^25 points).  Results do not depend on it.
        self.chunk_batch_points = None
        self.dual_graph = 'lattice'      # 'adaptive': extract_dual_mesh on the adaptive dual graph (cells as large as their level; nksr_amd/meshing.py); set it BEFORE reconstruct() when the field is spread over ranks: the halos are deeper for it, chunking.halo_inner
        self.chunk_spill_dir = None      # chunk mode, batches parked on a CPU chunk_tmp_device: a directory -> the parked batches live in unlinked files there
        #                                  (chunking.spill_to_disk: out-of-core beyond host memory; not part of the reference surface)
        self.coarse_precond = None  # matrix-free solve: None = automatic (coarse-level block preconditioner for 5+ levels), False = Jacobi only,
        #                             or {'first_level', 'steps', 'ratio'} (fields/kernel_field.py _coarse_precond)
        self.row_format = None      # matrix-free solve: None / 'dense' = 27-slot kernel rows (the fast one), 'factors' = 16-byte factor records the sweep
        #                             rebuilds the rows from (kernel_dim 4: a fifth of the row memory, twice the time per application; fields/kernel_field.py)
        self.keep_solve_inputs = False   # parity tests: the field keeps the site sets / weights of its solve (field._solve_inputs)
        self.col_format = 1        # physical layout of the assembled matrix (include/nksr_hip.h); int32 columns when M > 2^21

    # ---- scale selection (NKSR-USAGE.md:129-137) ---------------------------------------------------
    def _global_scale(self, xyz, detail_level, voxel_size):
        if voxel_size is not None:
            return self.hparams.voxel_size / float(voxel_size)
        if detail_level is None:
            return 1.0
        from .density import scale_for_detail_level
        return scale_for_detail_level(xyz, float(detail_level), self.hparams.voxel_size)

    # ---- one chunk: hierarchy -> features -> kernel solve -> mask -------------------------------------
    def _reconstruct_single(self, xyz, normal, approx_kernel_grad, solver_max_iter, solver_tol, fused_mode, chunks=None):
        """``chunks`` = (ids, key_lo, key_hi, frame): the cloud is a batch of chunks in the exploded frame (nksr_amd/chunking.py) -- one
        hierarchy, one network pass, ONE block-diagonal solve whose diagonal blocks (segments) are the chunks; every chunk keeps
        the solver weights of its own point / normal-site counts (models/nksr_net.py:103-111)."""
        from . import ops
        with ops.key_hint(self._key_bits(xyz, normal)):
            return self._reconstruct_hinted(xyz, normal, approx_kernel_grad, solver_max_iter, solver_tol, fused_mode, chunks)

    def _key_bits(self, xyz, normal):
        """ONE readback before anything is built: the boxes of the cloud and of its normals (nksr_bbox; a non-finite value anywhere
        comes back as NaN) = the input check, the lattice range check and the bit range of every Morton key sort that follows."""
        from . import ops
        from .density import bbox_center
        from .svh import inv_w0_f32
        lo, hi, _ = bbox_center(xyz)
        nlo, _, _ = bbox_center(normal)
        v = torch.cat([lo, hi, nlo[:1]]).tolist()
        if not all(abs(c) < float('inf') for c in v[:6]):        # (NaN compares false)
            raise RuntimeError('non-finite coordinates in the input')
        if not abs(v[6]) < float('inf'):
            raise RuntimeError('non-finite normals in the input')
        inv = inv_w0_f32(self.hparams.voxel_size)
        amax = max(abs(c) for c in v[:6])
        if not (amax * inv < (1 << 20) - 8):
            raise RuntimeError('coordinates out of range: |x| / voxel_size must stay below 2^20 (got %g); '
                               'recentre the cloud or use a larger voxel_size' % (amax * inv))
        import math
        return ops.KeyBits([math.floor(c * inv) - 1 for c in v[:3]], [math.floor(c * inv) + 1 for c in v[3:6]], depth=self.hparams.tree_depth)

    def _reconstruct_hinted(self, xyz, normal, approx_kernel_grad, solver_max_iter, solver_tol, fused_mode, chunks=None):
        hp = self.hparams
        t = {}
        tic = time.perf_counter()
        # one Morton sort of the cloud serves the hierarchy builds, the encoder and the assembly
        from .nn.network import sort_cloud
        from .svh import inv_w0_f32
        ks, xyz, normal = sort_cloud(xyz, normal, inv_w0_f32(hp.voxel_size))
        cells = SparseFeatureHierarchy.cells_with_points(ks, hp.tree_depth)       # shared by both hierarchies
        enc_svh = SparseFeatureHierarchy(hp.voxel_size, hp.tree_depth, self.device).build_point_splatting_sorted(xyz, ks, cells)
        enc = self.network.encoder(xyz, normal, enc_svh, 0, sorted_keys=ks)
        enc.cells = cells
        feat, dec_svh, udf_svh = self.network.unet(enc, enc_svh, adaptive_depth=hp.adaptive_depth)
        if all(dec_svh.grids[d] is None for d in range(hp.adaptive_depth)):
            raise RuntimeError('empty decoder hierarchy')
        field = KernelField(svh=dec_svh, interpolator=self.network.interpolators, features=feat.basis_features,
                            approx_kernel_grad=approx_kernel_grad)
        field.solver_config.update({'max_iter': int(solver_max_iter), 'tol': float(solver_tol), 'sync_timing': self.sync_timing,
                                    'col_format': self.col_format, 'coarse_precond': self.coarse_precond, 'row_format': self.row_format})
        normal_xyz = torch.cat([dec_svh.get_voxel_centers(d) for d in range(hp.adaptive_depth)])
        normal_value = torch.cat([feat.normal_features[d] for d in range(hp.adaptive_depth)])
        if self.sync_timing:
            torch.cuda.current_stream().synchronize()
        t['t_network'] = time.perf_counter() - tic
        inputs = dict(pos_xyz=enc.xyz, normal_xyz=normal_xyz, normal_value=-normal_value,
                      pos_weight=hp.solver.pos_weight / xyz.shape[0],
                      normal_weight=hp.solver.normal_weight / normal_xyz.shape[0] * hp.voxel_size ** 2,
                      reg_weight=1.0, pos_sorted_keys=enc.keys,
                      normal_sorted_keys=dec_svh.level(0).keys if hp.adaptive_depth == 1 else None)
        if chunks is not None:
            from .fields.kernel_field import Segments
            ids, klo, khi, _ = chunks
            seg = Segments(dec_svh, torch.tensor(klo, dtype=torch.int64), torch.tensor(khi, dtype=torch.int64), ids)
            # per-site sqrt(weight) of the site's own chunk: pos_weight / N_c and normal_weight / Nn_c * voxel_size^2
            pseg = seg.of_keys(enc.keys)
            off = dec_svh.offsets
            nseg_sites = torch.cat([seg.unknown_seg[off[d]:off[d] + dec_svh.num_voxels(d)] for d in range(hp.adaptive_depth)]).long()
            n_p = torch.bincount(pseg, minlength=seg.nseg).double().clamp_min(1.0)
            n_n = torch.bincount(nseg_sites, minlength=seg.nseg).double().clamp_min(1.0)
            swp = torch.sqrt(float(hp.solver.pos_weight) / n_p).float()
            swn = torch.sqrt(float(hp.solver.normal_weight) / n_n * hp.voxel_size ** 2).float()
            inputs.update(pos_weight=swp[pseg].contiguous(), normal_weight=swn[nseg_sites].contiguous(), segments=seg)
            field.segments = seg
        field.solve(fused_mode=fused_mode, **inputs)
        if self.keep_solve_inputs:
            field._solve_inputs = inputs
        if bool(hp.udf.enabled):          # models/nksr_net.py:124-130
            mask = NeuralField(svh=udf_svh, decoder=self.network.udf_decoder, features=feat.udf_features)
            mask.set_level_set(2 * hp.voxel_size)
        else:
            mask = LayerField(dec_svh, hp.adaptive_depth)
        field.set_mask_field(mask)
        field.meshing_depth = int(hp.adaptive_depth)
        field.dual_graph = self.dual_graph
        t.update({k: v for k, v in field.solve_info.items() if k.startswith('t_')})
        self.timing = t
        field.timing = t           # chunk mode solves several chunks at once: the per-field copy is the race-free one
        return field

    @torch.no_grad()       # the inference API: nothing here builds an autograd graph (training drives KernelField directly)
    def reconstruct(self, xyz, normal=None, sensor=None, detail_level=0.0, voxel_size=None, chunk_size=-1.0,
                    overlap_ratio=0.05, approx_kernel_grad=False, solver_max_iter=2000, solver_tol=1e-5,
                    fused_mode=True, preprocess_fn=None, sharded_input=False, chunk_owner=None, chunk_bounds=None):
        """``sharded_input`` / ``chunk_owner`` / ``chunk_bounds`` (chunk mode under torch.distributed only; not part of the
        reference surface): every rank passes just the points of the chunks it owns, see chunking.reconstruct_by_chunk."""
        if xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.shape[1] != 3:
            raise RuntimeError('xyz must be a float32 [N,3] tensor')
        xyz = xyz.to(self.device)
        normal = normal.to(self.device) if normal is not None else None
        sensor = sensor.to(self.device) if sensor is not None else None
        chunked = chunk_size is not None and chunk_size > 0
        if chunked and (voxel_size is not None or detail_level not in (None, 0.0)):
            raise RuntimeError('detail_level / voxel_size are not supported together with chunk_size: scale the '
                               'cloud by 0.1/voxel_size beforehand (NKSR-USAGE.md:137)')
        if chunked:
            from .chunking import reconstruct_by_chunk
            return reconstruct_by_chunk(self, xyz, normal, sensor, float(chunk_size), float(overlap_ratio),
                                        approx_kernel_grad, solver_max_iter, solver_tol, fused_mode, preprocess_fn,
                                        sharded_input=sharded_input, chunk_owner=chunk_owner, chunk_bounds=chunk_bounds)
        if preprocess_fn is not None:
            xyz, normal, sensor = preprocess_fn(xyz, normal, sensor)
        if normal is None:
            raise RuntimeError('oriented input required: pass normal=, or sensor= together with '
                               'preprocess_fn=nksr.get_estimate_normal_preprocess_fn(...)')
        if xyz.shape[0] < 8:
            raise RuntimeError('need at least 8 points to reconstruct (got %d)' % xyz.shape[0])
        scale = self._global_scale(xyz, detail_level, voxel_size)          # (non-finite input is caught by the box readback of _key_bits)
        xs = (xyz * scale).contiguous() if scale != 1.0 else xyz.contiguous()
        field = self._reconstruct_single(xs, normal.to(torch.float32).contiguous(), approx_kernel_grad, solver_max_iter,
                                         solver_tol, fused_mode)
        field.set_scale(scale)
        return field
