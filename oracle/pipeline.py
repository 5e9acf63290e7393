"""Oracle: the reconstruct() -> extract_dual_mesh() call sequence on CPU.

Restates, step for step, what nksr_amd.Reconstructor does (which mirrors the reference call
sequence examples/recons_waymo_cpu.py:48-63 / examples/recons_simple.py:25-27 and the solver
weights of models/nksr_net.py:103-112).  Used (a) as the checker of the end-to-end GPU parity
tests and (b) as bench.py's ``cpu_baseline`` ("port": the reference CPU path itself is the
absent wheel).  numpy + scipy.sparse only.
"""
import time

import numpy as np

from . import field, hierarchy, kernel, meshing, solve, spec


def default_interpolators(depth, K=4, H=16, seed=0, init_scale=0.0):
    """Same construction as nksr_amd.nn.network.Interpolator (torch.Generator seeded)."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(depth):
        W1 = torch.randn(H, K, generator=gen) / K ** 0.5
        W2 = torch.randn(H, H, generator=gen) / H ** 0.5
        W3 = float(init_scale) * torch.randn(K, H, generator=gen) / H ** 0.5
        out.append(kernel.Interpolator(W1.numpy(), np.zeros(H), W2.numpy(), np.zeros(H), W3.numpy(), np.zeros(K)))
    return out


def splat_trilinear(level, xyz, feat, voxel_size_d, return_weights=False):
    """Hat-function splat of per-point features onto the voxels of one level."""
    inv_w = np.float32(spec.inv_w0_f32(voxel_size_d))
    p = xyz.astype(np.float32) * inv_w
    base = np.floor(p - np.float32(0.5)).astype(np.int32)
    acc = np.zeros((level.n, feat.shape[1]), np.float64)
    ws = np.zeros(level.n, np.float64)
    for co in spec.CORNER_OFFSETS:
        ijk = base + co[None]
        w = np.prod(np.float32(1.0) - np.abs(p - (ijk.astype(np.float32) + np.float32(0.5))), axis=1)
        j = level.lookup(ijk)
        ok = (j >= 0) & (w > 0)
        np.add.at(acc, j[ok], feat[ok].astype(np.float64) * w[ok, None])
        np.add.at(ws, j[ok], w[ok])
    if return_weights:
        return acc.astype(np.float32), ws.astype(np.float32)
    return acc.astype(np.float32)


def reconstruct(xyz, normal, voxel_size=0.1, depth=4, adaptive_depth=1, kernel_dim=4, hidden=16, pos_weight=1e4,
                normal_weight=1e4, reg_weight=1.0, tol=1e-5, max_iter=2000, approx_kernel_grad=False, interps=None,
                feats=None, timing=None, net_params=None, udf=False):
    """xyz already in model units (finest voxel = voxel_size).  Returns a dict field.
    ``net_params`` (oracle.network.export_params): run the full encoder / U-Net restatement; without
    it the analytic branch is used (identical results while the residual heads are zero)."""
    t0 = time.perf_counter()
    net_normals = udf_feats = None
    if net_params is not None:
        from . import network as onet
        hier, feats, net_normals, _, _ = onet.forward(net_params, xyz, normal, voxel_size, depth, kernel_dim, adaptive_depth, udf=udf)
        udf_feats = onet.forward.last_udf
        if interps is None:
            P = net_params
            interps = [kernel.Interpolator(*[P['interpolators.%d.%s' % (d, k)] for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')])
                       for d in range(depth)]
    else:
        hier = hierarchy.Hierarchy(voxel_size, depth).build_point_neighborhood(xyz)
    if feats is None:
        feats = []
        for L in hier.levels:
            b = np.zeros((L.n, kernel_dim), np.float32)
            b[:, 0] = 1
            feats.append(b)
    if interps is None:
        interps = default_interpolators(depth, kernel_dim, hidden)
    nxyz, nval = [], []
    for d in range(adaptive_depth):
        L = hier.levels[d]
        if net_normals is not None:
            nf = net_normals[d]
        else:
            # unit mean normal; a cancelled splat (|sum w n| < 1e-2 sum w) or a barely touched voxel (sum w < 1e-3) stays short
            # (oracle/network.py NORMAL_MIN_LENGTH / NORMAL_MIN_WEIGHT)
            s, ws = splat_trilinear(L, xyz, normal, voxel_size * (1 << d), return_weights=True)
            nf = s / np.maximum(np.maximum(np.linalg.norm(s, axis=1), np.float32(1e-2) * ws), np.float32(1e-3))[:, None]
        nxyz.append(L.centers())
        nval.append(nf.astype(np.float32))
    nxyz, nval = np.concatenate(nxyz), np.concatenate(nval)
    t1 = time.perf_counter()
    A, b, G, Q, psis = solve.assemble(hier, feats, interps, xyz, nxyz, -nval, pos_weight / xyz.shape[0],
                                      normal_weight / nxyz.shape[0] * voxel_size ** 2, reg_weight, approx_kernel_grad)
    t2 = time.perf_counter()
    alpha, iters, rel = solve.pcg_jacobi(A, b, tol, max_iter)
    t3 = time.perf_counter()
    if timing is not None:
        timing.update({'hierarchy+features': t1 - t0, 'assemble': t2 - t1, 'pcg': t3 - t2, 'iters': iters,
                       'M': A.shape[0], 'nnz': A.nnz})
    return {'hier': hier, 'feats': feats, 'interps': interps, 'psis': psis, 'alpha': alpha, 'A': A, 'b': b,
            'iters': iters, 'rel': rel, 'normal_xyz': nxyz, 'normal_value': nval,
            'approx_kernel_grad': approx_kernel_grad, 'voxel_size': voxel_size, 'udf_feats': udf_feats,
            'adaptive_depth': adaptive_depth}


def evaluate(fld, xyz, grad=False):
    return field.evaluate_f(fld['hier'], fld['feats'], fld['interps'], fld['psis'], fld['alpha'], xyz, grad,
                            fld['approx_kernel_grad'])


def extract_dual_mesh(fld, mise_iter=0, grid_upsample=1, info=None, dual_graph='lattice'):
    """dual_graph = 'adaptive': cells as large as their level (oracle/dual_adaptive.py) instead of one uniform lattice."""
    mask_fn = None
    if fld.get('udf_feats') is not None:      # NeuralField mask, level set 2 * voxel_size (models/nksr_net.py:130)
        from . import network as onet
        mask_fn = lambda p: onet.udf_decode(fld['hier'], fld['udf_feats'], p) < np.float32(fld.get('udf_level_set', 2 * fld['voxel_size']))
    if dual_graph == 'adaptive':
        from . import dual_adaptive
        return dual_adaptive.extract(fld['voxel_size'], [L.ijk for L in fld['hier'].levels[:fld.get('adaptive_depth', 1)]],
                                     lambda p: evaluate(fld, p)[0], mise_iter, grid_upsample, mask_fn=mask_fn, info=info)
    return meshing.extract(fld['voxel_size'], fld['hier'].levels[0], lambda p: evaluate(fld, p)[0], mise_iter, grid_upsample,
                           mask_fn=mask_fn, info=info, coarser=fld['hier'].levels[1:fld.get('adaptive_depth', 1)])
