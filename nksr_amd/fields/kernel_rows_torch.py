"""Differentiable statement of the dense-slot kernel rows in torch ops -- the TRAINING path only (SURVEY.md section 8f-4:
``solve_non_fused`` runs under autograd at models/nksr_net.py:105-112 and the loss back-propagates into the basis features
and the interpolator weights).  The solve-time path never comes here: rows, operator, PCG and evaluation are HIP
(csrc/kfield.hip, fused.hip, pcg.hip).  What this file is for: the vector-Jacobian products  sum_r g_r . dR_r/dtheta  that the
implicit-function backward of the solve and of evaluate_f need (kernel_field._SolveFunction / _EvaluateFunction).  They are
taken by torch autograd through this statement of  R(theta)  (DESIGN.md section 2.3):

    K_d(x, c_j) = <phi_d(x), psi_j> B((x - c_j) / w_d),   phi_d(x) = t + MLP_d(t),  t = trilinear interpolation of the level's
    basis features at x,  psi_j = phi_d(c_j) = t_j + MLP_d(t_j) with t_j the voxel's own feature,  B = tensor-product quadratic
    B-spline;  gradient rows carry dB/dx and -- unless approx_kernel_grad -- the forward-mode tangent of phi through
    trilerp + MLP.

Integer decisions (containing cell, half bits, neighbour table) are the product's own (same fp32 product x * inv_w0, same
shifts; the neighbour tables are the hierarchy's), so the rows agree with nksr_kernel_rows to fp32 rounding
(tests/test_gpu_parity.py::test_torch_rows_match_the_hip_rows).
"""
import torch

from .._lib import call, ptr, stream

_OX = torch.arange(27) // 9
_OY = (torch.arange(27) // 3) % 3
_OZ = torch.arange(27) % 3


def _bspline3(u):
    """weights / derivatives of the centres at offset -1, 0, +1 for local coordinate u in [0, 1): [..., 3]"""
    um, uc = 1.0 - u, u - 0.5
    w = torch.stack([0.5 * um * um, 0.75 - uc * uc, 0.5 * u * u], -1)
    dw = torch.stack([u - 1.0, -2.0 * uc, u], -1)
    return w, dw


def _mlp_tangent(interp, t, dts):
    """phi = t + MLP(t) and its forward-mode tangents for the input tangents ``dts`` (list of [n, K]); ReLU masks are the
    constants they are almost everywhere.  The module's parameters may live on another device (``.to`` is differentiable)."""
    W1, b1, W2, b2, W3, b3 = [q.to(t.device, torch.float32) for q in (interp.W1, interp.b1, interp.W2, interp.b2, interp.W3, interp.b3)]
    a1 = t @ W1.T + b1
    m1 = (a1 > 0).to(t.dtype)
    h1 = a1 * m1
    a2 = h1 @ W2.T + b2
    m2 = (a2 > 0).to(t.dtype)
    h2 = a2 * m2
    phi = t + h2 @ W3.T + b3
    out = []
    for dt in dts:
        d1 = (dt @ W1.T) * m1
        d2 = (d1 @ W2.T) * m2
        out.append(dt + d2 @ W3.T)
    return phi, out


def rows(svh, interps, feats, xyz, grad, approx, scale=1.0):
    """Dense-slot rows of the sites ``xyz`` (model units, [n, 3]) as differentiable functions of ``feats`` (per level [n_d, K])
    and the interpolators' parameters.
    Returns (val [n, L, 27] if not grad else dval [n, 3, L, 27],  idx [n, L, 27] int64 global unknown index or -1)."""
    dev = xyz.device
    n, L = xyz.shape[0], svh.depth
    off = svh.offsets
    ox, oy, oz = _OX.to(dev), _OY.to(dev), _OZ.to(dev)
    inv_w0 = torch.tensor(svh.inv_w0, dtype=torch.float32, device=dev)
    with torch.no_grad():
        p = xyz.to(torch.float32) * inv_w0                       # ONE fp32 product decides every cell (csrc/common.h half_index)
        h0 = torch.floor(p * 2.0).to(torch.int64)
    vals, idxs = [], []
    for d in range(L):
        g = svh.level(d)
        interp = interps[d]
        feat = feats[d].to(dev, torch.float32)
        K = feat.shape[1]
        with torch.no_grad():
            hd = h0 >> d
            cell_ijk = (hd >> 1)
            hb = (hd & 1)
            u = p * (2.0 ** -d) - cell_ijk.to(torch.float32)
            ijk32 = cell_ijk.to(torch.int32).contiguous()
            keys = torch.empty(n, dtype=torch.int64, device=dev)
            call('nksr_encode_keys', ptr(ijk32), n, d, ptr(keys), stream())
            cell = g.hash.query(keys).long() if g.num_voxels else torch.full((n,), -1, dtype=torch.int64, device=dev)
            inside = cell >= 0
            nbr = torch.where(inside[:, None], g.nbr[cell.clamp(min=0)].long(), torch.full((1, 1), -1, dtype=torch.int64, device=dev)) \
                if g.num_voxels else torch.full((n, 27), -1, dtype=torch.int64, device=dev)
            have = nbr >= 0
            inv_w = float(svh.inv_w0) * 2.0 ** -d
            v = u + 0.5 - hb.to(torch.float32)                    # trilinear coordinate relative to the lower corner centre
        # trilinear interpolation of the basis features (+ spatial tangents)
        t = torch.zeros((n, K), dtype=torch.float32, device=dev)
        dts = [torch.zeros((n, K), dtype=torch.float32, device=dev) for _ in range(3)] if (grad and not approx) else []
        if g.num_voxels:
            for c in range(8):
                cx, cy, cz = c >> 2, (c >> 1) & 1, c & 1
                s = (hb[:, 0] + cx) * 9 + (hb[:, 1] + cy) * 3 + (hb[:, 2] + cz)
                j = nbr.gather(1, s[:, None])[:, 0]
                ok = (j >= 0).to(torch.float32)
                wx = v[:, 0] if cx else 1.0 - v[:, 0]
                wy = v[:, 1] if cy else 1.0 - v[:, 1]
                wz = v[:, 2] if cz else 1.0 - v[:, 2]
                fj = feat[j.clamp(min=0)] * ok[:, None]
                t = t + (wx * wy * wz)[:, None] * fj
                if dts:
                    sx, sy, sz = (1.0 if cx else -1.0), (1.0 if cy else -1.0), (1.0 if cz else -1.0)
                    dts[0] = dts[0] + (sx * wy * wz * inv_w)[:, None] * fj
                    dts[1] = dts[1] + (wx * sy * wz * inv_w)[:, None] * fj
                    dts[2] = dts[2] + (wx * wy * sz * inv_w)[:, None] * fj
        phi, dphi = _mlp_tangent(interp, t, dts)
        psi = _mlp_tangent(interp, feat, [])[0] if g.num_voxels else feat    # psi_j = phi_d(c_j): the voxel's own feature through the MLP
        bw, bd = _bspline3(u)                                        # [n, 3(axis), 3(offset)]
        bx, by, bz = bw[:, 0][:, ox], bw[:, 1][:, oy], bw[:, 2][:, oz]
        B = bx * by * bz                                             # [n, 27]
        if g.num_voxels:
            pj = psi[nbr.clamp(min=0)] * have[..., None].to(torch.float32)       # [n, 27, K]
            dot = (phi[:, None, :] * pj).sum(-1)
        else:
            pj = None
            dot = torch.zeros((n, 27), dtype=torch.float32, device=dev)
        m = have.to(torch.float32)
        if not grad:
            vals.append(scale * dot * B * m)
        else:
            dB = [bd[:, 0][:, ox] * by * bz * inv_w, bx * bd[:, 1][:, oy] * bz * inv_w, bx * by * bd[:, 2][:, oz] * inv_w]
            comp = []
            for a in range(3):
                r = dot * dB[a]
                if dphi and pj is not None:
                    r = r + (dphi[a][:, None, :] * pj).sum(-1) * B
                comp.append(scale * r * m)
            vals.append(torch.stack(comp, 1))                        # [n, 3, 27]
        idxs.append(torch.where(have, nbr + off[d], nbr))
    idx = torch.stack(idxs, 1)                                       # [n, L, 27]
    if not grad:
        return torch.stack(vals, 1), idx                             # [n, L, 27]
    return torch.stack(vals, 2), idx                                 # [n, 3, L, 27]


def apply_rows(R, idx, x, grad):
    """(R x) per row: [n] for value rows, [n, 3] for gradient rows."""
    xg = x[idx.clamp(min=0)] * (idx >= 0).to(x.dtype)                # [n, L, 27]
    if not grad:
        return (R * xg).sum((1, 2))
    return (R * xg[:, None]).sum((2, 3))
