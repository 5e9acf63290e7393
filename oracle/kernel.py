"""Oracle: neural-kernel evaluation (restates nksr.fields.KernelField internals).

Spec.  For level d with voxel width w_d:
    K_d(x, c_j) = < phi_d(x), psi_j > * B((x - c_j) / w_d)
    phi_d(x)    = t + MLP_d(t),  t = trilinear interpolation of basis_features[d] at x
    psi_j       = phi_d(c_j)     (= feat_j + MLP_d(feat_j))
    B           = tensor-product quadratic B-spline (27 supporting voxels per level)
Reference anchors: KernelField(svh, interpolator, features, approx_kernel_grad)
models/nksr_net.py:91-96; evaluate_f(xyz, grad) models/loss.py:189-198,225;
interpolator hyper-parameters configs/default/train.yaml:23-25.
"""
import numpy as np
from . import spec


class Interpolator:
    """Per-level MLP  K -> H -> H -> K  with ReLU and a residual skip."""

    def __init__(self, W1, b1, W2, b2, W3, b3):
        self.W1, self.b1, self.W2, self.b2, self.W3, self.b3 = [np.asarray(a, np.float32) for a in (W1, b1, W2, b2, W3, b3)]

    def forward(self, t, Jt=None):
        h1p = t @ self.W1.T + self.b1
        h1 = np.maximum(h1p, 0)
        h2p = h1 @ self.W2.T + self.b2
        h2 = np.maximum(h2p, 0)
        phi = t + h2 @ self.W3.T + self.b3
        if Jt is None:
            return phi.astype(np.float32), None
        # Jt: [n, K, 3]
        d1 = np.einsum('hk,nka->nha', self.W1, Jt) * (h1p > 0)[:, :, None]
        d2 = np.einsum('gh,nha->nga', self.W2, d1) * (h2p > 0)[:, :, None]
        J = Jt + np.einsum('kg,nga->nka', self.W3, d2)
        return phi.astype(np.float32), J.astype(np.float32)


def voxel_psi(feat_d, interp_d):
    """psi_j = phi_d(c_j): the trilinear stencil collapses onto voxel j itself."""
    return interp_d.forward(np.asarray(feat_d, np.float32))[0]


def _nbr(L, ijk_cell, cell, s, o, fallback):
    """Neighbour slot s of every site's cell: table for active cells, coordinate lookup otherwise."""
    ok = cell >= 0
    j = np.where(ok, L.nbr[np.where(ok, cell, 0), s], -1)
    if fallback and (~ok).any():
        j = np.where(ok, j, L.lookup(ijk_cell + o[None]))
    return j


def site_features(hier, feats, interps, xyz, need_jac, fallback=False):
    """Per level: (cell, u, phi [n,K], Jphi [n,K,3] in world units or None)."""
    out = []
    H0, _ = spec.half_index(xyz, hier.voxel_size)
    for d, (cell, u, hb) in enumerate(hier.site_cells(xyz)):
        L = hier.levels[d]
        Icell = ((H0 >> d) >> 1).astype(np.int32)
        n = xyz.shape[0]
        K = feats[d].shape[1]
        inv_w = np.float32(spec.inv_w0_f32(hier.voxel_size) * np.float32(2.0 ** (-d)))
        v = (u + np.float32(0.5) - hb.astype(np.float32)).astype(np.float32)
        t = np.zeros((n, K), np.float32)
        Jt = np.zeros((n, K, 3), np.float32) if need_jac else None
        ok = cell >= 0
        cs = np.where(ok, cell, 0)
        for c, co in enumerate(spec.CORNER_OFFSETS):
            o = hb - 1 + co[None, :]
            s = (o[:, 0] + 1) * 9 + (o[:, 1] + 1) * 3 + (o[:, 2] + 1)
            j = L.nbr[cs, s]
            j = np.where(ok, j, -1)
            if fallback and (~ok).any():
                j = np.where(ok, j, L.lookup(Icell + o))
            f = np.where((j >= 0)[:, None], feats[d][np.maximum(j, 0)], np.float32(0)).astype(np.float32)
            wa = np.where(co[None, :] == 1, v, np.float32(1.0) - v).astype(np.float32)
            w = wa[:, 0] * wa[:, 1] * wa[:, 2]
            t += f * w[:, None]
            if need_jac:
                sg = np.where(co == 1, np.float32(1.0), np.float32(-1.0)).astype(np.float32)
                Jt[:, :, 0] += f * (sg[0] * wa[:, 1] * wa[:, 2] * inv_w)[:, None]
                Jt[:, :, 1] += f * (wa[:, 0] * sg[1] * wa[:, 2] * inv_w)[:, None]
                Jt[:, :, 2] += f * (wa[:, 0] * wa[:, 1] * sg[2] * inv_w)[:, None]
        phi, J = interps[d].forward(t, Jt)
        out.append((cell, u, phi, J, inv_w, Icell))
    return out


def kernel_rows(hier, feats, interps, psis, xyz, grad, approx_kernel_grad, fallback=False):
    """Dense-slot rows of the kernel matrix at the sites ``xyz``.

    Returns cols [n, L, 27] (global unknown index or -1), val [n, L, 27] and, when
    ``grad``, dval [n, 3, L, 27] = d/dx_a K(x, c_j).
    """
    n = xyz.shape[0]
    Lv = hier.depth
    cols = np.full((n, Lv, 27), -1, np.int64)
    val = np.zeros((n, Lv, 27), np.float32)
    dval = np.zeros((n, 3, Lv, 27), np.float32) if grad else None
    sf = site_features(hier, feats, interps, xyz, need_jac=grad and not approx_kernel_grad, fallback=fallback)
    for d, (cell, u, phi, J, inv_w, Icell) in enumerate(sf):
        L = hier.levels[d]
        ok = cell >= 0
        cs = np.where(ok, cell, 0)
        bw = [spec.bspline3(u[:, a]) for a in range(3)]
        for s, o in enumerate(spec.NBR_OFFSETS):
            j = _nbr(L, Icell, cell, s, o, fallback)
            present = j >= 0
            psi = psis[d][np.maximum(j, 0)]
            dot = np.einsum('nk,nk->n', phi, psi).astype(np.float32)
            bx, by, bz = bw[0][0][:, o[0] + 1], bw[1][0][:, o[1] + 1], bw[2][0][:, o[2] + 1]
            B = bx * by * bz
            cols[:, d, s] = np.where(present, j + hier.offsets[d], -1)
            val[:, d, s] = np.where(present, dot * B, np.float32(0))
            if grad:
                dbx, dby, dbz = bw[0][1][:, o[0] + 1], bw[1][1][:, o[1] + 1], bw[2][1][:, o[2] + 1]
                dB = [dbx * by * bz * inv_w, bx * dby * bz * inv_w, bx * by * dbz * inv_w]
                for a in range(3):
                    g = dot * dB[a]
                    if J is not None:
                        g = g + np.einsum('nk,nk->n', J[:, :, a], psi).astype(np.float32) * B
                    dval[:, a, d, s] = np.where(present, g, np.float32(0))
    return cols, val, dval


def rows_to_csr(cols, val, M):
    """[n, slots] dense-slot rows -> scipy CSR (explicit zeros of absent slots dropped)."""
    import scipy.sparse as sp
    n = cols.shape[0]
    c = cols.reshape(n, -1)
    v = val.reshape(n, -1)
    m = c >= 0
    rows = np.repeat(np.arange(n), m.sum(1))
    return sp.csr_matrix((v[m].astype(np.float32), (rows, c[m])), shape=(n, M), dtype=np.float32)
