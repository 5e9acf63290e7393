"""Oracle: normal-equation assembly + Jacobi-PCG (restates KernelField.solve*).

Reference anchors: solve_non_fused(pos_xyz, normal_xyz, normal_value, pos_weight,
normal_weight, reg_weight) models/nksr_net.py:105-112; weights
pos_weight = solver.pos_weight / N_pos, normal_weight = solver.normal_weight /
N_nrm * voxel_size**2 (models/nksr_net.py:103-111, configs/default/train.yaml:27-29);
solver_tol examples/recons_waymo.py:33.
System:  (w_p G^T G + w_n Q^T Q + reg I) alpha = w_n Q^T n.
"""
import numpy as np
import scipy.sparse as sp
from . import kernel


def assemble(hier, feats, interps, pos_xyz, normal_xyz, normal_value, pos_weight, normal_weight, reg_weight,
             approx_kernel_grad=False):
    M = hier.num_unknowns
    psis = [kernel.voxel_psi(feats[d], interps[d]) for d in range(hier.depth)]
    gc, gv, _ = kernel.kernel_rows(hier, feats, interps, psis, pos_xyz, False, approx_kernel_grad)
    G = kernel.rows_to_csr(gc, gv, M)
    qc, _, qd = kernel.kernel_rows(hier, feats, interps, psis, normal_xyz, True, approx_kernel_grad)
    nq = normal_xyz.shape[0]
    # row (j, a) of Q: stack the three components
    Q = sp.vstack([kernel.rows_to_csr(qc, qd[:, a], M) for a in range(3)]).tocsr()
    nvec = np.concatenate([normal_value[:, a] for a in range(3)]).astype(np.float32)
    G64, Q64 = G.astype(np.float64), Q.astype(np.float64)
    A = (pos_weight * (G64.T @ G64) + normal_weight * (Q64.T @ Q64) + reg_weight * sp.identity(M, format='csr')).tocsr()
    A.sort_indices()
    b = (normal_weight * (Q64.T @ nvec.astype(np.float64)))
    return A.astype(np.float32), b.astype(np.float32), G, Q, psis


def csr_spmv(indptr, indices, data, x):
    """y = A x, fp32 storage, fp32 accumulate, row by row in column order."""
    y = np.add.reduceat((data * x[indices]).astype(np.float32), indptr[:-1].astype(np.int64)) if len(data) else np.zeros(len(indptr) - 1, np.float32)
    empty = indptr[1:] == indptr[:-1]
    y = y.astype(np.float32)
    y[empty] = 0
    return y


def pcg_jacobi(A, b, tol=1e-5, max_iter=2000, fixed_iters=None):
    """Jacobi-preconditioned CG; fp32 vectors, fp64 dot products.
    Stops when ||r||_2 <= tol * ||b||_2 (checked after the residual update)."""
    indptr, indices, data = A.indptr, A.indices, A.data.astype(np.float32)
    dinv = (np.float32(1.0) / A.diagonal().astype(np.float32)).astype(np.float32)
    x = np.zeros_like(b, dtype=np.float32)
    r = b.astype(np.float32).copy()
    z = r * dinv
    p = z.copy()
    rz = float(np.dot(r.astype(np.float64), z.astype(np.float64)))
    bnorm2 = float(np.dot(b.astype(np.float64), b.astype(np.float64)))
    it = 0
    rel = 1.0
    if bnorm2 == 0.0:
        return x, 0, 0.0
    n_it = max_iter if fixed_iters is None else fixed_iters
    while it < n_it:
        Ap = csr_spmv(indptr, indices, data, p)
        pAp = float(np.dot(p.astype(np.float64), Ap.astype(np.float64)))
        alpha = np.float32(rz / pAp)
        x = x + alpha * p
        r = r - alpha * Ap
        it += 1
        rr = float(np.dot(r.astype(np.float64), r.astype(np.float64)))
        rel = (rr / bnorm2) ** 0.5
        if fixed_iters is None and rel <= tol:
            break
        z = r * dinv
        rz_new = float(np.dot(r.astype(np.float64), z.astype(np.float64)))
        beta = np.float32(rz_new / rz)
        rz = rz_new
        p = z + beta * p
    return x, it, rel
