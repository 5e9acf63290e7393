"""Oracle: kernel-field evaluation  f(x) = sum_d sum_{j in N27} alpha_j K_d(x, c_j).

Reference anchors: field.evaluate_f(xyz, grad) -> .value/.gradient
models/loss.py:189-198,222-225; sign convention f>0 inside, outward normal =
-grad f (models/loss.py:192-196, models/nksr_net.py:108).
"""
import numpy as np
from . import kernel


def evaluate_f(hier, feats, interps, psis, alpha, xyz, grad=False, approx_kernel_grad=False, batch=200000):
    vals, grads = [], []
    for s in range(0, xyz.shape[0], batch):
        x = xyz[s:s + batch]
        cols, val, dval = kernel.kernel_rows(hier, feats, interps, psis, x, grad, approx_kernel_grad, fallback=True)
        a = np.where(cols >= 0, alpha[np.maximum(cols, 0)], np.float32(0)).astype(np.float32)
        vals.append((a * val).reshape(x.shape[0], -1).sum(1, dtype=np.float64).astype(np.float32))
        if grad:
            grads.append((a[:, None] * dval).reshape(x.shape[0], 3, -1).sum(2, dtype=np.float64).astype(np.float32))
    v = np.concatenate(vals) if vals else np.zeros(0, np.float32)
    g = (np.concatenate(grads) if grads else np.zeros((0, 3), np.float32)) if grad else None
    return v, g
