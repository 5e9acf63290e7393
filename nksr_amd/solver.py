"""CSR SpMV + Jacobi-PCG wrappers (csrc/pcg.hip).  The solve behind
``KernelField.solve*`` (reference call site models/nksr_net.py:105-112; ``solver_tol``
examples/recons_waymo.py:33)."""
import ctypes as C

import torch

from ._lib import call, lib, ptr, stream


def spmv(rowptr, cols, vals, x):
    """y = A x.  ``cols`` / ``vals`` must come from KernelField.assemble (padded storage)."""
    M = rowptr.numel() - 1
    nnz = int(rowptr[M].item())
    y = torch.empty(M, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.nksr_spmv_workspace_bytes(nnz)), dtype=torch.uint8, device=x.device)
    call('nksr_spmv_plan', ptr(rowptr), M, nnz, ptr(ws), stream())
    call('nksr_spmv_csr', ptr(rowptr), ptr(cols), ptr(vals), M, nnz, ptr(x), ptr(y), ptr(ws), stream())
    return y


def pcg_solve(rowptr, cols, vals, diag, b, tol=1e-5, max_iter=2000, check_every=16, workspace=None):
    """Returns (x, iterations, relative residual).  x0 = 0, stop on ||r|| <= tol ||b||."""
    M = b.numel()
    nnz = int(rowptr[M].item())
    x = torch.empty(M, dtype=torch.float32, device=b.device)
    nbytes = int(lib.nksr_pcg_workspace_bytes(M, nnz))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = (C.c_double * 2)()
    call('nksr_pcg_solve', ptr(rowptr), ptr(cols), ptr(vals), ptr(diag), M, nnz, ptr(b), ptr(x), float(tol), int(max_iter),
         int(check_every), ptr(workspace), info, stream())
    return x, int(info[0]), float(info[1])


def profile_spmv(enable):
    """Toggle live HIP-event timing of the SpMV launches inside pcg_solve; returns the
    (milliseconds, launches) accumulated since the previous call."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    call('nksr_pcg_profile', int(bool(enable)), C.byref(ms), C.byref(n))
    return float(ms.value), int(n.value)


def csr_logical(rowptr, cols, vals):
    """Undo the SpMV's tile interleave: returns (cols, vals) in plain CSR order (length nnz).
    Test / export helper -- not on the hot path."""
    nnz = int(rowptr[-1].item())
    k = torch.arange(nnz, device=cols.device)
    m = k & 255
    phys = (k & ~255) + 4 * (m & 63) + (m >> 6)
    return cols[phys], vals[phys]
