"""CSR SpMV + Jacobi-PCG wrappers (csrc/pcg.hip).  The solve behind
``KernelField.solve*`` (reference call site models/nksr_net.py:105-112; ``solver_tol``
examples/recons_waymo.py:33)."""
import ctypes as C

import torch

from ._lib import call, lib, ptr, stream


last_fallbacks = 0      # Jacobi fallbacks of the last pcg_solve (segments whose coarse-level block lost definiteness)


def col_format(cols):
    """Physical layout of a matrix from KernelField.assemble (include/nksr_hip.h, col_format): int32 columns in
    256-entry tiles (0) or three 21-bit columns per int64 word in 192-entry tiles (1)."""
    return 1 if cols.dtype == torch.int64 else 0


def spmv(rowptr, cols, vals, x):
    """y = A x.  ``cols`` / ``vals`` must come from KernelField.assemble (padded storage)."""
    M = rowptr.numel() - 1
    nnz = int(rowptr[M].item())
    fmt = col_format(cols)
    y = torch.empty(M, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.nksr_spmv_workspace_bytes(nnz)), dtype=torch.uint8, device=x.device)
    call('nksr_spmv_plan', ptr(rowptr), M, nnz, fmt, ptr(ws), stream())
    call('nksr_spmv_csr', ptr(rowptr), ptr(cols), ptr(vals), M, nnz, fmt, ptr(x), ptr(y), ptr(ws), stream())
    return y


def pcg_solve(rowptr, cols, vals, diag, b, tol=1e-5, max_iter=2000, check_every=16, workspace=None, precond=None):
    """Returns (x, iterations, relative residual).  x0 = 0, stop on ||r|| <= tol ||b||.  ``precond``: a CoarsePrecondT
    (KernelField._coarse_precond) -- Chebyshev steps on the coarse levels' diagonal block instead of Jacobi there; if that
    polynomial loses definiteness (r.z <= 0) the solve restarts with Jacobi alone on the device (``last_fallbacks`` counts it)."""
    global last_fallbacks
    M = b.numel()
    nnz = int(rowptr[M].item())
    x = torch.empty(M, dtype=torch.float32, device=b.device)
    nbytes = int(lib.nksr_pcg_workspace_bytes(M, nnz))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=b.device)
    info = (C.c_double * 3)()
    call('nksr_pcg_solve', ptr(rowptr), ptr(cols), ptr(vals), ptr(diag), M, nnz, col_format(cols), ptr(b), ptr(x), float(tol), int(max_iter),
         int(check_every), ptr(workspace), C.byref(precond) if precond is not None else None, info, stream())
    last_fallbacks = int(info[2])
    if info[1] < 0:
        raise RuntimeError('PCG breakdown (r.z <= 0 with the Jacobi preconditioner): the system is not positive definite')
    return x, int(info[0]), float(info[1])


def profile_spmv(enable):
    """Toggle live HIP-event timing of the SpMV launches inside pcg_solve; returns the
    (milliseconds, launches) accumulated since the previous call."""
    ms, n = C.c_double(0.0), C.c_int64(0)
    call('nksr_pcg_profile', int(bool(enable)), C.byref(ms), C.byref(n))
    return float(ms.value), int(n.value)


def profile_spmv_samples():
    """Milliseconds of every application behind the totals the last profile_spmv() call returned."""
    n = int(lib.nksr_pcg_profile_samples(None, 0))
    buf = (C.c_float * max(n, 1))()
    lib.nksr_pcg_profile_samples(C.cast(buf, C.c_void_p), n)
    return [float(buf[i]) for i in range(n)]


def profile_spmv_bytes():
    """(algorithmic, physical, SURVEY-formula) bytes of the operator applications timed since the last call."""
    a, b = C.c_double(0.0), C.c_double(0.0)
    call('nksr_pcg_profile_bytes', C.byref(a), C.byref(b))
    return float(a.value), float(b.value), float(lib.nksr_pcg_profile_survey_bytes())


def csr_logical(rowptr, cols, vals):
    """Undo the SpMV's tile interleave: returns (cols, vals) in plain CSR order (length nnz).
    Test / export helper -- not on the hot path."""
    nnz = int(rowptr[-1].item())
    k = torch.arange(nnz, device=cols.device)
    if col_format(cols) == 0:
        m = k & 255
        phys = (k & ~255) + 4 * (m & 63) + (m >> 6)
        return cols[phys], vals[phys]
    t, m = k // 192, k % 192
    phys = t * 192 + 3 * (m & 63) + (m >> 6)
    word = cols[phys // 3]
    c = (word >> (21 * (phys % 3))) & 0x1FFFFF
    return c.to(torch.int32), vals[phys]
