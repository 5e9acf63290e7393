// Jacobi-PCG driver shared by the assembled-CSR solve (csrc/pcg.hip) and the matrix-free fused solve
// (csrc/fused.hip): the iteration is the same, only y = A p differs.
#pragma once
#include "common.h"

struct PcgOperator {
    // enqueue y = A p on `st`; `done` (device flag) != 0 must turn the launches into no-ops
    virtual int apply(const float* p, float* y, const int* done, hipStream_t st) = 0;
    // bytes one application moves: algorithmic figure and what the layout really streams
    virtual void bytes(double* algorithmic, double* physical) = 0;
    virtual ~PcgOperator() {}
};

// bytes of the vector workspace (r, z, p, y, partial sums, scalars) for M unknowns
size_t nksr_pcg_vector_bytes(int32_t M);
// x0 = 0, stop on ||r|| <= tol ||b||; info_out[0] = iterations, [1] = relative residual.  Syncs every check_every iterations.
// Preconditioner: Jacobi (diag); with `pc` the unknowns of the coarse levels (the last pc->n) get pc->steps Chebyshev steps on their
// diagonal block instead (see nksr_coarse_precond_t, include/nksr_hip.h).
int nksr_pcg_run(PcgOperator& A, const float* diag, int32_t M, const float* b, float* x, float tol, int max_iter, int check_every,
                 void* vector_workspace, double* info_out, hipStream_t st, const nksr_coarse_precond_t* pc = nullptr);
