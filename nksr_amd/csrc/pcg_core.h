// Jacobi-PCG driver shared by the assembled-CSR solve (csrc/pcg.hip) and the matrix-free fused solve
// (csrc/fused.hip): the iteration is the same, only y = A p differs.
#pragma once
#include "common.h"

struct PcgOperator {
    // enqueue y = A p on `st`; `done` (device flag) != 0 must turn the launches into no-ops.  seg_done (may be NULL): device
    // flags, segment c's at seg_done[c * seg_stride] -- an operator that knows its segments (nksr_fused_op_t.item_seg /
    // unknown_seg) may skip the rows and unknowns of finished ones (their y is never read again); seg_done_count (device, may be
    // NULL): how many segments have finished, seg_count: how many there are
    virtual int apply(const float* p, float* y, const int* done, const int* seg_done, int seg_stride, const int* seg_done_count, int seg_count,
                      hipStream_t st) = 0;
    // bytes one application moves: the algorithmic minimum of THIS operator's data (what `roofline.achieved` is priced on), what the
    // layout really streams, and the figure of SURVEY.md section 8d's formula (CSR: equal to the first)
    virtual void bytes(double* algorithmic, double* physical, double* survey_formula) = 0;
    virtual ~PcgOperator() {}
};

// bytes of the vector workspace (r, z, p, y, partial sums, scalars) for M unknowns
size_t nksr_pcg_vector_bytes(int32_t M);
// x0 = 0, stop on ||r|| <= tol ||b||; info_out (3 doubles): [0] = iterations, [1] = relative residual, [2] = Jacobi fallbacks.  Syncs every check_every iterations.
// Preconditioner: Jacobi (diag); with `pc` the unknowns of the coarse levels (the last pc->n) get pc->steps Chebyshev steps on their
// diagonal block instead (see nksr_coarse_precond_t, include/nksr_hip.h).  `seg`: independent diagonal blocks with their own CG
// scalars (nksr_segments_t); the workspace is then nksr_pcg_vector_workspace_bytes_seg.
int nksr_pcg_run(PcgOperator& A, const float* diag, int32_t M, const float* b, float* x, float tol, int max_iter, int check_every,
                 void* vector_workspace, double* info_out, hipStream_t st, const nksr_coarse_precond_t* pc = nullptr,
                 const nksr_segments_t* seg = nullptr);
