// Matrix-free ("fused") normal-equation operator  y = (sum_s R_s^T R_s + reg I) x  and its Jacobi-PCG solve.
// Reference: reconstruct(..., fused_mode=True) (examples/recons_waymo.py:33, recons_waymo_cpu.py:58, gis_app.py:40) -- the
// memory-lean solve that never materialises the system matrix; KernelField.solve (the assembled twin is solve_non_fused,
// models/nksr_net.py:105-112).  R_s are the dense-slot kernel rows of a site set (G: one row per input point, Q: three
// gradient rows per normal site), already multiplied by sqrt(weight), stored LEVEL-MAJOR: rows[d][r][27].
//
// Every site of a level-d cell c couples to the same 27 voxels (c's stencil), so both products run cell by cell:
//   forward     t_d[r]   = sum_s rows[d][r][s] * x[nbr_d[c][s]]        (x stencil of the cell loaded once, 27 lanes)
//   transposed  P[c][s]  = sum_{r in c} rows[d][r][s] * t[r],   y_j = reg x_j + sum_{s'} P[nbr_d[j][s']][26 - s']
// with t = sum_d t_d.  Work items = (set, level, cell, <= 32 consecutive rows); one item per 32-lane half of a wavefront
// (27 lanes active), so fine cells (a handful of rows) and coarse cells (thousands of rows, cut into many items) balance.
// Both passes read every row once, coalesced (the rows of a cell are contiguous: sites are Morton-sorted): 2 x 4 bytes
// per dense slot per application and no column indices at all -- HBM-bound.  Fixed summation orders, no float atomics:
// deterministic.  No assembly: the solve starts right after the kernel rows.
#include "common.h"
#include "pcg_core.h"
#include <stdlib.h>

#define FZ_RC 32
#define FZ_BLOCK 256
#define FZ_MAX_SETS 2
#define FZ_DEFAULT_VARIANT 0

struct FusedArgs {
    nksr_hier_t hier;
    nksr_fused_set_t sets[FZ_MAX_SETS];
    int nsets;
    int M;
    int64_t row_off[FZ_MAX_SETS];                     // first row of the set in the concatenated t vector
    int64_t rows_total;
    int32_t lin_base[FZ_MAX_SETS][NKSR_MAX_DEPTH];    // index of (set, level, cell 0) in the per-cell item offsets
    int32_t lin_total;
};

static int fz_args(FusedArgs& A, const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 1 || nsets > FZ_MAX_SETS) return nksr_set_error(NKSR_ERR_ARG, "1..%d site sets", FZ_MAX_SETS);
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    A.nsets = nsets;
    A.M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int64_t lin = 0, rows = 0;
    for (int s = 0; s < nsets; ++s) {
        if (sets[s].ncomp != 1 && sets[s].ncomp != 3) return nksr_set_error(NKSR_ERR_ARG, "ncomp must be 1 or 3");
        if (sets[s].n * sets[s].ncomp >= ((int64_t)1 << 31)) return nksr_set_error(NKSR_ERR_CAPACITY, "site set too large");
        A.sets[s] = sets[s];
        A.row_off[s] = rows;
        rows += sets[s].n * sets[s].ncomp;
        for (int d = 0; d < h->depth; ++d) { A.lin_base[s][d] = (int32_t)lin; lin += h->lv[d].n; }
    }
    if (lin >= ((int64_t)1 << 31) - 1) return nksr_set_error(NKSR_ERR_CAPACITY, "too many cells");
    A.rows_total = rows;
    A.lin_total = (int32_t)lin;
    return NKSR_OK;
}

__device__ __forceinline__ void fz_decode_lin(const FusedArgs& A, int lin, int& set, int& d, int& c) {
    set = 0; d = 0;
    for (int s = 0; s < A.nsets; ++s)
        for (int l = 0; l < A.hier.depth; ++l)
            if (lin >= A.lin_base[s][l]) { set = s; d = l; }
    c = lin - A.lin_base[set][d];
}

__global__ void k_fz_item_counts(FusedArgs A, int32_t* __restrict__ counts) {
    const int lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin > A.lin_total) return;
    if (lin == A.lin_total) { counts[lin] = 0; return; }
    int set, d, c;
    fz_decode_lin(A, lin, set, d, c);
    const nksr_fused_set_t& S = A.sets[set];
    const int nrows = (S.end[d][c] - S.start[d][c]) * S.ncomp;
    counts[lin] = (nrows + FZ_RC - 1) / FZ_RC;
}

__global__ void k_fz_item_fill(FusedArgs A, const int32_t* __restrict__ offsets, int4* __restrict__ items) {
    const int lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= A.lin_total) return;
    int set, d, c;
    fz_decode_lin(A, lin, set, d, c);
    const nksr_fused_set_t& S = A.sets[set];
    const int r0 = S.start[d][c] * S.ncomp, r1 = S.end[d][c] * S.ncomp;
    int it = offsets[lin];
    for (int r = r0; r < r1; r += FZ_RC, ++it) items[it] = make_int4(set * 8 + d, c, r, r + FZ_RC < r1 ? r + FZ_RC : r1);
}

__device__ __forceinline__ float half_sum(float p) {      // sum over the 32 lanes of this half-wave, fixed tree
    p += __shfl_xor(p, 16, 32);
    p += __shfl_xor(p, 8, 32);
    p += __shfl_xor(p, 4, 32);
    p += __shfl_xor(p, 2, 32);
    p += __shfl_xor(p, 1, 32);
    return p;
}

// t_d[r] = sum_s rows[d][r][s] * x[stencil of the item's cell][s]
// Every half-wave carries FZ_ILP consecutive items at once: a fine cell holds only a handful of rows, so one item per
// half-wave is a chain of three dependent loads (item -> neighbour table -> x) in front of two or three row loads, and the
// kernel ran at the latency of that chain (measured: 700 us for 1.9 GB).  Consecutive items are neighbouring cells of the
// same set and level (similar row counts), so their chains and row loads overlap.
template <int ILP, int RU>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_forward(FusedArgs A, const int4* __restrict__ items, int nitems,
                                                        const float* __restrict__ x, float* __restrict__ tpart,
                                                        const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * ILP;
    if (i0 >= nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[ILP];
    int nb[ILP], nrows[ILP], maxrows = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) it[k] = items[i0 + k < nitems ? i0 + k : nitems - 1];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        nb[k] = act ? A.hier.lv[it[k].x & 7].nbr[(int64_t)it[k].y * 27 + s] : -1;
        nrows[k] = i0 + k < nitems ? it[k].w - it[k].z : 0;
        maxrows = nrows[k] > maxrows ? nrows[k] : maxrows;
    }
    float xs[ILP];
    const float* base[ILP];
    float* tp[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        const int set = it[k].x >> 3, d = it[k].x & 7;
        const nksr_fused_set_t& S = A.sets[set];
        xs[k] = nb[k] >= 0 ? x[A.hier.lv[d].offset + nb[k]] : 0.f;
        base[k] = S.rows + ((int64_t)d * (S.n * S.ncomp) + it[k].z) * 27 + (act ? s : 0);
        tp[k] = tpart + (int64_t)d * A.rows_total + A.row_off[set] + it[k].z;
    }
    for (int j = 0; j < maxrows; j += RU) {
        float v[ILP][RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) v[k][u] = (act && j + u < nrows[k]) ? base[k][(int64_t)(j + u) * 27] : 0.f;
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const float p = half_sum(v[k][u] * xs[k]);
                if (s == 0 && j + u < nrows[k]) tp[k][j + u] = p;
            }
    }
}

__global__ void k_fz_tsum(int depth, int64_t rows_total, const float* __restrict__ tpart, float* __restrict__ t,
                          const int* __restrict__ done) {
    if (done && *done) return;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_total) return;
    float a = 0.f;
    for (int d = 0; d < depth; ++d) a += tpart[(int64_t)d * rows_total + r];
    t[r] = a;
}

// P[item][s] = sum_{r in item} rows[d][r][s] * w[r];  MODE 0: w = t (operator), 1: w = target (right-hand side), 2: w = the row value itself (diagonal)
template <int MODE, int ILP, int RU>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_transposed(FusedArgs A, const int4* __restrict__ items, int nitems,
                                                           const float* __restrict__ t, float* __restrict__ part,
                                                           const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * ILP;
    if (i0 >= nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[ILP];
    int nrows[ILP], maxrows = 0;
    const float* base[ILP];
    const float* w[ILP];
    float acc[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) it[k] = items[i0 + k < nitems ? i0 + k : nitems - 1];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        const int set = it[k].x >> 3, d = it[k].x & 7;
        const nksr_fused_set_t& S = A.sets[set];
        base[k] = S.rows + ((int64_t)d * (S.n * S.ncomp) + it[k].z) * 27 + (act ? s : 0);
        w[k] = MODE == 0 ? t + A.row_off[set] + it[k].z : (MODE == 1 ? (S.target ? S.target + it[k].z : nullptr) : nullptr);
        nrows[k] = i0 + k < nitems ? it[k].w - it[k].z : 0;
        if (MODE == 1 && w[k] == nullptr) nrows[k] = 0;          // a set without targets adds nothing to the right-hand side
        maxrows = nrows[k] > maxrows ? nrows[k] : maxrows;
        acc[k] = 0.f;
    }
    for (int j = 0; j < maxrows; j += RU) {
        float v[ILP][RU], wk[ILP][RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                const bool live = j + u < nrows[k];
                v[k][u] = live ? base[k][(int64_t)(j + u) * 27] : 0.f;
                wk[k][u] = (MODE != 2 && live) ? w[k][j + u] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) acc[k] = fmaf(v[k][u], MODE == 2 ? v[k][u] : wk[k][u], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < ILP; ++k)
        if (i0 + k < nitems) part[(int64_t)(i0 + k) * 32 + s] = act ? acc[k] : 0.f;
}

// y_j = (MODE 0: reg x_j, 1: 0, 2: reg) + sum over the 27 neighbour cells c of j, sets, items of c:  P[item][26 - s']
// One half-wave per unknown, lane = neighbour slot: the 27 (cell -> items -> block entry) chains run side by side (a thread per
// unknown walking them one after the other took 1.4 ms); fixed tree reduction.
template <int MODE>
__global__ void __launch_bounds__(256) k_fz_gather(FusedArgs A, const int32_t* __restrict__ offsets, const float* __restrict__ part,
                                                  const float* __restrict__ x, float reg, float* __restrict__ y,
                                                  const int* __restrict__ done) {
    if (done && *done) return;
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 5;
    if (j >= A.M) return;
    const int sp = threadIdx.x & 31;
    int d = 0;
    while (d + 1 < A.hier.depth && j >= A.hier.lv[d + 1].offset) ++d;
    const nksr_level_t& lv = A.hier.lv[d];
    const int i = j - lv.offset;
    float acc = 0.f;
    const int c = sp < 27 ? lv.nbr[(int64_t)i * 27 + sp] : -1;
    if (c >= 0) {
        int i0[FZ_MAX_SETS], i1[FZ_MAX_SETS];
#pragma unroll
        for (int set = 0; set < FZ_MAX_SETS; ++set) {
            i0[set] = i1[set] = 0;
            if (set < A.nsets) {
                const int lin = A.lin_base[set][d] + c;
                i0[set] = offsets[lin];
                i1[set] = offsets[lin + 1];
            }
        }
#pragma unroll
        for (int set = 0; set < FZ_MAX_SETS; ++set)
            for (int itx = i0[set]; itx < i1[set]; ++itx) acc += part[(int64_t)itx * 32 + (26 - sp)];
    }
    acc = half_sum(acc);
    if (sp == 0) y[j] = acc + (MODE == 0 ? reg * x[j] : (MODE == 2 ? reg : 0.f));
}

struct FusedWork {
    float* tpart;   // [L][rows_total]
    float* t;       // [rows_total]
    float* part;    // [nitems][32]
};
static size_t fz_align(size_t v) { return (v + 255) / 256 * 256; }
static FusedWork fz_carve(void* ws, const FusedArgs& A, int64_t nitems) {
    FusedWork w;
    char* p = (char*)ws;
    w.tpart = (float*)p; p += fz_align((size_t)A.hier.depth * A.rows_total * sizeof(float));
    w.t = (float*)p; p += fz_align((size_t)A.rows_total * sizeof(float));
    w.part = (float*)p;
    (void)nitems;
    return w;
}

extern "C" size_t nksr_fused_workspace_bytes(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, int64_t nitems) {
    FusedArgs A;
    if (fz_args(A, h, sets, nsets)) return 0;
    return fz_align((size_t)A.hier.depth * A.rows_total * sizeof(float)) + fz_align((size_t)A.rows_total * sizeof(float)) +
           fz_align((size_t)nitems * 32 * sizeof(float)) + 256;
}

extern "C" int64_t nksr_fused_cells(const nksr_hier_t* h, int nsets) {
    int64_t n = 0;
    for (int d = 0; d < h->depth; ++d) n += h->lv[d].n;
    return n * nsets;
}

extern "C" int nksr_fused_item_counts(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, int32_t* counts_out, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    hipLaunchKernelGGL(k_fz_item_counts, dim3(nksr_blocks((int64_t)A.lin_total + 1, 256)), dim3(256), 0, (hipStream_t)stream, A, counts_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_items(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, int32_t* items_out,
                                void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    if (A.lin_total > 0) {
        hipLaunchKernelGGL(k_fz_item_fill, dim3(nksr_blocks(A.lin_total, 256)), dim3(256), 0, (hipStream_t)stream, A, offsets, (int4*)items_out);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}

// (items per half-wave, rows per trip): probe variants, NKSR_FZ_VARIANT = 0..3; the default is the fastest one measured
static int g_fz_variant = -1;
static int fz_variant() {
    if (g_fz_variant < 0) {
        const char* e = getenv("NKSR_FZ_VARIANT");
        g_fz_variant = e ? atoi(e) : FZ_DEFAULT_VARIANT;
        if (g_fz_variant < 0 || g_fz_variant > 3) g_fz_variant = FZ_DEFAULT_VARIANT;
    }
    return g_fz_variant;
}
static dim3 fz_grid(int64_t nitems, int ilp) { return dim3(nksr_blocks((nitems + ilp - 1) / ilp * 32, FZ_BLOCK)); }

static int fz_apply(const FusedArgs& A, const int32_t* offsets, const int4* items, int nitems, float reg, const FusedWork& w,
                    const float* x, float* y, const int* done, hipStream_t st) {
    if (nitems > 0) {
        const dim3 blk(FZ_BLOCK), gs(nksr_blocks(A.rows_total, 256));
#define FZ_APPLY(I, R)                                                                                                          \
        hipLaunchKernelGGL((k_fz_forward<I, R>), fz_grid(nitems, I), blk, 0, st, A, items, nitems, x, w.tpart, done);          \
        hipLaunchKernelGGL(k_fz_tsum, gs, dim3(256), 0, st, A.hier.depth, A.rows_total, (const float*)w.tpart, w.t, done);     \
        hipLaunchKernelGGL((k_fz_transposed<0, I, R>), fz_grid(nitems, I), blk, 0, st, A, items, nitems, (const float*)w.t, w.part, done)
        switch (fz_variant()) {
            case 0: { FZ_APPLY(4, 1); break; }
            case 1: { FZ_APPLY(8, 1); break; }
            case 2: { FZ_APPLY(4, 2); break; }
            default: { FZ_APPLY(8, 2); break; }
        }
    }
    hipLaunchKernelGGL((k_fz_gather<0>), dim3(nksr_blocks((int64_t)A.M * 32, 256)), dim3(256), 0, st, A, offsets, (const float*)w.part, x, reg, y, done);
    return NKSR_OK;
}

extern "C" int nksr_fused_apply(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, const int32_t* items,
                                int64_t nitems, float reg, void* workspace, const float* x, float* y, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    if (A.M <= 0) return NKSR_OK;
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    fz_apply(A, offsets, (const int4*)items, (int)nitems, reg, fz_carve(workspace, A, nitems), x, y, nullptr, (hipStream_t)stream);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_rhs_diag(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, const int32_t* items,
                                   int64_t nitems, float reg, void* workspace, float* b_out, float* diag_out, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    if (A.M <= 0) return NKSR_OK;
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    const FusedWork w = fz_carve(workspace, A, nitems);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = fz_grid(nitems, 4), gm(nksr_blocks((int64_t)A.M * 32, 256));
    const float* nof = nullptr;
    const int* nod = nullptr;
    if (b_out) {
        if (nitems > 0) hipLaunchKernelGGL((k_fz_transposed<1, 4, 1>), grid, dim3(FZ_BLOCK), 0, st, A, (const int4*)items, (int)nitems, nof, w.part, nod);
        hipLaunchKernelGGL((k_fz_gather<1>), gm, dim3(256), 0, st, A, offsets, (const float*)w.part, nof, reg, b_out, nod);
    }
    if (diag_out) {
        if (nitems > 0) hipLaunchKernelGGL((k_fz_transposed<2, 4, 1>), grid, dim3(FZ_BLOCK), 0, st, A, (const int4*)items, (int)nitems, nof, w.part, nod);
        hipLaunchKernelGGL((k_fz_gather<2>), gm, dim3(256), 0, st, A, offsets, (const float*)w.part, nof, reg, diag_out, nod);
    }
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

struct FusedOperator : PcgOperator {
    FusedArgs A; const int32_t* offsets; const int4* items; int nitems; float reg; FusedWork w;
    int apply(const float* p, float* y, const int* done, hipStream_t st) override { return fz_apply(A, offsets, items, nitems, reg, w, p, y, done, st); }
    void bytes(double* alg, double* phys) override {
        // SURVEY.md section 8d, matrix-free operator: G and Q once in each direction at 8 bytes per stored entry (value + index)
        // + the vectors.  The dense-slot layout stores no indices: 4 bytes per slot per direction, plus the partial t vectors,
        // the per-item stencil / block traffic and the item records
        const double slots = 27.0 * A.hier.depth * (double)A.rows_total;
        *alg = 2.0 * 8.0 * slots + 12.0 * A.M + 4.0;
        *phys = 2.0 * 4.0 * slots + (2.0 * A.hier.depth + 3.0) * 4.0 * (double)A.rows_total + (2.0 * 128.0 + 2.0 * 16.0 + 216.0) * nitems + 8.0 * A.M;
    }
};

extern "C" int nksr_pcg_solve_fused(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, const int32_t* items,
                                    int64_t nitems, float reg, const float* diag, const float* b, float* x, float tol, int max_iter,
                                    int check_every, void* workspace, void* pcg_workspace, double* info_out, void* stream) {
    FusedOperator op;
    if (int rc = fz_args(op.A, h, sets, nsets)) return rc;
    if (op.A.M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!workspace || !pcg_workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    op.offsets = offsets; op.items = (const int4*)items; op.nitems = (int)nitems; op.reg = reg;
    op.w = fz_carve(workspace, op.A, nitems);
    return nksr_pcg_run(op, diag, op.A.M, b, x, tol, max_iter, check_every, pcg_workspace, info_out, (hipStream_t)stream);
}

extern "C" size_t nksr_pcg_vector_workspace_bytes(int32_t M) { return nksr_pcg_vector_bytes(M); }
