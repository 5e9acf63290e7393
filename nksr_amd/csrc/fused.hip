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

#define FZ_RC 32
#define FZ_BLOCK 256
#define FZ_MAX_SETS 2
#define FZ_ILP 4

struct FusedArgs {
    nksr_hier_t hier;
    nksr_fused_set_t sets[FZ_MAX_SETS];
    int nsets;
    int M;
    int64_t row_off[FZ_MAX_SETS];                     // first row of the set in the concatenated t vector
    int64_t rows_total;
};

static int fz_args(FusedArgs& A, const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 1 || nsets > FZ_MAX_SETS) return nksr_set_error(NKSR_ERR_ARG, "1..%d site sets", FZ_MAX_SETS);
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    A.nsets = nsets;
    A.M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int64_t rows = 0;
    for (int s = 0; s < nsets; ++s) {
        if (sets[s].ncomp != 1 && sets[s].ncomp != 3) return nksr_set_error(NKSR_ERR_ARG, "ncomp must be 1 or 3");
        if (sets[s].n * sets[s].ncomp >= ((int64_t)1 << 31)) return nksr_set_error(NKSR_ERR_CAPACITY, "site set too large");
        A.sets[s] = sets[s];
        A.row_off[s] = rows;
        rows += sets[s].n * sets[s].ncomp;
    }
    A.rows_total = rows;
    return NKSR_OK;
}

__device__ __forceinline__ int fz_level(const nksr_hier_t& h, int j) {
    int d = 0;
    while (d + 1 < h.depth && j >= h.lv[d + 1].offset) ++d;
    return d;
}

// Work items: the rows of ALL site sets inside one cell (= one unknown's voxel), sets in order, cut into pieces of <= FZ_RC rows.
// record = { level | rows of set 0 << 4 | rows of set 1 << 12,  cell,  first row in set 0,  first row in set 1 }
__device__ __forceinline__ void fz_cell_rows(const FusedArgs& A, int d, int c, int r0[FZ_MAX_SETS], int n[FZ_MAX_SETS]) {
#pragma unroll
    for (int s = 0; s < FZ_MAX_SETS; ++s) {
        r0[s] = n[s] = 0;
        if (s < A.nsets) {
            const nksr_fused_set_t& S = A.sets[s];
            r0[s] = S.start[d][c] * S.ncomp;
            n[s] = (S.end[d][c] - S.start[d][c]) * S.ncomp;
        }
    }
}

__global__ void k_fz_item_counts(FusedArgs A, int32_t* __restrict__ counts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > A.M) return;
    if (j == A.M) { counts[j] = 0; return; }
    const int d = fz_level(A.hier, j);
    int r0[FZ_MAX_SETS], n[FZ_MAX_SETS];
    fz_cell_rows(A, d, j - A.hier.lv[d].offset, r0, n);
    counts[j] = (n[0] + n[1] + FZ_RC - 1) / FZ_RC;
}

__global__ void k_fz_item_fill(FusedArgs A, const int32_t* __restrict__ offsets, int4* __restrict__ items) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= A.M) return;
    const int d = fz_level(A.hier, j), c = j - A.hier.lv[d].offset;
    int r0[FZ_MAX_SETS], n[FZ_MAX_SETS];
    fz_cell_rows(A, d, c, r0, n);
    const int total = n[0] + n[1];
    int it = offsets[j];
    for (int q = 0; q < total; q += FZ_RC, ++it) {          // piece = rows [q, q + FZ_RC) of the sequence (set 0 rows, set 1 rows)
        const int e = q + FZ_RC < total ? q + FZ_RC : total;
        const int a0 = q < n[0] ? q : n[0], a1 = e < n[0] ? e : n[0];
        const int b0 = q > n[0] ? q - n[0] : 0, b1 = e > n[0] ? e - n[0] : 0;
        items[it] = make_int4(d | ((a1 - a0) << 4) | ((b1 - b0) << 12), c, r0[0] + a0, r0[1] + b0);
    }
}

__device__ __forceinline__ float half_sum(float p) {      // sum over the 32 lanes of this half-wave, fixed tree
    p += __shfl_xor(p, 16, 32);
    p += __shfl_xor(p, 8, 32);
    p += __shfl_xor(p, 4, 32);
    p += __shfl_xor(p, 2, 32);
    p += __shfl_xor(p, 1, 32);
    return p;
}

// one item of a half-wave's bundle, decoded
struct FzItem {
    const float* base[FZ_MAX_SETS];     // first row of the item in set s (this lane's slot)
    int64_t trow[FZ_MAX_SETS];          // index of that row in the concatenated t vector
    int n[FZ_MAX_SETS];
    int d, c;
};
__device__ __forceinline__ FzItem fz_decode(const FusedArgs& A, const int4 it, bool valid, int s) {
    FzItem I;
    I.d = it.x & 15;
    I.c = it.y;
    I.n[0] = valid ? (it.x >> 4) & 255 : 0;
    I.n[1] = valid ? (it.x >> 12) & 255 : 0;
    const int r0[FZ_MAX_SETS] = {it.z, it.w};
#pragma unroll
    for (int q = 0; q < FZ_MAX_SETS; ++q) {
        const nksr_fused_set_t& S = A.sets[q];
        I.base[q] = q < A.nsets ? S.rows + ((int64_t)I.d * (S.n * S.ncomp) + r0[q]) * 27 + (s < 27 ? s : 0) : nullptr;
        I.trow[q] = A.row_off[q] + r0[q];
    }
    return I;
}

// t_d[r] = sum_s rows[d][r][s] * x[stencil of the item's cell][s]
// Every half-wave carries FZ_ILP consecutive items at once: a fine cell holds only a handful of rows, so one item per
// half-wave is a chain of three dependent loads (item -> neighbour table -> x) in front of a few row loads, and the kernel
// ran at the latency of that chain.  Consecutive items are neighbouring cells (similar row counts): their chains and row
// loads overlap.
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_forward(FusedArgs A, const int4* __restrict__ items, int nitems,
                                                        const float* __restrict__ x, float* __restrict__ tpart,
                                                        const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * FZ_ILP;
    if (i0 >= nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[FZ_ILP];
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k) it[k] = items[i0 + k < nitems ? i0 + k : nitems - 1];
    FzItem I[FZ_ILP];
    int nb[FZ_ILP], maxrows = 0;
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k) {
        I[k] = fz_decode(A, it[k], i0 + k < nitems, s);
        nb[k] = act ? A.hier.lv[I[k].d].nbr[(int64_t)I[k].c * 27 + s] : -1;
        const int tot = I[k].n[0] + I[k].n[1];
        maxrows = tot > maxrows ? tot : maxrows;
    }
    float xs[FZ_ILP];
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k) xs[k] = nb[k] >= 0 ? x[A.hier.lv[I[k].d].offset + nb[k]] : 0.f;
    for (int j = 0; j < maxrows; ++j) {
        float v[FZ_ILP];
#pragma unroll
        for (int k = 0; k < FZ_ILP; ++k) {
            const int jb = j - I[k].n[0];
            v[k] = 0.f;
            if (act) {
                if (j < I[k].n[0]) v[k] = I[k].base[0][(int64_t)j * 27];
                else if (jb < I[k].n[1]) v[k] = I[k].base[1][(int64_t)jb * 27];
            }
        }
#pragma unroll
        for (int k = 0; k < FZ_ILP; ++k) {
            const float p = half_sum(v[k] * xs[k]);
            const int jb = j - I[k].n[0];
            if (s == 0) {
                float* tp = tpart + (int64_t)I[k].d * A.rows_total;
                if (j < I[k].n[0]) tp[I[k].trow[0] + j] = p;
                else if (jb < I[k].n[1]) tp[I[k].trow[1] + jb] = p;
            }
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
template <int MODE>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_transposed(FusedArgs A, const int4* __restrict__ items, int nitems,
                                                           const float* __restrict__ t, float* __restrict__ part,
                                                           const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * FZ_ILP;
    if (i0 >= nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[FZ_ILP];
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k) it[k] = items[i0 + k < nitems ? i0 + k : nitems - 1];
    FzItem I[FZ_ILP];
    const float* w[FZ_ILP][FZ_MAX_SETS];
    float acc[FZ_ILP];
    int maxrows = 0;
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k) {
        I[k] = fz_decode(A, it[k], i0 + k < nitems, s);
#pragma unroll
        for (int q = 0; q < FZ_MAX_SETS; ++q) {
            w[k][q] = MODE == 0 ? t + I[k].trow[q] : nullptr;
            if (MODE == 1) {      // a set without targets adds nothing to the right-hand side
                const float* tg = q < A.nsets ? A.sets[q].target : nullptr;
                w[k][q] = tg ? tg + (I[k].trow[q] - A.row_off[q]) : nullptr;
                if (!tg) I[k].n[q] = (q == 0) ? -I[k].n[q] : 0;       // set 0 rows are skipped but still counted in the row sequence
            }
        }
        const int tot = (I[k].n[0] < 0 ? -I[k].n[0] : I[k].n[0]) + I[k].n[1];
        maxrows = tot > maxrows ? tot : maxrows;
        acc[k] = 0.f;
    }
    for (int j = 0; j < maxrows; ++j) {
        float v[FZ_ILP], wk[FZ_ILP];
#pragma unroll
        for (int k = 0; k < FZ_ILP; ++k) {
            const int na = I[k].n[0] < 0 ? -I[k].n[0] : I[k].n[0];
            const int jb = j - na;
            v[k] = wk[k] = 0.f;
            if (j < na) {
                if (I[k].n[0] > 0) { v[k] = I[k].base[0][(int64_t)j * 27]; wk[k] = MODE == 2 ? v[k] : w[k][0][j]; }
            } else if (jb < I[k].n[1]) {
                v[k] = I[k].base[1][(int64_t)jb * 27];
                wk[k] = MODE == 2 ? v[k] : w[k][1][jb];
            }
        }
#pragma unroll
        for (int k = 0; k < FZ_ILP; ++k) acc[k] = fmaf(v[k], wk[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < FZ_ILP; ++k)
        if (i0 + k < nitems) part[(int64_t)(i0 + k) * 32 + s] = act ? acc[k] : 0.f;
}

// y_j = (MODE 0: reg x_j, 1: 0, 2: reg) + sum over the 27 neighbour cells c of j and the items of c:  P[item][26 - s']
// One half-wave per unknown, lane = neighbour slot: the 27 (cell -> items -> block entry) chains run side by side; fixed tree
// reduction.  The item offsets are indexed by the cell's own unknown index.
template <int MODE>
__global__ void __launch_bounds__(256) k_fz_gather(FusedArgs A, const int32_t* __restrict__ offsets, const float* __restrict__ part,
                                                  const float* __restrict__ x, float reg, float* __restrict__ y,
                                                  const int* __restrict__ done) {
    if (done && *done) return;
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 5;
    if (j >= A.M) return;
    const int sp = threadIdx.x & 31;
    const int d = fz_level(A.hier, j);
    const nksr_level_t& lv = A.hier.lv[d];
    const int i = j - lv.offset;
    float acc = 0.f;
    const int c = sp < 27 ? lv.nbr[(int64_t)i * 27 + sp] : -1;
    if (c >= 0) {
        const int i0 = offsets[lv.offset + c], i1 = offsets[lv.offset + c + 1];
        for (int itx = i0; itx < i1; ++itx) acc += part[(int64_t)itx * 32 + (26 - sp)];
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
    (void)nsets;                     // one entry per unknown (= cell): the rows of all site sets inside it share its work items
    int64_t n = 0;
    for (int d = 0; d < h->depth; ++d) n += h->lv[d].n;
    return n;
}

extern "C" int nksr_fused_item_counts(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, int32_t* counts_out, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    hipLaunchKernelGGL(k_fz_item_counts, dim3(nksr_blocks((int64_t)A.M + 1, 256)), dim3(256), 0, (hipStream_t)stream, A, counts_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_items(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, int32_t* items_out,
                                void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, h, sets, nsets)) return rc;
    if (A.M > 0) {
        hipLaunchKernelGGL(k_fz_item_fill, dim3(nksr_blocks(A.M, 256)), dim3(256), 0, (hipStream_t)stream, A, offsets, (int4*)items_out);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}

static int fz_apply(const FusedArgs& A, const int32_t* offsets, const int4* items, int nitems, float reg, const FusedWork& w,
                    const float* x, float* y, const int* done, hipStream_t st) {
    if (nitems > 0) {
        const dim3 grid(nksr_blocks(((int64_t)nitems + FZ_ILP - 1) / FZ_ILP * 32, FZ_BLOCK));
        hipLaunchKernelGGL(k_fz_forward, grid, dim3(FZ_BLOCK), 0, st, A, items, nitems, x, w.tpart, done);
        hipLaunchKernelGGL(k_fz_tsum, dim3(nksr_blocks(A.rows_total, 256)), dim3(256), 0, st, A.hier.depth, A.rows_total, (const float*)w.tpart, w.t, done);
        hipLaunchKernelGGL((k_fz_transposed<0>), grid, dim3(FZ_BLOCK), 0, st, A, items, nitems, (const float*)w.t, w.part, done);
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
    const dim3 grid(nksr_blocks((nitems + FZ_ILP - 1) / FZ_ILP * 32, FZ_BLOCK)), gm(nksr_blocks((int64_t)A.M * 32, 256));
    const float* nof = nullptr;
    const int* nod = nullptr;
    if (b_out) {
        if (nitems > 0) hipLaunchKernelGGL((k_fz_transposed<1>), grid, dim3(FZ_BLOCK), 0, st, A, (const int4*)items, (int)nitems, nof, w.part, nod);
        hipLaunchKernelGGL((k_fz_gather<1>), gm, dim3(256), 0, st, A, offsets, (const float*)w.part, nof, reg, b_out, nod);
    }
    if (diag_out) {
        if (nitems > 0) hipLaunchKernelGGL((k_fz_transposed<2>), grid, dim3(FZ_BLOCK), 0, st, A, (const int4*)items, (int)nitems, nof, w.part, nod);
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
