// Matrix-free ("fused") normal-equation operator  y = (sum_s R_s^T R_s + reg I) x  and its Jacobi-PCG solve.
// Reference: reconstruct(..., fused_mode=True) (examples/recons_waymo.py:33, recons_waymo_cpu.py:58, gis_app.py:40) -- the
// memory-lean solve that never materialises the system matrix; KernelField.solve (the assembled twin is solve_non_fused,
// models/nksr_net.py:105-112).  R_s are the dense-slot kernel rows of a site set (G: one row per input point, Q: three
// gradient rows per normal site), already multiplied by sqrt(weight), stored LEVEL-MAJOR in ONE array
// rows_all[d][r][27] (r runs over the rows of set 0, then set 1).
//
// Every site of a level-d cell c couples to the same 27 voxels (c's stencil), so both products run cell by cell:
//   forward     t_d[r]   = sum_s rows[d][r][s] * x[nbr[c][s]]          (x stencil of the cell loaded once, 27 lanes)
//   transposed  P[c][s]  = sum_{r in c} rows[d][r][s] * t[r],   y_j = reg x_j + sum_{s'} P[nbr[j][s']][26 - s']
// with t = sum_d t_d.  Work items = (set, level, cell, <= 32 consecutive rows); one item per 32-lane half of a wavefront
// (27 lanes active), so fine cells (a handful of rows) and coarse cells (thousands of rows, cut into many items) balance.
// Both passes read every row once (the rows of a cell are contiguous: sites are Morton-sorted): 2 x 4 bytes per dense slot
// per application and no column indices at all.  Fixed summation orders, no float atomics: deterministic.  No assembly: the
// solve starts right after the kernel rows.
//
// An item record holds everything a half-wave needs (row offset, global index of its cell, row count, level): the kernels
// index three flat arrays with uniform base pointers.  (The first version looked levels and sets up in the argument
// struct per item -- per-lane indexing of a kernel argument compiles to ~8 dependent global loads per item and was most of
// the runtime for cells with 3 rows.)
#include "common.h"
#include "pcg_core.h"
#include <stdlib.h>

#define FZ_RC 32
#define FZ_BLOCK 256
#define FZ_MAX_SETS 2
#define FZ_DEFAULT_VARIANT 2

// ---- work items ------------------------------------------------------------------------------------------------------------
// item = { trow: first row (index into the concatenated row list of all sets), cell: global unknown index of the cell,
//          meta: rows | level << 8, 0 };  offsets[set * M + cell] .. [+1] = the items of (set, cell)
struct ItemArgs {
    nksr_hier_t hier;
    nksr_fused_set_t sets[FZ_MAX_SETS];
    int nsets;
    int M;
    int64_t row_off[FZ_MAX_SETS];
};

static int fz_item_args(ItemArgs& A, const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 1 || nsets > FZ_MAX_SETS) return nksr_set_error(NKSR_ERR_ARG, "1..%d site sets", FZ_MAX_SETS);
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    A.nsets = nsets;
    A.M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int64_t rows = 0;
    for (int s = 0; s < nsets; ++s) {
        if (sets[s].ncomp != 1 && sets[s].ncomp != 3) return nksr_set_error(NKSR_ERR_ARG, "ncomp must be 1 or 3");
        A.sets[s] = sets[s];
        A.row_off[s] = rows;
        rows += sets[s].n * sets[s].ncomp;
    }
    if (rows >= ((int64_t)1 << 31) || (int64_t)nsets * A.M >= ((int64_t)1 << 31) - 1) return nksr_set_error(NKSR_ERR_CAPACITY, "site sets too large");
    return NKSR_OK;
}

__device__ __forceinline__ int fz_level(const nksr_hier_t& h, int j) {
    int d = 0;
    while (d + 1 < h.depth && j >= h.lv[d + 1].offset) ++d;
    return d;
}

__global__ void k_fz_item_counts(ItemArgs A, int32_t* __restrict__ counts) {
    const int lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin > A.nsets * A.M) return;
    if (lin == A.nsets * A.M) { counts[lin] = 0; return; }
    const int set = lin / A.M, j = lin - set * A.M;
    const int d = fz_level(A.hier, j), c = j - A.hier.lv[d].offset;
    const nksr_fused_set_t& S = A.sets[set];
    const int nrows = (S.end[d][c] - S.start[d][c]) * S.ncomp;
    counts[lin] = (nrows + FZ_RC - 1) / FZ_RC;
}

__global__ void k_fz_item_fill(ItemArgs A, const int32_t* __restrict__ offsets, int4* __restrict__ items) {
    const int lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= A.nsets * A.M) return;
    const int set = lin / A.M, j = lin - set * A.M;
    const int d = fz_level(A.hier, j), c = j - A.hier.lv[d].offset;
    const nksr_fused_set_t& S = A.sets[set];
    const int r0 = S.start[d][c] * S.ncomp, r1 = S.end[d][c] * S.ncomp;
    int it = offsets[lin];
    for (int r = r0; r < r1; r += FZ_RC, ++it)
        items[it] = make_int4((int)A.row_off[set] + r, j, ((r + FZ_RC < r1 ? FZ_RC : r1 - r)) | (d << 8), 0);
}

// ---- the operator ----------------------------------------------------------------------------------------------------------
struct FusedArgs {               // uniform scalars and base pointers only
    const float* rows_all;       // [depth][rows_total][27]
    const float* targets_all;    // [rows_total]
    const int32_t* nbr_all;      // [M,27] global unknown index or -1
    const int32_t* offsets;      // [nsets * M + 1]
    const int4* items;
    int nitems, nsets, M, depth;
    int64_t rows_total;
};

__device__ __forceinline__ float half_sum(float p) {      // sum over the 32 lanes of this half-wave, fixed tree
    p += __shfl_xor(p, 16, 32);
    p += __shfl_xor(p, 8, 32);
    p += __shfl_xor(p, 4, 32);
    p += __shfl_xor(p, 2, 32);
    p += __shfl_xor(p, 1, 32);
    return p;
}

// sums of FOUR rows over the 32 lanes of a half-wave in 6 lane exchanges instead of 4 x 5: a transposing butterfly -- after the
// xor-16 step a lane keeps two of the four rows (its own + its partner's share), after xor-8 one, then 3 plain steps.
// Returns, in every lane, the total of row  2 * bit4(lane) + bit3(lane).
// lane exchanges inside a 16-lane row as DPP modifiers of a VALU move (no LDS crossbar): quad_perm [1,0,3,2] (xor 1),
// [2,3,0,1] (xor 2), row_half_mirror (l -> 7 - l: the other quad of an 8-lane group once quads are uniform), row_ror:8 (xor 8)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_sum4(float p0, float p1, float p2, float p3, int lane) {
    const bool hi = (lane >> 4) & 1, b = (lane >> 3) & 1;
    const float r0 = __shfl_xor(hi ? p0 : p2, 16, 32), r1 = __shfl_xor(hi ? p1 : p3, 16, 32);       // the only two crossbar trips
    const float q0 = (hi ? p2 : p0) + r0, q1 = (hi ? p3 : p1) + r1;          // rows {2,3} in the upper 16 lanes, {0,1} in the lower
    float r = (b ? q1 : q0) + dpp_move<0x128>(b ? q0 : q1);                  // row_ror:8
    r += dpp_move<0xB1>(r);                                                  // xor 1
    r += dpp_move<0x4E>(r);                                                  // xor 2
    r += dpp_move<0x141>(r);                                                 // row_half_mirror: the other quad
    return r;
}

// t_d[r] = sum_s rows[d][r][s] * x[stencil of the item's cell][s]
// Every half-wave carries ILP consecutive items at once (their load chains item -> neighbour row -> x overlap), four rows per
// trip.  The row sums leave through 4 lanes per trip (16 contiguous bytes): one store instruction per four rows, and the
// cross-lane exchanges -- the LDS crossbar was this kernel's bottleneck at one 5-step reduction per row -- drop 3.3x.
template <int ILP>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_forward(FusedArgs A, const float* __restrict__ x, float* __restrict__ tpart,
                                                        const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * ILP;
    if (i0 >= A.nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) it[k] = A.items[i0 + k < A.nitems ? i0 + k : A.nitems - 1];
    int nb[ILP], nrows[ILP], maxrows = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        nb[k] = act ? A.nbr_all[(int64_t)it[k].y * 27 + s] : -1;
        nrows[k] = i0 + k < A.nitems ? (it[k].z & 255) : 0;
        maxrows = nrows[k] > maxrows ? nrows[k] : maxrows;
    }
    float xs[ILP];
    const float* base[ILP];
    float* tp[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        xs[k] = nb[k] >= 0 ? x[nb[k]] : 0.f;
        base[k] = A.rows_all + ((int64_t)(it[k].z >> 8) * A.rows_total + it[k].x) * 27 + (act ? s : 0);
        tp[k] = tpart + (int64_t)(it[k].z >> 8) * A.rows_total + it[k].x;
    }
    const int myrow = 2 * ((s >> 4) & 1) + ((s >> 3) & 1);         // the row of a trip whose total this lane ends up with
    for (int j = 0; j < maxrows; j += 4) {
        float v[ILP][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) v[k][u] = (act && j + u < nrows[k]) ? base[k][(int64_t)(j + u) * 27] : 0.f;
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const float r = half_sum4(v[k][0] * xs[k], v[k][1] * xs[k], v[k][2] * xs[k], v[k][3] * xs[k], s);
            if ((s & 7) == 0 && j + myrow < nrows[k]) tp[k][j + myrow] = r;
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
// The <= 32 weights of an item arrive as ONE coalesced load (lane j holds the weight of row j) and are handed out by lane
// broadcasts, instead of one same-address load per row.
template <int MODE, int ILP, int RU>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_transposed(FusedArgs A, const float* __restrict__ t, float* __restrict__ part,
                                                           const int* __restrict__ done) {
    if (done && *done) return;
    const int hw = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    const int i0 = hw * ILP;
    if (i0 >= A.nitems) return;
    const int s = threadIdx.x & 31;
    const bool act = s < 27;
    int4 it[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) it[k] = A.items[i0 + k < A.nitems ? i0 + k : A.nitems - 1];
    int nrows[ILP], maxrows = 0;
    const float* base[ILP];
    float acc[ILP], wreg[ILP];
    const float* w = MODE == 0 ? t : A.targets_all;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        nrows[k] = i0 + k < A.nitems ? (it[k].z & 255) : 0;
        maxrows = nrows[k] > maxrows ? nrows[k] : maxrows;
        base[k] = A.rows_all + ((int64_t)(it[k].z >> 8) * A.rows_total + it[k].x) * 27 + (act ? s : 0);
        wreg[k] = (MODE != 2 && s < nrows[k]) ? w[it[k].x + s] : 0.f;
        acc[k] = 0.f;
    }
    for (int j = 0; j < maxrows; j += RU) {
        float v[ILP][RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) v[k][u] = (j + u < nrows[k]) ? base[k][(int64_t)(j + u) * 27] : 0.f;
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                // row j + u is the same lane of both halves' items: two scalar lane reads + a select, no crossbar
                float wk = v[k][u];
                if (MODE != 2) {
                    const float wl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[k]), (j + u) & 31)),
                                wh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wreg[k]), 32 + ((j + u) & 31)));
                    wk = (threadIdx.x & 32) ? wh : wl;
                }
                acc[k] = fmaf(v[k][u], wk, acc[k]);
            }
    }
#pragma unroll
    for (int k = 0; k < ILP; ++k)
        if (i0 + k < A.nitems) part[(int64_t)(i0 + k) * 32 + s] = act ? acc[k] : 0.f;
}

// y_j = (MODE 0: reg x_j, 1: 0, 2: reg) + sum over the 27 neighbour cells c of j, sets, items of c:  P[item][26 - s']
// One half-wave per unknown, lane = neighbour slot: the 27 (cell -> items -> block entry) chains run side by side; fixed tree
// reduction.
template <int MODE>
__global__ void __launch_bounds__(256) k_fz_gather(FusedArgs A, const float* __restrict__ part, const float* __restrict__ x, float reg,
                                                  float* __restrict__ y, const int* __restrict__ done) {
    if (done && *done) return;
    // (giving each XCD a contiguous eighth of the unknowns cut this pass's HBM fetches from 1.51 to 0.35 GB -- the partial blocks
    // are re-read by all eight L2s -- and still ran 1.5x SLOWER, like the same mapping did for the other two passes: 2x)
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 5;
    if (j >= A.M) return;
    const int sp = threadIdx.x & 31;
    float acc = 0.f;
    const int c = sp < 27 ? A.nbr_all[(int64_t)j * 27 + sp] : -1;
    if (c >= 0) {
        int i0[FZ_MAX_SETS], i1[FZ_MAX_SETS];
#pragma unroll
        for (int set = 0; set < FZ_MAX_SETS; ++set) {
            i0[set] = i1[set] = 0;
            if (set < A.nsets) {
                i0[set] = A.offsets[(int64_t)set * A.M + c];
                i1[set] = A.offsets[(int64_t)set * A.M + c + 1];
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
    float* tpart;   // [depth][rows_total]
    float* t;       // [rows_total]
    float* part;    // [nitems][32]
};
static size_t fz_align(size_t v) { return (v + 255) / 256 * 256; }
static FusedWork fz_carve(void* ws, int depth, int64_t rows_total) {
    FusedWork w;
    char* p = (char*)ws;
    w.tpart = (float*)p; p += fz_align((size_t)depth * rows_total * sizeof(float));
    w.t = (float*)p; p += fz_align((size_t)rows_total * sizeof(float));
    w.part = (float*)p;
    return w;
}

extern "C" size_t nksr_fused_workspace_bytes(int32_t depth, int64_t rows_total, int64_t nitems) {
    return fz_align((size_t)depth * rows_total * sizeof(float)) + fz_align((size_t)rows_total * sizeof(float)) +
           fz_align((size_t)nitems * 32 * sizeof(float)) + 256;
}

extern "C" int64_t nksr_fused_cells(const nksr_hier_t* h, int nsets) {
    int64_t n = 0;
    for (int d = 0; d < h->depth; ++d) n += h->lv[d].n;
    return n * nsets;
}

extern "C" int nksr_fused_item_counts(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, int32_t* counts_out, void* stream) {
    ItemArgs A;
    if (int rc = fz_item_args(A, h, sets, nsets)) return rc;
    hipLaunchKernelGGL(k_fz_item_counts, dim3(nksr_blocks((int64_t)A.nsets * A.M + 1, 256)), dim3(256), 0, (hipStream_t)stream, A, counts_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_items(const nksr_hier_t* h, const nksr_fused_set_t* sets, int nsets, const int32_t* offsets, int32_t* items_out,
                                void* stream) {
    ItemArgs A;
    if (int rc = fz_item_args(A, h, sets, nsets)) return rc;
    if (A.M > 0) {
        hipLaunchKernelGGL(k_fz_item_fill, dim3(nksr_blocks((int64_t)A.nsets * A.M, 256)), dim3(256), 0, (hipStream_t)stream, A, offsets, (int4*)items_out);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}

static int fz_args(FusedArgs& A, const nksr_fused_op_t* op) {
    if (!op) return nksr_set_error(NKSR_ERR_ARG, "operator is NULL");
    if (op->depth < 1 || op->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", op->depth);
    if (op->nsets < 1 || op->nsets > FZ_MAX_SETS) return nksr_set_error(NKSR_ERR_ARG, "1..%d site sets", FZ_MAX_SETS);
    if (op->M > 0 && (!op->rows_all || !op->nbr_all || !op->offsets || !op->workspace || (op->nitems > 0 && !op->items)))
        return nksr_set_error(NKSR_ERR_ARG, "operator has NULL arrays");
    if (op->rows_total >= ((int64_t)1 << 31) || op->nitems >= ((int64_t)1 << 31)) return nksr_set_error(NKSR_ERR_CAPACITY, "operator too large");
    A.rows_all = op->rows_all; A.targets_all = op->targets_all; A.nbr_all = op->nbr_all; A.offsets = op->offsets;
    A.items = (const int4*)op->items; A.nitems = (int)op->nitems; A.nsets = op->nsets; A.M = op->M; A.depth = op->depth;
    A.rows_total = op->rows_total;
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

static int fz_apply(const FusedArgs& A, float reg, const FusedWork& w, const float* x, float* y, const int* done, hipStream_t st) {
    if (A.nitems > 0) {
        const dim3 blk(FZ_BLOCK), gs(nksr_blocks(A.rows_total, 256));
#define FZ_APPLY(I, R)                                                                                                      \
        hipLaunchKernelGGL((k_fz_forward<I>), fz_grid(A.nitems, I), blk, 0, st, A, x, w.tpart, done);                      \
        hipLaunchKernelGGL(k_fz_tsum, gs, dim3(256), 0, st, A.depth, A.rows_total, (const float*)w.tpart, w.t, done);      \
        hipLaunchKernelGGL((k_fz_transposed<0, I, R>), fz_grid(A.nitems, I), blk, 0, st, A, (const float*)w.t, w.part, done)
        switch (fz_variant()) {
            case 0: { FZ_APPLY(4, 1); break; }
            case 1: { FZ_APPLY(8, 1); break; }
            case 2: { FZ_APPLY(4, 2); break; }
            default: { FZ_APPLY(8, 2); break; }
        }
    }
    hipLaunchKernelGGL((k_fz_gather<0>), dim3(nksr_blocks((int64_t)A.M * 32, 256)), dim3(256), 0, st, A, (const float*)w.part, x, reg, y, done);
    return NKSR_OK;
}

extern "C" int nksr_fused_apply(const nksr_fused_op_t* op, float reg, const float* x, float* y, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, op)) return rc;
    if (A.M <= 0) return NKSR_OK;
    fz_apply(A, reg, fz_carve(op->workspace, A.depth, A.rows_total), x, y, nullptr, (hipStream_t)stream);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_rhs_diag(const nksr_fused_op_t* op, float reg, float* b_out, float* diag_out, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, op)) return rc;
    if (A.M <= 0) return NKSR_OK;
    const FusedWork w = fz_carve(op->workspace, A.depth, A.rows_total);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = fz_grid(A.nitems, 4), gm(nksr_blocks((int64_t)A.M * 32, 256));
    const float* nof = nullptr;
    const int* nod = nullptr;
    if (b_out) {
        if (!A.targets_all) return nksr_set_error(NKSR_ERR_ARG, "targets_all is NULL");
        if (A.nitems > 0) hipLaunchKernelGGL((k_fz_transposed<1, 4, 1>), grid, dim3(FZ_BLOCK), 0, st, A, nof, w.part, nod);
        hipLaunchKernelGGL((k_fz_gather<1>), gm, dim3(256), 0, st, A, (const float*)w.part, nof, reg, b_out, nod);
    }
    if (diag_out) {
        if (A.nitems > 0) hipLaunchKernelGGL((k_fz_transposed<2, 4, 1>), grid, dim3(FZ_BLOCK), 0, st, A, nof, w.part, nod);
        hipLaunchKernelGGL((k_fz_gather<2>), gm, dim3(256), 0, st, A, (const float*)w.part, nof, reg, diag_out, nod);
    }
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

struct FusedOperator : PcgOperator {
    FusedArgs A; float reg; FusedWork w;
    int apply(const float* p, float* y, const int* done, hipStream_t st) override { return fz_apply(A, reg, w, p, y, done, st); }
    void bytes(double* alg, double* phys) override {
        // SURVEY.md section 8d, matrix-free operator: G and Q once in each direction at 8 bytes per stored entry (value + index)
        // + the vectors.  The dense-slot layout stores no indices: 4 bytes per slot per direction, plus the partial t vectors,
        // the item records, the per-item stencil (neighbour row + x) and partial-block traffic
        const double slots = 27.0 * A.depth * (double)A.rows_total;
        *alg = 2.0 * 8.0 * slots + 12.0 * A.M + 4.0;
        *phys = 2.0 * 4.0 * slots + (2.0 * A.depth + 3.0) * 4.0 * (double)A.rows_total + (2.0 * 16.0 + 2.0 * 108.0 + 2.0 * 128.0) * A.nitems + 8.0 * A.M;
    }
};

extern "C" int nksr_pcg_solve_fused(const nksr_fused_op_t* opd, float reg, const float* diag, const float* b, float* x, float tol, int max_iter,
                                    int check_every, void* pcg_workspace, double* info_out, void* stream) {
    FusedOperator op;
    if (int rc = fz_args(op.A, opd)) return rc;
    if (op.A.M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!pcg_workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    op.reg = reg;
    op.w = fz_carve(opd->workspace, op.A.depth, op.A.rows_total);
    return nksr_pcg_run(op, diag, op.A.M, b, x, tol, max_iter, check_every, pcg_workspace, info_out, (hipStream_t)stream);
}

extern "C" size_t nksr_pcg_vector_workspace_bytes(int32_t M) { return nksr_pcg_vector_bytes(M); }
