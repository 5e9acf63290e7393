// Matrix-free ("fused") normal-equation operator  y = (sum_s R_s^T R_s + reg I) x  and its Jacobi-PCG solve.
// Reference: reconstruct(..., fused_mode=True) (examples/recons_waymo.py:33, recons_waymo_cpu.py:58, gis_app.py:40) -- the
// memory-lean solve that never materialises the system matrix; KernelField.solve (the assembled twin is solve_non_fused,
// models/nksr_net.py:105-112).  R_s are the dense-slot kernel rows of a site set (G: one row per input point, Q: three
// gradient rows per normal site), already multiplied by sqrt(weight), stored LEVEL-MAJOR in ONE array
// rows_all[d][r][27] (r runs over the rows of set 0, then set 1).
//
// Every site of a level-d cell c couples to the same 27 voxels (c's stencil):
//   t[r]     = sum_d sum_s rows[d][r][s] * x[nbr[c_d(r)][s]]
//   P[c][s]  = sum_{r in c} rows[d][r][s] * t[r],          y_j = reg x_j + sum_{s'} P[nbr[j][s']][26 - s']
// The sites are Morton-sorted, so the rows of a cell are contiguous AT EVERY LEVEL at once.  One pass does both products
// (k_fz_sweep): a work item = 32 consecutive rows of a set, one item per 32-lane half of a wavefront (lane = stencil slot, 27
// active).  The half-wave walks its rows four at a time with the x stencil and the running block P of the CURRENT cell of every
// level in registers; when a row enters another cell of level d the finished block is written out and the new stencil
// (neighbour row + 27 x values) is fetched.  t[r] needs only the row's own slots, so it is formed (one transposing butterfly
// per four rows) and used while the row is still in registers: every kernel row is read from HBM ONCE per application, there
// are no column indices, no t vector in memory and no second pass over the rows.  A cell whose rows span several items gets one
// partial block per item; k_fz_gather sums, per unknown, the blocks of its 27 neighbour cells in a fixed order.  No float
// atomics: deterministic.
//
// (Round-2 history: the first version ran the two products as separate passes over (level, cell, <= 32 rows) items -- rows read
// twice, t through memory: 1.05 ms per application at the bench workload.)
#include "common.h"
#include "pcg_core.h"
#include <stdlib.h>

#define FZ_RC 32
// (round 3: reading the kernel rows with the non-temporal hint made the sweep 9-13 % SLOWER -- a 108-byte row shares its cache lines
// with its neighbours in the list, which the four rows of a trip and the next trip fetch again)
#define FZ_ROW_LOAD(p) (*(p))
#define FZ_BLOCK 256
#define FZ_HW (FZ_BLOCK / 32)              // items (half-waves) of a workgroup
#define FZ_WG_ROWS (FZ_RC * FZ_HW)         // 256 consecutive rows per workgroup
#define FZ_STAGE0 96                       // level-0 / level-1 cells of a workgroup whose final blocks are staged in LDS (a workgroup holds
#define FZ_STAGE1 32                       // ~64 / ~8 of them at four rows per level-0 cell; more than fit are written word by word)

// Round 4 -- where the blocks go.  A cell's block P[c][27] is final as soon as all rows of the cell have been seen.  Rounds 2-3
// wrote one partial block per 32-row item a cell touches, summed them in a second pass (k_fz_cellsum: 15 % of the cells had two
// blocks) into a cell-major array and gathered that with one lane per neighbour slot (27 scattered lines per unknown).  Now
//   * a cell that lies inside ONE workgroup's 256 rows is finished there: blocks of cells that cross an item boundary meet in LDS
//     at the end of the workgroup and are summed in item order -- only cells whose rows span several workgroups (the coarse ones:
//     ~1 % of the cells) still go through partial blocks (one per workgroup) + k_fz_cellsum;
//   * the per-cell sums are stored SLOT-MAJOR, ct[s][c]: the second product's gather then runs with one lane per UNKNOWN,
//     y_j = sum_s' ct[26 - s'][nbrT[s'][j]] -- 27 coalesced loads of the (slot-major) neighbour table and 27 gathers whose 64
//     lanes hit a few neighbouring lines (consecutive unknowns have consecutive neighbours) instead of 54.
// Every summation order is fixed and depends on the rows' position inside their segment only (segments are padded to whole
// workgroups): deterministic, and a chunk's result does not depend on its batch mates.

// ---- tables ----------------------------------------------------------------------------------------------------------------
// All they need is row_cells[d][r] (written by nksr_kernel_rows next to the rows): the rows of a cell are one contiguous run.
// span[0][j] / span[1][j]: first / last row of cell j (-1 = the cell has no rows)
__global__ void k_fz_spans(int depth, int64_t rows_total, const int32_t* __restrict__ row_cells, int32_t* __restrict__ first,
                           int32_t* __restrict__ last) {
    const int64_t lin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= rows_total * depth) return;
    const int64_t r = lin % rows_total;
    const int c = row_cells[lin];
    if (c < 0) return;
    if (r == 0 || row_cells[lin - 1] != c) first[c] = (int32_t)r;
    if (r == rows_total - 1 || row_cells[lin + 1] != c) last[c] = (int32_t)r;
}

// A UNIT = a maximal run of rows that lie in the same cell at every level (the rows of one level-0 cell, as a rule).  Item i of the
// sweep = the units that start in the 32-row window [32 i, 32 i + 32): rows [item_begin[i], item_begin[i + 1]) -- variable length,
// aligned to unit boundaries; an item is empty when a long unit covers its window.  Eight items are a workgroup.
__global__ void k_fz_item_begin(int depth, int64_t rows_total, int nitems, int nentries, const int32_t* __restrict__ row_cells,
                                int32_t* __restrict__ item_begin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nentries) return;
    int64_t r = (int64_t)i * FZ_RC;
    if (i >= nitems || r >= rows_total) { item_begin[i] = (int32_t)rows_total; return; }
    while (r > 0 && r < rows_total) {
        bool start = false;
        for (int d = 0; d < depth; ++d) start = start || row_cells[(int64_t)d * rows_total + r] != row_cells[(int64_t)d * rows_total + r - 1];
        if (start) break;
        ++r;
    }
    item_begin[i] = (int32_t)r;
}
// workgroup of the sweep that processes row r: the last w with item_begin[8 w] <= r
__device__ __forceinline__ int fz_wg_of(const int32_t* __restrict__ item_begin, int nwg, int r) {
    int lo = 0, hi = nwg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (item_begin[mid * FZ_HW] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// partial blocks of a cell: one per workgroup its rows reach into -- if that is more than one; a cell inside one workgroup has none
// (the sweep finishes it)
__global__ void k_fz_block_counts(int M, int nwg, const int32_t* __restrict__ item_begin, const int32_t* __restrict__ first,
                                  const int32_t* __restrict__ last, int32_t* __restrict__ wgfirst, int32_t* __restrict__ counts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > M) return;
    int n = 0;
    if (j < M) {
        int w0 = 0;
        if (first[j] >= 0) {
            w0 = fz_wg_of(item_begin, nwg, first[j]);
            n = fz_wg_of(item_begin, nwg, last[j]) - w0 + 1;
            if (n == 1) n = 0;
        }
        wgfirst[j] = w0;
    }
    counts[j] = n;
}

__device__ __forceinline__ int fz_level(const nksr_hier_t& h, int j) {
    int d = 0;
    while (d + 1 < h.depth && j >= h.lv[d + 1].offset) ++d;
    return d;
}

// nbr32[j][0..26]: global unknown index of the neighbour voxels or -1;  [27]: (first block of j) - (first workgroup of j), so that
// the block of workgroup w is nbr32[j][27] + w;  [28] / [29]: first / last row of the cell (-1: none);  [30] / [31]: where the cell's
// rows lie in the COMPACT row array (see below): first word / 4, and the 27-bit mask of the neighbours that exist
// One workgroup per 64 unknowns writes BOTH tables from one read of the hierarchy's neighbour rows: nbr32 row-major as it is formed,
// nbrT slot-major out of an LDS tile (round 6: two kernels read the rows, then nbr32 again: 3.7 + 1.4 ms per scene step).
__global__ void __launch_bounds__(256) k_fz_tables(nksr_hier_t hier, int M, const int32_t* __restrict__ offsets, const int32_t* __restrict__ first,
                                                   const int32_t* __restrict__ last, const int32_t* __restrict__ wgfirst,
                                                   const int32_t* __restrict__ rowbase4, int32_t* __restrict__ nbr32, int32_t* __restrict__ nbrT) {
    __shared__ int32_t tile[64][33];
    const int j0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {         // (i >> 5 is uniform over a half-wave: the ballot below is one unknown's)
        const int jj = i >> 5, s = i & 31, j = j0 + jj;
        int v = -1;
        if (j < M) {
            v = 0;
            if (s < 27) {
                const int d = fz_level(hier, j), c = j - hier.lv[d].offset;
                const int nb = hier.lv[d].nbr[(int64_t)c * 27 + s];
                v = nb >= 0 ? nb + hier.lv[d].offset : -1;
            } else if (s == 27) {
                v = offsets[j] - wgfirst[j];
            } else if (s == 28) {
                v = first[j];
            } else if (s == 29) {
                v = last[j];
            } else if (s == 30) {
                v = rowbase4 ? rowbase4[j] : 0;
            }
        }
        // the mask of the existing neighbours: the 27 lanes of this half-wave have just decided it
        const unsigned m = (unsigned)(__ballot(s < 27 && v >= 0) >> ((threadIdx.x & 32) ? 32 : 0)) & 0x7FFFFFFu;
        if (s == 31 && j < M) v = (int)m;
        tile[jj][s] = v;
        if (j < M) nbr32[(int64_t)j * 32 + s] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * 64; i += 256) {
        const int s = i >> 6, jj = i & 63;
        if (j0 + jj < M) nbrT[(int64_t)s * M + j0 + jj] = tile[jj][s];
    }
}
// ---- COMPACT rows (round 6).  A slot of a kernel row whose neighbour voxel does not exist is a structural zero -- a quarter of the
// slots by the round-5 review's count; 4 % on the 64-chunk scene when measured: HISTORY.md).  All rows of a cell share the cell's 27-bit neighbour mask, so they are stored with the
// k = popcount(mask) existing slots only, in slot order: cell j (rows first .. last of the list, one contiguous run at every level)
// owns the words [4 b4, 4 b4 + rows k) of ONE array (levels follow each other; within a level the cells lie in row order, every
// block padded to 16 bytes with zeros); words 0 .. 3 of the array are zero: a lane whose slot does not exist reads word 0 at
// stride 0.  sizes4[j] = 16-byte units of cell j's block (the caller's exclusive scan + 1 gives b4).
__global__ void k_fz_row_sizes(nksr_hier_t hier, int M, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                               int64_t* __restrict__ sizes4) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > M) return;
    int64_t v = 0;
    if (j < M && first[j] >= 0) {
        const int d = fz_level(hier, j), c = j - hier.lv[d].offset;
        int k = 0;
        for (int q = 0; q < 27; ++q) k += hier.lv[d].nbr[(int64_t)c * 27 + q] >= 0 ? 1 : 0;
        v = ((int64_t)(last[j] - first[j] + 1) * k + 3) >> 2;
    }
    sizes4[j] = v;
}
// ---- the operator ----------------------------------------------------------------------------------------------------------
struct FusedArgs {               // uniform scalars and base pointers only
    const float* rows_all;       // [depth][rows_total][27]
    const float* targets_all;    // [rows_total]
    const int32_t* row_cells;    // [depth][rows_total]
    const int32_t* nbr32;        // [M][32]
    const int32_t* nbrT;         // [27][M]
    const int32_t* item_begin;   // [8 nwg + 1] first row of every item
    const int32_t* offsets;      // [M + 1] partial blocks of a cell
    const int32_t* multi;        // cells with partial blocks: the n_big cells with more than FZ_BIG blocks first
    int n_multi, n_big, M, depth;
    int hw_total;                // items (32-row windows of the row list)
    int64_t rows_total, nblocks;
    unsigned long long* nnz_counter;   // the set-up pass (MODE 1) adds the non-zero slots it sees (may be NULL)
    const int32_t* item_seg;     // [hw_total] segment of every 32-row item, [M] segment of every unknown (batched chunks; may be NULL)
    const int32_t* unknown_seg;
    const int* seg_done;         // device flags of the PCG: segment c finished <=> seg_done[c * seg_stride] != 0 (may be NULL)
    int seg_stride;
    const int* seg_done_count;   // device: finished segments so far (may be NULL); the per-cell sums and the gather look at their
    int seg_gate;                // unknowns' segments only once >= seg_gate have finished (the look-up costs ~20 % of those passes)
    // factor form (kernel_dim 4, see k_kernel_factors in kfield.hip): the rows are rebuilt in registers from 16-byte records
    const float4* fac_vec;       // [depth][rows_total] phi (position / header rows) or J_a (gradient rows), times sqrt(w); NULL = dense rows
    const float4* fac_pos;       // [rows_total] p = x * inv_w0 (3 floats) + the row's kind (int bits)
    const float4* psi_all;       // [M] psi of every unknown (levels concatenated)
    float inv_w0;
    int dense_from;              // set-up pass: the rebuilt rows of the levels >= dense_from are also written out dense
    float* dense_out;            // [depth - dense_from][rows_total][27] or NULL (the coarse-level block of the preconditioner reads them)
    int compact;                 // rows_all is the COMPACT array (k_fz_row_sizes): a row holds the slots of its cell's existing neighbours only
    int64_t rows_words;          // words of rows_all the sweep streams (dense: 27 depth rows_total)
};
__device__ __forceinline__ bool fz_seg_done(const FusedArgs& A, const int32_t* seg_of, int64_t i) {
    return A.seg_done && seg_of && A.seg_done[(int64_t)seg_of[i] * A.seg_stride] != 0;
}
__device__ __forceinline__ bool fz_gate_open(const FusedArgs& A) {
    return A.seg_done && A.unknown_seg && A.seg_done_count && *A.seg_done_count >= A.seg_gate;
}
// true when the FZ_GI (four) cells idx[i0 ..] of this half-wave all belong to finished segments (gated, see seg_gate).
// n: valid entries from i0 on.
#define FZ_GI 4
__device__ __forceinline__ bool fz_group_done(const FusedArgs& A, const int32_t* idx, int64_t i0, int n, int lane32, bool upper) {
    if (!fz_gate_open(A)) return false;
    bool dn = true;
    if (lane32 < FZ_GI && lane32 < n) {
        const int64_t j = idx ? idx[i0 + lane32] : i0 + lane32;
        dn = A.seg_done[(int64_t)A.unknown_seg[j] * A.seg_stride] != 0;
    }
    return (unsigned)(__ballot(dn) >> (upper ? 32 : 0)) == 0xFFFFFFFFu;
}

// lane `l` of the caller's own half-wave, l uniform: two scalar lane reads + a select (no crossbar)
__device__ __forceinline__ int half_lane_i(int v, int l, bool upper) {
    const int lo = __builtin_amdgcn_readlane(v, l), hi = __builtin_amdgcn_readlane(v, 32 + l);
    return upper ? hi : lo;
}
__device__ __forceinline__ float half_lane_f(float v, int l, bool upper) { return __int_as_float(half_lane_i(__float_as_int(v), l, upper)); }


// ---- factor form: the 27 slots of a row from its 16-byte records -----------------------------------------------------------------
// Quadratic B-spline weight of the centre at offset o - 1 for local coordinate u, written  a + b (u - c)^2  with per-lane constants
// (o = 0: 0.5 (1 - u)^2, 1: 0.75 - (u - 0.5)^2, 2: 0.5 u^2); its derivative is 2 b (u - c).  u = frac(p 2^-level) is exact: the
// level-d cell of a site is floor(p 2^-d), and p 2^-d - floor(p 2^-d) needs no rounding.
// (Measured and dropped: the same sums on v_mfma_f32_4x4x1 -- one quad lane per row, weights as a K = 2 polynomial product,
// dot products as K = 4: 65 matrix instructions per trip, 27.2 ms per application of the 64-chunk scene against 22.4 ms for
// this plain vector form at 7.3e9 VALU instructions; the dense-row sweep issues 2.3e9 and takes 11 ms.)
#define FZ_RCAP 288                        // rows of a workgroup whose records are staged in LDS (a workgroup holds ~256; a unit that runs
                                           // past this reads its records from memory)
struct FzSpline { float a[3], b[3], c[3]; };      // per axis, for the lane's slot:  w(u) = a + b (u - c)^2,  dw/du = 2 b (u - c)
__device__ __forceinline__ FzSpline fz_spline_consts(int s) {
    FzSpline q;
    const int o[3] = {s / 9, (s / 3) % 3, s % 3};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {          // o = 0: 0.5 (1 - u)^2;  1: 0.75 - (u - 0.5)^2;  2: 0.5 u^2
        q.a[ax] = o[ax] == 1 ? 0.75f : 0.f;
        q.b[ax] = o[ax] == 1 ? -1.f : 0.5f;
        q.c[ax] = o[ax] == 0 ? 1.f : (o[ax] == 1 ? 0.5f : 0.f);
    }
    return q;
}
template <bool DER>
__device__ __forceinline__ void fz_axis(float p, float scale, float a, float b, float c, float k, float& w, float& dw) {
    const float u = __builtin_amdgcn_fractf(p * scale);
    const float t = u - c, bt = b * t;
    w = fmaf(bt, t, a);
    if (DER) dw = bt * k;
}
__device__ __forceinline__ float fz_dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// MODE 0: the operator (t from x).  MODE 1: the set-up pass -- right-hand side (t = target) into ct / part, Jacobi diagonal
// (P2 += rows^2) into ct2 / part2, and the count of non-zero slots, all in one sweep over the rows.
//
// Round 4, second half -- the sweep walks UNITS, not rows.  A unit is a maximal run of rows that lie in the same cell at every level
// (in practice: the rows of one level-0 cell, ~4 of them); an item = the units that START in a 32-row window (item_begin[]: variable
// length, aligned to unit boundaries), one item per 32-lane half of a wavefront, lane = stencil slot.  Per trip a half-wave takes up to
// U rows of its current unit: the stencil of the unit's level-0 cell is fetched once per UNIT (its neighbour row one unit ahead, so
// that every load of a trip is independent), level-0 blocks are always complete when their unit ends (no exchange), and the coarser
// levels change cell only BETWEEN units -- no trip is ever cut.  The row-streaming version spent ~600 instructions per 8 rows on
// finding out which rows change cell at which level (rows x levels masks, half-lane selects, the cut loop); this one ~150.
//
// FAC (round 5): the rows are not read but REBUILT -- the workgroup stages the 16-byte factor records of its ~256 rows in LDS (one
// coalesced burst per level: the only streaming loads of the kernel, all in flight at once), every cell change also fetches the psi
// stencil of the new cell next to its x stencil, and a trip turns records + stencil into the same w[u][d] the dense form loads:
//   position rows (<= 4 per trip):   w = B_s(u) <phi, psi_s>
//   a normal site (header + 3 rows): w_a = <phi, psi_s> dB_s/dx_a + <J_a, psi_s> B_s    (B, dB shared by the three rows)
// Everything downstream of w -- both products, blocks, exchange, staging -- is the dense form's code.
template <int MODE, int D, int U, bool FAC, bool CMP = false>
__global__ void __attribute__((amdgpu_waves_per_eu(FAC ? 3 : (D <= 4 ? 6 : 5)))) __launch_bounds__(FZ_BLOCK) k_fz_cells(FusedArgs A, const float* __restrict__ x, float* __restrict__ part,
                                                      float* __restrict__ part2, float* __restrict__ ct, float* __restrict__ ct2,
                                                      const int* __restrict__ done) {
    if (done && *done) return;
    // blocks of coarse cells that cross an item boundary inside this workgroup: [item][level][0: the item's first cell, which began
    // before it / 1: its last cell, which goes on after it] -- summed after the row loop (see below)
    __shared__ float xv[FZ_HW][D][2][32];
    __shared__ float xv2[MODE == 1 ? FZ_HW : 1][MODE == 1 ? D : 1][2][32];
    __shared__ int xm[FZ_HW][D][2];                                  // their cells (-1: no entry; bit 30: the cell ends in that item)
    __shared__ int xmb[FZ_HW][D][2];                                 // their block bases
    __shared__ int xbase[FZ_HW][D];                                  // block base of the current cell of every level
    // final blocks of the workgroup's level-0 / level-1 cells, staged so that they leave slot-major in contiguous runs (lane = cell):
    // written word by word, 27 different lines per block, the per-cell sums cost the sweep 130 us of partial-line writes
    constexpr int CAP0 = FZ_STAGE0, CAP1 = D > 1 ? FZ_STAGE1 : 1;
    __shared__ float st0[CAP0 * 27], st1[CAP1 * 27];
    __shared__ float st0b[MODE == 1 ? CAP0 * 27 : 1], st1b[MODE == 1 ? CAP1 * 27 : 1];
    __shared__ int stf[CAP0 + CAP1];
    static_assert(!FAC || U == 4, "a normal site is one trip of four rows");
    // (+ four overflow slots per half-wave: the records of a trip past the staged window are copied there first)
    constexpr int RS = FZ_RCAP + 4 * FZ_HW;
    __shared__ float4 fv[FAC ? D * RS : 1], fp[FAC ? RS : 1];
    for (int i = threadIdx.x; i < CAP0 + CAP1; i += FZ_BLOCK) stf[i] = 0;
    const int W0 = __builtin_amdgcn_readfirstlane(A.item_begin[blockIdx.x * FZ_HW]);
    if (FAC) {
        // the records of the workgroup's rows [W0, W1): (D + 1) coalesced bursts, all loads before the first store.  Rows past the
        // last one (a trip reads four) are zeroed: never used (their t is 0) but they must be finite
        const int W1 = __builtin_amdgcn_readfirstlane(A.item_begin[(blockIdx.x + 1) * FZ_HW]);
        if (MODE == 0 && W1 > W0 && fz_seg_done(A, A.item_seg, W0 >> 5)) return;      // (a workgroup lies inside ONE segment)
        const int nw = W1 - W0 < FZ_RCAP ? W1 - W0 : FZ_RCAP;
        if (nw > 0) {
            const int t0 = threadIdx.x, t1 = threadIdx.x + FZ_BLOCK;
            float4 v0[D + 1], v1[D + 1];
#pragma unroll
            for (int d = 0; d <= D; ++d) {
                const float4* src = (d < D ? A.fac_vec + (int64_t)d * A.rows_total : A.fac_pos) + W0;
                v0[d] = src[t0 < nw ? t0 : nw - 1];
                v1[d] = src[t1 < nw ? t1 : nw - 1];
            }
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int d = 0; d <= D; ++d) {
                float4* dst = d < D ? fv + d * RS : fp;
                if (t0 < nw + 4 && t0 < FZ_RCAP) dst[t0] = t0 < nw ? v0[d] : zero;
                if (t1 < nw + 4 && t1 < FZ_RCAP) dst[t1] = t1 < nw ? v1[d] : zero;
            }
        }
    }
    __syncthreads();
    const int hwi = threadIdx.x >> 5;
    const int item = blockIdx.x * FZ_HW + hwi;
    const int s = threadIdx.x & 31;
    const bool act = s < 27, upper = (threadIdx.x & 32) != 0;
    const int sh = upper ? 32 : 0;
    const int sc = act ? s : 26;
    const int Rb = A.item_begin[item];                               // (row numbers fit an int: rows_total < 2^31 - 512)
    int Re = A.item_begin[item + 1];
    // (the rows of a segment whose conjugate gradients have finished are skipped: a workgroup lies inside ONE segment -- segments
    // are padded to whole workgroups)
    if (MODE == 0 && Re > Rb && fz_seg_done(A, A.item_seg, Rb >> 5)) Re = Rb;
    // first cell of the staged levels with rows in this workgroup (uniform)
    const int cw0 = __builtin_amdgcn_readfirstlane((int64_t)W0 < A.rows_total ? A.row_cells[W0] : -1);
    const int cw1 = __builtin_amdgcn_readfirstlane((D > 1 && (int64_t)W0 < A.rows_total) ? A.row_cells[A.rows_total + W0] : -1);
    if (s < 2 * D) xm[hwi][s >> 1][s & 1] = -1;
    // the cells of the item's first 32 rows, one row per lane (every unit of the item starts among them)
    const int len32 = Re - Rb < 32 ? Re - Rb : 32;
    int cells[D];
    unsigned chg[D > 1 ? D : 2];     // levels >= 1, bit l: row l lies in another cell than row l - 1
    unsigned um = 0;                 // bit l: a unit starts at row l
#pragma unroll
    for (int d = 0; d < D; ++d) {
        cells[d] = s < len32 ? A.row_cells[(int64_t)d * A.rows_total + Rb + s] : -1;
        const int before = __shfl_up(cells[d], 1, 32);
        const unsigned m = (unsigned)(__ballot(s > 0 && s < len32 && cells[d] != before) >> sh);
        if (d > 0) chg[d] = m;
        um |= m;
    }
    um |= 1u;
    // of the current cell of every coarse level two bits are kept: do its rows start / end inside this item (bits 2 d, 2 d + 1)
    unsigned inside = 0;
    float P[D], P2[MODE == 1 ? D : 1], xs[D];
    bool have[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { P[d] = 0.f; P2[MODE == 1 ? d : 0] = 0.f; xs[d] = 0.f; have[d] = false; }
    // nbv: the neighbour row of the cell that becomes current at level d (lanes 27 / 28 / 29: its block base, first / last row)
    auto enter = [&](int d, int nbv) {
        const int first = __shfl(nbv, 28, 32), last = __shfl(nbv, 29, 32);
        if (s == 27) xbase[hwi][d] = nbv;
        inside = (inside & ~(3u << (2 * d))) | ((first >= Rb ? 1u : 0u) << (2 * d)) | ((last < Re ? 2u : 0u) << (2 * d));
    };
    // the final block of a cell that lies inside this workgroup: staged (levels 0 and 1, while there is room) or written word by word
    auto finish = [&](int d, int cell, float p, float p2) {
        const int rel = d == 0 ? cell - cw0 : cell - cw1;
        if (d == 0 && cw0 >= 0 && (unsigned)rel < (unsigned)CAP0) {
            if (act) { st0[rel * 27 + s] = p; if (MODE == 1) st0b[MODE == 1 ? rel * 27 + s : 0] = p2; }
            if (s == 0) stf[rel] = 1;
        } else if (d == 1 && cw1 >= 0 && (unsigned)rel < (unsigned)CAP1) {
            if (act) { st1[rel * 27 + s] = p; if (MODE == 1) st1b[MODE == 1 ? rel * 27 + s : 0] = p2; }
            if (s == 0) stf[CAP0 + rel] = 1;
        } else if (act) {
            ct[(int64_t)s * A.M + cell] = p;
            if (MODE == 1) ct2[(int64_t)s * A.M + cell] = p2;
        }
    };
    // the running block of coarse level d leaves (`cell`: its cell): complete (all rows of the cell lie in this item) -> final;
    // otherwise -> the workgroup exchange
    auto emit = [&](int d, int cell, float p, float p2) {
        const unsigned f = (inside >> (2 * d)) & 3u;
        if (f == 3u) {
            finish(d, cell, p, p2);
        } else {
            const int k = (int)(f & 1u);
            xv[hwi][d][k][s] = p;
            if (MODE == 1) xv2[MODE == 1 ? hwi : 0][MODE == 1 ? d : 0][k][s] = p2;
            if (s == 0) { xm[hwi][d][k] = cell | ((f & 2u) ? (1 << 30) : 0); xmb[hwi][d][k] = xbase[hwi][d]; }
        }
    };
    // end row of the unit that starts at position p (of the item's first 32 rows)
    auto unit_end = [&](int p) {
        const unsigned after = um & ~((2u << p) - 1u);
        return after ? Rb + (int)__builtin_ctz(after) : Re;
    };
    // ---- the first unit: its stencils at all levels (one round trip for the neighbour rows, one for x)
    bool work = Rb < Re;
    int pos = 0, r = Rb, uend = unit_end(0);
    // DENSE rows: one pointer per level (slot s of row 0), the U rows of a trip at immediate offsets.  COMPACT rows (CMP): the lane's
    // word of row r and the lane's row stride = the k existing slots of the cell (pointer = the lane's rank among them) -- or stride
    // 0 at word 0 of the array, a zero, when the lane's neighbour does not exist (or the row has no cell at the level); re-based
    // whenever the cell of a level changes, advanced by the rows of every trip.  Rows past the unit's (or the item's) last one are
    // read too (the words that follow: other rows, the array is padded) and never used: their t is 0
    const float* wp[D];
    int kq[CMP ? D : 1];
    auto locate = [&](int d, int nbv, int cell, int rcur) {
        if (!CMP) return;
        const unsigned m = (unsigned)__shfl(nbv, 31, 32);
        const int first = __shfl(nbv, 28, 32), b4 = __shfl(nbv, 30, 32);
        const bool pres = cell >= 0 && act && ((m >> s) & 1u);
        const int k = __popc(m);
        kq[CMP ? d : 0] = pres ? k : 0;
        wp[d] = pres ? A.rows_all + ((int64_t)b4 * 4 + (int64_t)(rcur - first) * k + __popc(m & ((1u << s) - 1u))) : A.rows_all;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (CMP) kq[CMP ? d : 0] = 0;
        wp[d] = FAC ? nullptr : (CMP ? A.rows_all : A.rows_all + (int64_t)d * A.rows_total * 27 + sc);
    }
    int c0 = __shfl(cells[0], 0, 32);
    int line0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ps[FAC ? D : 1];                                          // FAC: the psi stencil of the current cell of every coarse level
    {
        int cd[D], nb0[D];
        float x0[D];
        float4 p0[FAC ? D : 1];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            cd[d] = __shfl(cells[d], 0, 32);
            nb0[d] = A.nbr32[(int64_t)(cd[d] >= 0 ? cd[d] : 0) * 32 + s];
        }
#pragma unroll
        for (int d = 1; d < D; ++d) {
            x0[d] = MODE == 0 ? x[(act && nb0[d] >= 0) ? nb0[d] : 0] : 0.f;
            if (FAC) p0[d] = A.psi_all[(act && nb0[d] >= 0) ? nb0[d] : 0];
        }
        line0 = nb0[0];
        locate(0, nb0[0], cd[0], r);
#pragma unroll
        for (int d = 1; d < D; ++d) {
            have[d] = work && cd[d] >= 0;
            enter(d, nb0[d]);
            locate(d, nb0[d], cd[d], r);
            xs[d] = (have[d] && act && nb0[d] >= 0) ? x0[d] : 0.f;
            if (FAC) ps[d] = (have[d] && act && nb0[d] >= 0) ? p0[d] : zero4;
        }
        if (FAC) ps[0] = zero4;
    }
    const FzSpline bq = fz_spline_consts(sc);
    int nnz = 0;                                                     // MODE 1: this lane's non-zero slots (stored entries of G and Q)
    while (__any(work)) {
        // all loads of the trip: U rows x D levels, the x stencil of the unit's level-0 cell, the neighbour row of the NEXT unit's
        // level-0 cell (all unconditional -- clamped addresses, results masked afterwards: a load under a branch makes the compiler
        // lose count of the outstanding loads and wait for all of them)
        float w[U][D];
        if (!FAC) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int d = 0; d < D; ++d) w[u][d] = CMP ? FZ_ROW_LOAD(wp[d] + u * kq[CMP ? d : 0]) : FZ_ROW_LOAD(wp[d] + ((int64_t)r + u) * 27);
        }
        const float xg = MODE == 0 ? x[(act && line0 >= 0) ? line0 : 0] : 0.f;
        float4 pg = zero4;
        if (FAC) pg = A.psi_all[(act && line0 >= 0) ? line0 : 0];
        float tg[U];
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) tg[u] = A.targets_all ? A.targets_all[r + u] : 0.f;       // (targets_all is padded like the rows)
        }
        const int npos = uend - Rb;                                   // the next unit starts here, if the item goes on
        const int cn = __shfl(cells[0], npos & 31, 32);
        const int line0n = A.nbr32[(int64_t)((uend < Re && cn >= 0) ? cn : 0) * 32 + s];
        int nt = work ? (uend - r < U ? uend - r : U) : 0;            // rows of this trip
        if (FAC) {
            // the records of rows r .. r + 3 from LDS (all lanes of the half-wave read the same address: a broadcast)
            int rl = r - W0;
            if (rl + U > FZ_RCAP) {
                // (rare: a unit that runs past the staged window -- a cell with dozens of points: its records are copied to the
                // half-wave's four overflow slots first)
                rl = FZ_RCAP + 4 * hwi;
                if (s < 4) {
                    fp[rl + s] = A.fac_pos[(int64_t)r + s];
#pragma unroll
                    for (int d = 0; d < D; ++d) fv[d * RS + rl + s] = A.fac_vec[(int64_t)d * A.rows_total + r + s];
                }
                __builtin_amdgcn_wave_barrier();                  // (lanes s < 4 write, every lane of the half-wave reads)
            }
            float4 pq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pq[u] = fp[rl + u];
            // a trip is EITHER a run of position rows OR one normal site (header + three gradient rows: always inside one unit)
            const bool site = __float_as_int(pq[0].w) == 1;
            const bool q1 = __float_as_int(pq[1].w) == 0, q2 = q1 && __float_as_int(pq[2].w) == 0, q3 = q2 && __float_as_int(pq[3].w) == 0;
            const int np = site ? U : 1 + (q1 ? 1 : 0) + (q2 ? 1 : 0) + (q3 ? 1 : 0);
            nt = nt < np ? nt : np;
            ps[0] = (c0 >= 0 && act && line0 >= 0) ? pg : zero4;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float scale = __int_as_float((127 - d) << 23);          // 2^-d
                const float k2 = 2.f * A.inv_w0 * scale;                       // 2 / w_d
                float4 vq[U];
#pragma unroll
                for (int u = 0; u < U; ++u) vq[u] = fv[d * RS + rl + u];
                if (site) {
                    float wx, wy, wz, dx, dy, dz;
                    fz_axis<true>(pq[0].x, scale, bq.a[0], bq.b[0], bq.c[0], k2, wx, dx);
                    fz_axis<true>(pq[0].y, scale, bq.a[1], bq.b[1], bq.c[1], k2, wy, dy);
                    fz_axis<true>(pq[0].z, scale, bq.a[2], bq.b[2], bq.c[2], k2, wz, dz);
                    const float yz = wy * wz, B = wx * yz;
                    const float g = fz_dot4(vq[0], ps[d]);
                    w[0][d] = 0.f;
                    w[1][d] = fmaf(fz_dot4(vq[1], ps[d]), B, g * (dx * yz));
                    w[2][d] = fmaf(fz_dot4(vq[2], ps[d]), B, g * ((wx * wz) * dy));
                    w[3][d] = fmaf(fz_dot4(vq[3], ps[d]), B, g * ((wx * wy) * dz));
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float wx, wy, wz, unused;
                        fz_axis<false>(pq[u].x, scale, bq.a[0], bq.b[0], bq.c[0], 0.f, wx, unused);
                        fz_axis<false>(pq[u].y, scale, bq.a[1], bq.b[1], bq.c[1], 0.f, wy, unused);
                        fz_axis<false>(pq[u].z, scale, bq.a[2], bq.b[2], bq.c[2], 0.f, wz, unused);
                        w[u][d] = fz_dot4(vq[u], ps[d]) * (wx * (wy * wz));
                    }
                }
            }
            if (MODE == 1 && A.dense_out) {
                // the coarse-level block of the preconditioner is assembled from DENSE rows: the levels it covers leave here
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (d >= A.dense_from && act)
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (u < nt) A.dense_out[((int64_t)(d - A.dense_from) * A.rows_total + r + u) * 27 + s] = w[u][d];
            }
        }
        const float x0 = (c0 >= 0 && act && line0 >= 0) ? xg : 0.f;
        float t[U];
        if (MODE == 0) {
            float prod[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                prod[u] = w[u][0] * x0;
#pragma unroll
                for (int d = 1; d < D; ++d) prod[u] = fmaf(w[u][d], xs[d], prod[u]);
            }
            const float rs = half_sum4(prod[0], prod[1], prod[2 % U], prod[3 % U], s);
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = u < nt ? half_lane_f(rs, 8 * u, upper) : 0.f;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = u < nt && act;
                t[u] = u < nt ? tg[u] : 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) { w[u][d] = ok ? w[u][d] : 0.f; nnz += w[u][d] != 0.f ? 1 : 0; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                P[d] = fmaf(w[u][d], t[u], P[d]);
                if (MODE == 1) P2[d] = fmaf(w[u][d], w[u][d], P2[d]);
            }
        r += nt;
        if (CMP) {
#pragma unroll
            for (int d = 0; d < D; ++d) wp[d] += nt * kq[CMP ? d : 0];
        }
        if (work && r >= uend) {
            // the unit is done: its level-0 block is final
            if (c0 >= 0) finish(0, c0, P[0], P2[0]);
            P[0] = 0.f;
            if (MODE == 1) P2[0] = 0.f;
            work = uend < Re;
            if (work) {
                pos = npos;
                c0 = cn;
                line0 = line0n;
                locate(0, line0, c0, r);
                uend = unit_end(pos);
                // coarse levels whose cell changes with this unit: the finished block leaves, the new stencil is fetched (rare: a
                // level-1 cell holds ~8 units)
#pragma unroll
                for (int d = 1; d < D; ++d)
                    if ((chg[d] >> pos) & 1u) {
                        const int before = __shfl(cells[d], (pos + 31) & 31, 32);
                        if (have[d]) emit(d, before, P[d], P2[MODE == 1 ? d : 0]);
                        P[d] = 0.f;
                        if (MODE == 1) P2[d] = 0.f;
                        xs[d] = 0.f;
                        if (FAC) ps[d] = zero4;
                        const int cd = __shfl(cells[d], pos, 32);
                        have[d] = cd >= 0;
                        if (!have[d]) locate(d, 0, -1, r);
                        if (have[d]) {
                            const int nbv = A.nbr32[(int64_t)cd * 32 + s];
                            enter(d, nbv);
                            locate(d, nbv, cd, r);
                            if (MODE == 0 && act && nbv >= 0) xs[d] = x[nbv];
                            if (FAC && act && nbv >= 0) ps[d] = A.psi_all[nbv];
                        }
                    }
            }
        }
    }
#pragma unroll
    for (int d = 1; d < D; ++d)
        if (have[d]) emit(d, __shfl(cells[d], pos, 32), P[d], P2[MODE == 1 ? d : 0]);
    if (MODE == 1 && A.nnz_counter) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o);
        if ((threadIdx.x & 63) == 0 && nnz) atomicAdd(A.nnz_counter, (unsigned long long)nnz);           // integer: order-free
    }
    // ---- coarse cells that cross item boundaries inside this workgroup: the LAST item of the workgroup that holds a piece of the cell
    // adds the pieces in item order.  A cell that lies inside the workgroup is final; one that reaches into other workgroups leaves
    // one partial block per workgroup (k_fz_cellsum adds those).  Everything is decided from the exchange in LDS: an item holds a
    // piece of cell c <=> one of its two entries names c; a cell began in the item that holds it as its LAST cell (entry 1); items
    // may be empty (a long unit covers their window).
    __syncthreads();
#pragma unroll
    for (int d = 1; d < D; ++d)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = xm[hwi][d][k];
            if (e < 0) continue;
            const int c = e & ~(1 << 30);
            const bool ends = (e >> 30) & 1;
            bool later = false;
            if (!ends)
                for (int h = hwi + 1; h < FZ_HW; ++h) later = later || (xm[h][d][0] >= 0 && (xm[h][d][0] & ~(1 << 30)) == c);
            if (later) continue;
            int h0 = hwi;
            bool started = k == 1;
            while (!started && h0 > 0) {
                --h0;
                started = xm[h0][d][1] >= 0 && (xm[h0][d][1] & ~(1 << 30)) == c;
            }
            float acc = 0.f, acc2 = 0.f;
            for (int h = h0; h <= hwi; ++h) {
                const int kk = (h == h0 && started) ? 1 : 0;
                const int eh = h == hwi ? e : xm[h][d][kk];
                if (eh < 0 || (eh & ~(1 << 30)) != c) continue;       // (an empty item in between)
                acc += xv[h][d][kk][s];
                if (MODE == 1) acc2 += xv2[MODE == 1 ? h : 0][MODE == 1 ? d : 0][kk][s];
            }
            if (started && ends) {
                finish(d, c, acc, acc2);
            } else {
                const int base = xmb[hwi][d][k];
                part[((int64_t)base + blockIdx.x) * 32 + s] = acc;
                if (MODE == 1) part2[((int64_t)base + blockIdx.x) * 32 + s] = acc2;
            }
        }
    // ---- the staged blocks leave slot-major: lane = cell, 27 stores of contiguous runs
    __syncthreads();
    {
        const int t = threadIdx.x;
        const bool l0 = t < CAP0, l1 = !l0 && t < CAP0 + CAP1;
        if ((l0 || l1) && stf[t]) {
            const float* src = l0 ? st0 + t * 27 : st1 + (t - CAP0) * 27;
            const float* srcb = MODE == 1 ? (l0 ? st0b + t * 27 : st1b + (t - CAP0) * 27) : nullptr;
            float* dst = ct + (l0 ? cw0 + t : cw1 + (t - CAP0));
            float* dstb = MODE == 1 ? ct2 + (l0 ? cw0 + t : cw1 + (t - CAP0)) : nullptr;
#pragma unroll
            for (int q = 0; q < 27; ++q) {
                dst[(int64_t)q * A.M] = src[q];
                if (MODE == 1) dstb[(int64_t)q * A.M] = srcb[q];
            }
        }
    }
}

// ct[s][c] = sum of the partial blocks of cell c (one per workgroup of the sweep its rows reach into: the cells of A.multi --
// coarse cells, hundreds of blocks each; cells inside one workgroup never get here).  A half-wave takes FOUR cells at once
// (lane = slot): the pass is latency-bound, the first two blocks of the four cells are requested together.  Fixed order.
__global__ void __launch_bounds__(256) k_fz_cellsum(FusedArgs A, const float* __restrict__ part, float* __restrict__ ct,
                                                   const int* __restrict__ done) {
    if (done && *done) return;
    const int s = threadIdx.x & 31;
    if ((int)blockIdx.x < A.n_big) {
        // a coarse cell with many blocks (A.multi[0 .. n_big)): the whole workgroup sums it -- half-wave h takes blocks h, h + 8, ...
        // (one serial chain over hundreds of blocks was the critical path of the pass), then the eight partial sums in order
        __shared__ float partial[8][32];
        const int cell = A.multi[blockIdx.x], h = threadIdx.x >> 5;
        if (fz_gate_open(A) && A.seg_done[(int64_t)A.unknown_seg[cell] * A.seg_stride] != 0) return;          // uniform over the workgroup
        const int b0 = A.offsets[cell], n = A.offsets[cell + 1] - b0;
        const float* p = part + (int64_t)(b0 + h) * 32 + s;
        float acc = 0.f;
        int b = h;
        // (sixteen blocks requested per trip, added in the order of the four-block trips they replace: a cell of hundreds of blocks was a
        // chain of as many dependent round trips / 4)
        for (; b + 120 < n; b += 128, p += 4096) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p[256 * q];
#pragma unroll
            for (int q = 0; q < 16; q += 4) acc += (v[q] + v[q + 1]) + (v[q + 2] + v[q + 3]);
        }
        for (; b + 24 < n; b += 32, p += 1024) acc += (p[0] + p[256]) + (p[512] + p[768]);
        for (; b < n; b += 8, p += 256) acc += p[0];
        partial[h][s] = acc;
        __syncthreads();
        if (h == 0 && s < 27) {
            float t = partial[0][s];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += partial[k][s];
            ct[(int64_t)s * A.M + cell] = t;
        }
        return;
    }
    const int first = A.n_big, ncell = A.n_multi - first;
    const int i0 = ((((int)blockIdx.x - A.n_big) * 256 + (int)threadIdx.x) >> 5) * FZ_GI;
    if (i0 >= ncell) return;
    if (fz_group_done(A, A.multi + first, i0, ncell - i0, s, (threadIdx.x & 32) != 0)) return;
    int cell[FZ_GI], b0[FZ_GI], n[FZ_GI];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) cell[k] = A.multi[first + (i0 + k < ncell ? i0 + k : ncell - 1)];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) {
        b0[k] = A.offsets[cell[k]];
        n[k] = i0 + k < ncell ? A.offsets[cell[k] + 1] - b0[k] : 0;
    }
    float acc[FZ_GI], a1[FZ_GI];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) {
        acc[k] = n[k] > 0 ? part[(int64_t)b0[k] * 32 + s] : 0.f;
        a1[k] = n[k] > 1 ? part[(int64_t)(b0[k] + 1) * 32 + s] : 0.f;
    }
    // the four cells' chains side by side (each cell's own order of additions as before: four blocks per trip, then one by one)
    int nmax = 0;
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) { acc[k] += a1[k]; nmax = n[k] > nmax ? n[k] : nmax; }
    for (int b = 2; b < nmax; b += 4) {
        float v[FZ_GI][4];
#pragma unroll
        for (int k = 0; k < FZ_GI; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k][q] = b + q < n[k] ? part[(int64_t)(b0[k] + b + q) * 32 + s] : 0.f;
#pragma unroll
        for (int k = 0; k < FZ_GI; ++k) {
            if (b + 4 <= n[k]) acc[k] += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (b + q < n[k]) acc[k] += v[k][q];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k)
        if (i0 + k < ncell && s < 27) ct[(int64_t)s * A.M + cell[k]] = acc[k];
}

// y_j = (MODE 0: reg x_j, 1: 0, 2: reg) + sum over the 27 neighbour cells c = nbrT[s'][j] of j:  ct[26 - s'][c]
// One lane per unknown: the 27 table loads are coalesced, the 27 gathers of 64 consecutive unknowns hit a few neighbouring lines
// each (consecutive unknowns have consecutive neighbours).  Fixed summation tree.  Workgroups are dealt to the XCDs round-robin by
// the hardware: workgroup 8 i + k takes the i-th group of the k-th EIGHTH of the unknowns, so that an XCD's L2 holds one contiguous
// (Morton-ordered) part of ct instead of every XCD fetching all of it.
template <int MODE>
__global__ void __launch_bounds__(256) k_fz_gather(FusedArgs A, int per_xcd, const float* __restrict__ ct, const float* __restrict__ x,
                                                  float reg, float* __restrict__ y, const int* __restrict__ done) {
    if (done && *done) return;
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
    const int local = grp * 256 + threadIdx.x;
    if (local >= per_xcd) return;
    const int j = xcd * per_xcd + local;
    if (j >= A.M) return;
    if (fz_gate_open(A) && A.seg_done[(int64_t)A.unknown_seg[j] * A.seg_stride] != 0) return;
    int c[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) c[k] = A.nbrT[(int64_t)k * A.M + j];
    float v[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = ct[(int64_t)(26 - k) * A.M + (c[k] >= 0 ? c[k] : 0)];
#pragma unroll
    for (int k = 0; k < 27; ++k) v[k] = c[k] >= 0 ? v[k] : 0.f;
    // fixed tree: 27 -> 9 -> 3 -> 1
    float r9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r9[k] = (v[3 * k] + v[3 * k + 1]) + v[3 * k + 2];
    const float r = ((r9[0] + r9[1]) + r9[2]) + ((r9[3] + r9[4]) + r9[5]) + ((r9[6] + r9[7]) + r9[8]);
    y[j] = r + (MODE == 0 ? reg * x[j] : (MODE == 2 ? reg : 0.f));
}
static void fz_gather_dims(int M, dim3& grid, int& per_xcd) {
    per_xcd = ((M + 7) / 8 + 255) / 256 * 256;                       // whole workgroups
    grid = dim3((unsigned)(per_xcd / 256 * 8));
}

static size_t fz_align(size_t v) { return (v + 255) / 256 * 256; }

// workspace: [nblocks][32] partial blocks (x 2: the set-up pass makes two) + the set-up pass's second slot-major array [27][M]
struct FusedWork { float* part; float* part2; float* ct; float* ct2; };
static size_t fz_blocks_bytes(int64_t nblocks) { return fz_align((size_t)(nblocks > 0 ? nblocks : 1) * 32 * sizeof(float)); }
extern "C" size_t nksr_fused_workspace_bytes(int64_t nblocks, int32_t M) {
    return 2 * fz_blocks_bytes(nblocks) + fz_align((size_t)(M > 0 ? M : 1) * 27 * sizeof(float));
}
static FusedWork fz_carve(const nksr_fused_op_t* op) {
    FusedWork w;
    w.part = (float*)op->workspace;
    w.part2 = (float*)((char*)op->workspace + fz_blocks_bytes(op->nblocks));
    w.ct2 = (float*)((char*)op->workspace + 2 * fz_blocks_bytes(op->nblocks));
    w.ct = op->cell_sums;
    return w;
}

static int fz_items(int64_t rows_total) { return (int)((rows_total + FZ_RC - 1) / FZ_RC); }
static int fz_nwg(int64_t rows_total) { return (fz_items(rows_total) + FZ_HW - 1) / FZ_HW; }
extern "C" int64_t nksr_fused_item_entries(int64_t rows_total) { return (int64_t)fz_nwg(rows_total) * FZ_HW + 1; }

extern "C" int nksr_fused_block_counts(int32_t depth, int32_t M, int64_t rows_total, const int32_t* row_cells, int32_t* span_out,
                                       int32_t* item_begin_out, int32_t* counts_out, void* stream) {
    if (depth < 1 || depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", depth);
    if (M <= 0) return NKSR_OK;
    if (rows_total < 0 || rows_total >= ((int64_t)1 << 31) - 2 * FZ_WG_ROWS) return nksr_set_error(NKSR_ERR_CAPACITY, "too many kernel rows");
    if (!span_out || !counts_out || !item_begin_out || (rows_total > 0 && !row_cells)) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipStream_t st = (hipStream_t)stream;
    NKSR_CHECK_HIP(hipMemsetAsync(span_out, 0xFF, (size_t)2 * M * sizeof(int32_t), st));
    const int nent = (int)nksr_fused_item_entries(rows_total);
    hipLaunchKernelGGL(k_fz_item_begin, dim3(nksr_blocks(nent, 256)), dim3(256), 0, st, depth, rows_total, fz_items(rows_total), nent, row_cells, item_begin_out);
    if (rows_total > 0)
        hipLaunchKernelGGL(k_fz_spans, dim3(nksr_blocks(rows_total * depth, 256)), dim3(256), 0, st, depth, rows_total, row_cells, span_out, span_out + M);
    hipLaunchKernelGGL(k_fz_block_counts, dim3(nksr_blocks((int64_t)M + 1, 256)), dim3(256), 0, st, M, fz_nwg(rows_total) > 0 ? fz_nwg(rows_total) : 1,
                       (const int32_t*)item_begin_out, (const int32_t*)span_out, (const int32_t*)(span_out + M), span_out + 2 * (int64_t)M, counts_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_row_sizes(const nksr_hier_t* h, const int32_t* span, int64_t* sizes4_out, void* stream) {
    if (!h || h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad hierarchy");
    const int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    if (M <= 0) return NKSR_OK;
    if (!span || !sizes4_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipLaunchKernelGGL(k_fz_row_sizes, dim3(nksr_blocks((int64_t)M + 1, 256)), dim3(256), 0, (hipStream_t)stream, *h, M, span, span + M, sizes4_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_tables(const nksr_hier_t* h, int64_t rows_total, const int32_t* item_begin, const int32_t* offsets, const int32_t* span,
                                 const int32_t* rowbase4, int32_t* nbr32_out, int32_t* nbrT_out, void* stream) {
    if (!h || h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad hierarchy");
    const int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    if (M <= 0) return NKSR_OK;
    if (!offsets || !span || !nbr32_out || !nbrT_out || !item_begin) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipLaunchKernelGGL(k_fz_tables, dim3(nksr_blocks(M, 64)), dim3(256), 0, (hipStream_t)stream, *h, M, offsets, span, span + M,
                       span + 2 * (int64_t)M, rowbase4, nbr32_out, nbrT_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

static int fz_args(FusedArgs& A, const nksr_fused_op_t* op) {
    if (!op) return nksr_set_error(NKSR_ERR_ARG, "operator is NULL");
    if (op->depth < 1 || op->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", op->depth);
    const bool fac = op->fac_vec != nullptr;
    if (fac && (!op->fac_pos || !op->psi_all || !(op->inv_w0 > 0.f) || (((uintptr_t)op->fac_vec | (uintptr_t)op->fac_pos | (uintptr_t)op->psi_all) & 15)))
        return nksr_set_error(NKSR_ERR_ARG, "factor form: fac_pos / psi_all / inv_w0 missing or arrays not 16-byte aligned");
    if (op->dense_out && (!fac || op->dense_from < 0 || op->dense_from >= op->depth))
        return nksr_set_error(NKSR_ERR_ARG, "dense_out needs the factor form and 0 <= dense_from < depth");
    if (op->M > 0 && ((!op->rows_all && !fac) || !op->row_cells || !op->nbr32 || !op->nbrT || !op->item_begin || !op->offsets || !op->workspace || !op->cell_sums ||
                      (op->n_multi > 0 && !op->multi) || op->n_big < 0 || op->n_big > op->n_multi))
        return nksr_set_error(NKSR_ERR_ARG, "operator has NULL arrays");
    if (op->rows_total < 0 || op->rows_total >= ((int64_t)1 << 31) - 2 * FZ_WG_ROWS || op->nblocks >= ((int64_t)1 << 31) - ((int64_t)1 << 26))
        return nksr_set_error(NKSR_ERR_CAPACITY, "operator too large");
    memset(&A, 0, sizeof(A));
    A.rows_all = op->rows_all; A.targets_all = op->targets_all; A.row_cells = op->row_cells; A.nbr32 = op->nbr32; A.nbrT = op->nbrT; A.item_begin = op->item_begin; A.offsets = op->offsets;
    A.multi = op->multi; A.n_multi = op->n_multi; A.n_big = op->n_big;
    A.M = op->M; A.depth = op->depth; A.rows_total = op->rows_total; A.nblocks = op->nblocks;
    A.nnz_counter = (unsigned long long*)op->nnz_counter;
    A.item_seg = op->item_seg; A.unknown_seg = op->unknown_seg;
    A.hw_total = fz_items(op->rows_total);
    A.fac_vec = (const float4*)op->fac_vec; A.fac_pos = (const float4*)op->fac_pos; A.psi_all = (const float4*)op->psi_all;
    A.inv_w0 = op->inv_w0; A.dense_from = op->dense_from; A.dense_out = op->dense_out;
    A.compact = op->compact;
    A.rows_words = op->compact ? op->rows_words : (int64_t)27 * op->depth * op->rows_total;
    if (op->compact && (fac || op->rows_words < 4)) return nksr_set_error(NKSR_ERR_ARG, "compact rows: not with the factor form; rows_words counts the leading zero block");
    return NKSR_OK;
}

// rows per trip
#define FZ_ROWS_PER_TRIP 4

template <int MODE, bool FAC, bool CMP>
static void fz_sweep_launch(const FusedArgs& A, const float* x, const FusedWork& w, const int* done, hipStream_t st) {
    constexpr int U = FZ_ROWS_PER_TRIP;
    const dim3 grid(nksr_blocks((int64_t)A.hw_total, FZ_HW)), blk(FZ_BLOCK);
    switch (A.depth) {
        case 1: hipLaunchKernelGGL((k_fz_cells<MODE, 1, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
        case 2: hipLaunchKernelGGL((k_fz_cells<MODE, 2, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
        case 3: hipLaunchKernelGGL((k_fz_cells<MODE, 3, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
        case 4: hipLaunchKernelGGL((k_fz_cells<MODE, 4, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
        case 5: hipLaunchKernelGGL((k_fz_cells<MODE, 5, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
        default: hipLaunchKernelGGL((k_fz_cells<MODE, 6, U, FAC, CMP>), grid, blk, 0, st, A, x, w.part, w.part2, w.ct, w.ct2, done); break;
    }
}
template <int MODE>
static void fz_sweep(const FusedArgs& A, const float* x, const FusedWork& w, const int* done, hipStream_t st) {
    if (A.hw_total <= 0) return;
    if (A.fac_vec) fz_sweep_launch<MODE, true, false>(A, x, w, done, st);
    else if (A.compact) fz_sweep_launch<MODE, false, true>(A, x, w, done, st);
    else fz_sweep_launch<MODE, false, false>(A, x, w, done, st);
}

static void fz_cellsum(const FusedArgs& A, const float* part, float* ct, const int* done, hipStream_t st) {
    if (A.n_multi > 0)
        hipLaunchKernelGGL(k_fz_cellsum, dim3(A.n_big + nksr_blocks(((int64_t)(A.n_multi - A.n_big) + FZ_GI - 1) / FZ_GI * 32, 256)), dim3(256), 0, st, A,
                           part, ct, done);
}

static int fz_apply(const FusedArgs& A, float reg, const FusedWork& w, const float* x, float* y, const int* done, hipStream_t st) {
    dim3 gg;
    int per_xcd;
    fz_gather_dims(A.M, gg, per_xcd);
    fz_sweep<0>(A, x, w, done, st);
    fz_cellsum(A, w.part, w.ct, done, st);
    hipLaunchKernelGGL((k_fz_gather<0>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.ct, x, reg, y, done);
    return NKSR_OK;
}

extern "C" int nksr_fused_apply(const nksr_fused_op_t* op, float reg, const float* x, float* y, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, op)) return rc;
    if (A.M <= 0) return NKSR_OK;
    fz_apply(A, reg, fz_carve(op), x, y, nullptr, (hipStream_t)stream);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_rhs_diag(const nksr_fused_op_t* op, float reg, float* b_out, float* diag_out, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, op)) return rc;
    if (A.M <= 0) return NKSR_OK;
    const FusedWork w = fz_carve(op);
    hipStream_t st = (hipStream_t)stream;
    dim3 gg;
    int per_xcd;
    fz_gather_dims(A.M, gg, per_xcd);
    const float* nof = nullptr;
    const int* nod = nullptr;
    if (b_out && !A.targets_all) return nksr_set_error(NKSR_ERR_ARG, "targets_all is NULL");
    if (!b_out && !diag_out) return NKSR_OK;
    // one sweep over the rows makes the per-cell sums of both (and counts the stored entries); cells without rows keep their zeros
    if (A.nnz_counter) (void)hipMemsetAsync(A.nnz_counter, 0, sizeof(unsigned long long), st);
    NKSR_CHECK_HIP(hipMemsetAsync(w.ct2, 0, (size_t)A.M * 27 * sizeof(float), st));
    fz_sweep<1>(A, nof, w, nod, st);
    if (b_out) {
        fz_cellsum(A, w.part, w.ct, nod, st);
        hipLaunchKernelGGL((k_fz_gather<1>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.ct, nof, reg, b_out, nod);
    }
    if (diag_out) {
        fz_cellsum(A, w.part2, w.ct2, nod, st);
        hipLaunchKernelGGL((k_fz_gather<2>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.ct2, nof, reg, diag_out, nod);
    }
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_expand_rows(const nksr_fused_op_t* op, void* stream) {
    FusedArgs A;
    if (int rc = fz_args(A, op)) return rc;
    if (A.M <= 0) return NKSR_OK;
    if (!A.fac_vec || !A.dense_out) return nksr_set_error(NKSR_ERR_ARG, "expand_rows needs the factor form and dense_out");
    A.nnz_counter = nullptr;
    const FusedWork w = fz_carve(op);
    const float* nof = nullptr;
    const int* nod = nullptr;
    fz_sweep<1>(A, nof, w, nod, (hipStream_t)stream);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

struct FusedOperator : PcgOperator {
    FusedArgs A; float reg; FusedWork w;
    int apply(const float* p, float* y, const int* done, const int* seg_done, int seg_stride, const int* seg_done_count, int seg_count,
              hipStream_t st) override {
        FusedArgs B = A;
        B.seg_done = (A.item_seg && A.unknown_seg) ? seg_done : nullptr;
        B.seg_stride = seg_stride;
        B.seg_done_count = seg_done_count;
        B.seg_gate = seg_count / 4 > 1 ? seg_count / 4 : 1;
        return fz_apply(B, reg, w, p, y, done, st);
    }
    void bytes(double* alg, double* phys, double* survey) override {
        // Algorithmic minimum of the matrix-free operator: every STORED entry of G and Q once (4 bytes:
        // the value; stored = the non-zero slots, counted by the set-up pass), the row -> cell map (4 bytes per row and level: the
        // only per-row index), one 27-entry stencil per cell (the column information, 108 bytes) and x, y once.
        // Physical: every dense slot once (zeros included), the row -> cell map, one 128-byte neighbour row per cell (sweep), the
        // slot-major per-cell sums written + read and the slot-major neighbour table (gather: 3 x 108 bytes per unknown), the
        // partial blocks of the cells that span workgroups written + read, x and y (+ x again for reg x).
        // SURVEY.md section 8d's formula prices an index per entry and both products: 2 x 8 bytes per stored entry + 12 M + 4.
        const double slots = 27.0 * A.depth * (double)A.rows_total;
        unsigned long long nnz = 0;                                  // (only reached with nksr_pcg_profile on, after a stream sync)
        if (A.nnz_counter) (void)hipMemcpy(&nnz, A.nnz_counter, sizeof(nnz), hipMemcpyDeviceToHost);
        const double stored = nnz > 0 ? (double)nnz : slots;
        *alg = 4.0 * stored + 4.0 * A.depth * (double)A.rows_total + (108.0 + 8.0) * A.M + 4.0;
        // (compact rows: the words the array holds -- the slots of existing neighbours + 16-byte padding per cell -- instead of every dense slot)
        *phys = 4.0 * (A.compact ? (double)A.rows_words : slots) + 4.0 * A.depth * (double)A.rows_total + 2.0 * 128.0 * (double)A.nblocks + (128.0 + 3.0 * 108.0 + 12.0) * A.M;
        *survey = 2.0 * 8.0 * stored + 12.0 * A.M + 4.0;
        if (A.fac_vec) {
            // Factor form (round 5): the operator no longer holds the entries of G and Q -- it rebuilds them.  Its minimum is what
            // defines it: one 16-byte record per row and level + one position record per row, the row -> cell map, and per cell the
            // 27-entry stencil (108 bytes), psi (16 bytes, every unknown's once), x and y.  Physical adds the 128-byte neighbour row
            // per cell, the slot-major per-cell sums written + read, the slot-major neighbour table and the partial blocks.
            const double rec = 16.0 * (A.depth + 1) * (double)A.rows_total + 4.0 * A.depth * (double)A.rows_total;
            *alg = rec + (108.0 + 16.0 + 8.0) * A.M + 4.0;
            *phys = rec + 2.0 * 128.0 * (double)A.nblocks + (128.0 + 16.0 + 3.0 * 108.0 + 12.0) * A.M;
        }
    }
};

extern "C" int nksr_pcg_solve_fused(const nksr_fused_op_t* opd, float reg, const float* diag, const float* b, float* x, float tol, int max_iter,
                                    int check_every, void* pcg_workspace, const nksr_coarse_precond_t* pc, const nksr_segments_t* seg,
                                    double* info_out, void* stream) {
    FusedOperator op;
    if (int rc = fz_args(op.A, opd)) return rc;
    if (op.A.M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!pcg_workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    op.reg = reg;
    op.w = fz_carve(opd);
    return nksr_pcg_run(op, diag, op.A.M, b, x, tol, max_iter, check_every, pcg_workspace, info_out, (hipStream_t)stream, pc, seg);
}

extern "C" size_t nksr_pcg_vector_workspace_bytes(int32_t M) { return nksr_pcg_vector_bytes(M); }
