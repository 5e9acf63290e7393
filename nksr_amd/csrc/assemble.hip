// Normal-equation assembly  A = sum_s w_s R_s^T R_s + reg I,  b = sum_s w_s R_s^T t_s
// (KernelField.solve_non_fused, reference call site models/nksr_net.py:105-112).
//
// Design -- two phases, no float atomics, fixed summation order:
//  1. cell blocks.  All sites (input points / normal samples) inside one level-d cell c share
//     their 27-voxel stencil at every level >= d, so their joint contribution is one dense
//     block  B[d][c] = sum_k w r_k[d][0:27]^T r_k[d:L][0:27]   of shape 27 x T_d, T_d = 27 (L-d).
//     Sites are Morton-sorted => contiguous per cell.  One wavefront per cell: lane l owns
//     columns l and l+64, the 27 row factors are broadcast with readlane, the site row is one
//     coalesced read.  Every site row is read once per level (instead of once per touching
//     matrix row: 27x less traffic than a direct gather).
//  2. structure.  k_row_count maps every structural upper slot of a row (column voxel exists and
//     the two B-spline supports overlap: integer test) to its column (colmap), counts them and
//     accumulates the in-degree of every destination row with INTEGER atomics (order-independent).
//     Exclusive scans then give the final row pointers: row = [mirrors][own upper][diagonal].
//  3. row gather.  Row i (voxel at level d) adds, for each of its 27 neighbour cells c, the block
//     row B[d][c][slot of i in c's stencil] into a structured (L-d) x 5^3 slot frame in LDS and
//     writes its own upper entries + diagonal straight into their final CSR slots; the mirrored
//     copies go to a list keyed by destination row.
//  4. a STABLE radix sort of that list on the destination-row bits (3 passes over HALF of the
//     entries) orders every row's mirrors by ascending source row; k_place_mirrors drops them into
//     the CSR.  Deterministic and exactly symmetric, no float atomics, no full COO.
#include "common.h"
#include <stdlib.h>

#define ASM_WAVES 1

struct AsmArgs {
    nksr_hier_t hier;
    nksr_siteset_t sets[3];
    int nsets;
    int M;
    int col_bits;
    float reg;
    float* blocks[NKSR_MAX_DEPTH];   // [n_d, 27, T_d]
    float* bvec[NKSR_MAX_DEPTH];     // [n_d, 27]
    int32_t* nsites[NKSR_MAX_DEPTH]; // [n_d]
    int32_t* colmap[NKSR_MAX_DEPTH]; // [n_d, (L-d) 125] column of every structural upper slot or -1
    int col_format;                  // physical layout of cols_out / vals_out (csr_phys)
};

__device__ __forceinline__ int row_level(const nksr_hier_t& h, int row) {
    int d = 0;
    while (d + 1 < h.depth && row >= h.lv[d + 1].offset) ++d;
    return d;
}

// physical CSR layout (see csrc/pcg.hip): tiles of 64 EPL entries, logical entry m of a tile at
// EPL (m % 64) + m / 64;  EPL = 4 (col_format 0, 256-entry tiles) or 3 (col_format 1, 192-entry tiles); col_format 2 = plain CSR
// order (the small coarse-level block of the PCG's preconditioner, csrc/pcg.hip)
__device__ __forceinline__ int64_t csr_phys(int64_t k, int fmt) {
    if (fmt == 2) return k;
    if (fmt == 0) {
        const int64_t m = k & 255;
        return (k & ~(int64_t)255) + 4 * (m & 63) + (m >> 6);
    }
    const uint32_t kk = (uint32_t)k;               // nnz < 2^31: 32-bit magic-number division, not the 64-bit expansion
    const uint32_t t = kk / 192u, m = kk - t * 192u;
    return (int64_t)(t * 192u + 3u * (m & 63u) + (m >> 6));
}

// ---- phase 1: one wavefront per (level, cell) on the fp32 matrix cores ------------------------------
// B[d][c] = sum_sets w (R_d)^T [R_d .. R_{L-1} | t]  is a (27 x K)(K x (T+1)) product, K = site rows of
// the cell: v_mfma_f32_32x32x2_f32 with M = 27 (of 32) block rows, NT = ceil((T+1)/32) column tiles and
// two site rows per instruction (lanes 0-31 feed row 2m, lanes 32-63 row 2m+1; lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]).  Column T carries the right-hand side
// (27 m is never a multiple of 32, so a spare column always exists).  The set weight scales the A
// operand (the host passes rows pre-multiplied by sqrt(w) and weight 1, see below).  Accumulator register r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
typedef float asm_f32x16 __attribute__((ext_vector_type(16)));

// entry `col` (slot col % 27 of level d + col / 27) of site row q: site-major rows [n ncomp, L, 27], or -- level_stride > 0 -- the
// level-major rows of the matrix-free operator ([L, level_stride, 27], site i at row row_index[i])
__device__ __forceinline__ float asm_row_value(const nksr_siteset_t& S, int64_t q, int L, int d, int col) {
    if (!S.level_stride) return S.val[(q * L + d) * 27 + col];
    const int dd = col / 27, s = col - dd * 27;
    const int64_t site = S.ncomp == 1 ? q : q / 3;
    const int64_t row = (S.row_index ? (int64_t)S.row_index[site] : site * S.ncomp) + (q - site * S.ncomp);
    return S.val[((int64_t)(d + dd - S.level_base) * S.level_stride + row) * 27 + s];
}

// writes the accumulated tile(s) of cell c: block rows whose voxel exists, the right-hand-side column, the site count
template <int NT>
__device__ __forceinline__ void cell_blocks_finalize(const AsmArgs& A, int d, int c, int lane, asm_f32x16 (&acc)[NT], int total) {
    const nksr_level_t& lv = A.hier.lv[d];
    const int T = (A.hier.depth - d) * 27;
    const int j = lane & 31, half = lane >> 5;
    if (lane == 0) A.nsites[d][c] = total;
    if (total == 0) return;
    float* out = A.blocks[d] + (int64_t)c * 27 * T;
    float* bv = A.bvec[d] + (int64_t)c * 27;
    // block row s is consumed by exactly one matrix row, the voxel at stencil slot s of this cell: rows whose
    // voxel does not exist are never read, so they are not written (about a third of the block traffic)
    const int nb = (lane < 27) ? lv.nbr[(int64_t)c * 27 + lane] : -1;
    const unsigned long long need = __ballot(nb >= 0);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = 32 * n + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (s < 27 && ((need >> s) & 1ull)) {
                if (col < T) out[s * T + col] = acc[n][r];
                else if (col == T) bv[s] = acc[n][r];
            }
        }
    }
}

// A coarse cell holds thousands of site rows (8x more per level): with one wavefront per cell the two coarsest levels of a
// tree_depth-5 chunk took 2.6 ms on ~8 k wavefronts.  The rows of such a cell are cut into PARTS -- a decision that depends on the
// cell alone: on a level that splits (asm_level_parts(d) > 1) a cell of R rows has min(level parts, R / ASM_PART_ROWS) parts of
// equal (even) length -- and its block is  sum over the parts, in order, of the part's own accumulator.  Two schedules give
// that same sum bit for bit: k_cell_blocks_part + k_cell_blocks_reduce (one wavefront per (cell, part), raw accumulator tiles
// through scratch; used when the level has few cells) and k_cell_blocks_seq (one wavefront per cell walks its parts; used when
// the level has enough cells to fill the chip, e.g. all chunks of a rank batched into one system).  The result therefore does
// not depend on how many other cells share the launch.
#define ASM_PART_ROWS 128
#define ASM_TRIP 4
__host__ __device__ __forceinline__ int asm_level_parts(int d) {          // d = 0..2: 1, 3: 4, 4: 16, 5: 64
    const int k = d >= 2 ? (1 << (2 * (d - 2))) : 1;
    return k > 64 ? 64 : k;
}
__device__ __forceinline__ int asm_cell_rows(const AsmArgs& A, int d, int c) {
    int R = 0;
    for (int si = 0; si < A.nsets; ++si) R += (A.sets[si].end[d][c] - A.sets[si].start[d][c]) * A.sets[si].ncomp;
    return R;
}
// rows [lo, hi) of part p of a cell with R rows
__device__ __forceinline__ void asm_part_range(int R, int level_parts, int p, int& lo, int& hi) {
    int np = R / ASM_PART_ROWS;
    np = np < 1 ? 1 : (np > level_parts ? level_parts : np);
    const int rpp = ((R + np - 1) / np + 1) & ~1;                        // rows per part, even (two rows per MFMA)
    lo = p * rpp;
    hi = (p + 1) * rpp < R ? (p + 1) * rpp : R;
    if (lo > R) lo = R;
    if (hi < lo) hi = lo;
}

// The operator's row list as ONE site set (level-major rows, one row per "site", no row index: nksr_amd/fields/kernel_field.py
// assemble(fused_op=...)): rows [r_lo, r_hi) of the list, same products in the same order as asm_accumulate_rows.  Every load is
// unconditional (clamped row, one pointer and one stride per lane and tile: the value of its column, or -- the lane of column T --
// the target), and the loads of the NEXT trip are requested before the products of this one: the pass ran at the latency of one
// trip after the other (~20 loads in flight per XCD against 200+ in the operator's sweep).
__device__ __forceinline__ bool asm_is_row_list(const AsmArgs& A) {
    return A.nsets == 1 && A.sets[0].level_stride > 0 && A.sets[0].ncomp == 1 && !A.sets[0].row_index;
}
template <int NT>
struct AsmLanePtr { const float* p[NT]; int mul[NT]; bool on[NT]; };
template <int NT>
__device__ __forceinline__ void asm_lm_load(const AsmLanePtr<NT>& P, int r0, int r_lo, int r_hi, int half, float (&b)[ASM_TRIP][NT]) {
#pragma unroll
    for (int u = 0; u < ASM_TRIP; ++u) {
        const int r = r0 + 2 * u + half;
        const int rc = r < r_hi ? r : r_lo;
#pragma unroll
        for (int n = 0; n < NT; ++n) b[u][n] = P.p[n][(int64_t)rc * P.mul[n]];
    }
}
template <int NT>
__device__ __forceinline__ void asm_lm_mfma(const AsmLanePtr<NT>& P, int r0, int r_hi, int half, int j, float w, float (&b)[ASM_TRIP][NT], asm_f32x16 (&acc)[NT]) {
#pragma unroll
    for (int u = 0; u < ASM_TRIP; ++u) {
        const bool ok = r0 + 2 * u + half < r_hi;
        float v[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) v[n] = (ok && P.on[n]) ? b[u][n] : 0.f;
        const float a = (j < 27) ? w * v[0] : 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[n], acc[n], 0, 0, 0);
    }
}
// lane pointers of level d's column tiles for the rows r_lo .. of ONE level-d cell (dense rows: they do not depend on the cell)
template <int NT>
__device__ __forceinline__ void asm_lm_pointers(const AsmArgs& A, int d, int r_lo, int lane, AsmLanePtr<NT>& P) {
    const nksr_siteset_t& S = A.sets[0];
    const int L = A.hier.depth;
    const int T = (L - d) * 27;
    const int j = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = 32 * n + j;
        const int dd = col / 27, sl = col - dd * 27;
        P.on[n] = col < T || (col == T && S.target);
        P.mul[n] = (col == T && S.target) ? 1 : 27;
        // (columns past T are never used but their loads are unconditional: they read level d -- the array may START at level
        // S.level_base, KernelField.assemble, and the launch's levels are >= it)
        P.p[n] = (col == T && S.target) ? S.target : S.val + (col < T ? (int64_t)(d + dd - S.level_base) * S.level_stride * 27 + sl : (int64_t)(d - S.level_base) * S.level_stride * 27);
        if (S.compact_nbr32 && !(col == T && S.target)) {
            // COMPACT rows (csrc/fused.hip, k_fz_row_sizes): the rows r_lo .. r_hi of this cell lie in ONE cell of every coarser level too;
            // that cell's block holds, per row, the slots of its existing neighbours.  A column whose neighbour does not exist (or past
            // T) reads the zero word of the array at stride 0.
            P.mul[n] = 0;
            P.p[n] = S.val;
            if (col < T) {
                const int cj = S.compact_cells[(int64_t)(d + dd) * S.level_stride + r_lo];
                if (cj >= 0) {
                    const int32_t* tb = S.compact_nbr32 + (int64_t)cj * 32;
                    const unsigned m = (unsigned)tb[31];
                    if ((m >> sl) & 1u) {
                        const int k = __popc(m);
                        P.mul[n] = k;
                        // (indexed by the ABSOLUTE row below: the pointer is taken back by the cell's first row)
                        P.p[n] = S.val + ((int64_t)tb[30] * 4 + __popc(m & ((1u << sl) - 1u)) - (int64_t)tb[28] * k);
                    }
                }
            }
        }
    }
}
template <int NT>
__device__ __forceinline__ void asm_accumulate_lm(const AsmArgs& A, int d, int r_lo, int r_hi, int lane, asm_f32x16 (&acc)[NT]) {
    if (r_hi <= r_lo) return;
    const nksr_siteset_t& S = A.sets[0];
    const int j = lane & 31, half = lane >> 5;
    AsmLanePtr<NT> P;
    asm_lm_pointers<NT>(A, d, r_lo, lane, P);
    const float w = S.weight;
    float bA[ASM_TRIP][NT], bB[ASM_TRIP][NT];
    asm_lm_load<NT>(P, r_lo, r_lo, r_hi, half, bA);
    for (int r0 = r_lo; r0 < r_hi; r0 += 4 * ASM_TRIP) {
        const int r1 = r0 + 2 * ASM_TRIP, r2 = r0 + 4 * ASM_TRIP;
        if (r1 < r_hi) asm_lm_load<NT>(P, r1, r_lo, r_hi, half, bB);
        asm_lm_mfma<NT>(P, r0, r_hi, half, j, w, bA, acc);
        if (r1 < r_hi) {
            if (r2 < r_hi) asm_lm_load<NT>(P, r2, r_lo, r_hi, half, bA);
            asm_lm_mfma<NT>(P, r1, r_hi, half, j, w, bB, acc);
        }
    }
}

// acc += the Gram products of rows [lo, hi) of cell c (row numbering: set 0's rows, then set 1's)
template <int NT>
__device__ __forceinline__ void asm_accumulate_rows(const AsmArgs& A, int d, int c, int lo, int hi, int lane, asm_f32x16 (&acc)[NT]) {
    const int L = A.hier.depth;
    const int T = (L - d) * 27;
    const int j = lane & 31, half = lane >> 5;
    int base = 0;
    for (int si = 0; si < A.nsets; ++si) {
        const nksr_siteset_t& S = A.sets[si];
        const int k0 = S.start[d][c], k1 = S.end[d][c];
        const int nrows = (k1 - k0) * S.ncomp;
        const int m_lo = lo > base ? lo - base : 0, m_hi = hi - base < nrows ? hi - base : nrows;      // this set's rows of the part
        const int64_t q0 = (int64_t)k0 * S.ncomp;
        const float w = S.weight;
        // ASM_TRIP row pairs per trip: their loads go out together (one pair per trip left the wavefront waiting on every load;
        // the pass is bound by the loads in flight -- 20 per XCD at four pairs, against 200+ in the operator's sweep)
        for (int m0 = m_lo; m0 < m_hi; m0 += 2 * ASM_TRIP) {
            float b[ASM_TRIP][NT];
#pragma unroll
            for (int u = 0; u < ASM_TRIP; ++u) {
                const int m = m0 + 2 * u + half;
                const bool valid = m < m_hi;
                const int64_t q = q0 + (valid ? m : m_lo);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = 32 * n + j;
                    float v = 0.f;
                    if (col < T) v = asm_row_value(S, q, L, d, col);
                    else if (col == T && S.target) v = S.target[q];
                    b[u][n] = valid ? v : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < ASM_TRIP; ++u) {
                const float a = (j < 27) ? w * b[u][0] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][n], acc[n], 0, 0, 0);
            }
        }
        base += nrows;
    }
}

template <int NT>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_cell_blocks_part(AsmArgs A, int d, int nsplit, float* __restrict__ scratch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const nksr_level_t& lv = A.hier.lv[d];
    const int64_t idx = (int64_t)blockIdx.x * ASM_WAVES + wave;
    if (idx >= (int64_t)lv.n * nsplit) return;
    const int c = (int)(idx / nsplit), p = (int)(idx - (int64_t)c * nsplit);
    asm_f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    int lo, hi;
    asm_part_range(asm_cell_rows(A, d, c), nsplit, p, lo, hi);
    if (asm_is_row_list(A)) asm_accumulate_lm<NT>(A, d, A.sets[0].start[d][c] + lo, A.sets[0].start[d][c] + hi, lane, acc);
    else asm_accumulate_rows<NT>(A, d, c, lo, hi, lane, acc);
    float* out = scratch + idx * (NT * 16 * 64) + lane;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(n * 16 + r) * 64] = acc[n][r];
}

template <int NT>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_cell_blocks_reduce(AsmArgs A, int d, int nsplit, const float* __restrict__ scratch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * ASM_WAVES + wave;
    if (c >= A.hier.lv[d].n) return;
    asm_f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int p = 0; p < nsplit; ++p) {
        const float* in = scratch + ((int64_t)c * nsplit + p) * (NT * 16 * 64) + lane;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] += in[(n * 16 + r) * 64];
    }
    int total = 0;
    for (int si = 0; si < A.nsets; ++si) total += A.sets[si].end[d][c] - A.sets[si].start[d][c];
    cell_blocks_finalize<NT>(A, d, c, lane, acc, total);
}

// the same sum, one wavefront per cell: part after part into a fresh accumulator, added to the total in order
template <int NT>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_cell_blocks_seq(AsmArgs A, int d, int nsplit) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * ASM_WAVES + wave;
    if (c >= A.hier.lv[d].n) return;
    asm_f32x16 tot[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[n][r] = 0.f;
    const int R = asm_cell_rows(A, d, c);
    for (int p = 0; p < nsplit; ++p) {
        int lo, hi;
        asm_part_range(R, nsplit, p, lo, hi);
        asm_f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        if (hi > lo) {
            if (asm_is_row_list(A)) asm_accumulate_lm<NT>(A, d, A.sets[0].start[d][c] + lo, A.sets[0].start[d][c] + hi, lane, acc);
            else asm_accumulate_rows<NT>(A, d, c, lo, hi, lane, acc);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[n][r] += acc[n][r];
    }
    int total = 0;
    for (int si = 0; si < A.nsets; ++si) total += A.sets[si].end[d][c] - A.sets[si].start[d][c];
    cell_blocks_finalize<NT>(A, d, c, lane, tot, total);
}

// LM: the site sets hold LEVEL-MAJOR rows (nksr_siteset_t.level_stride > 0)
template <int NT, bool LM>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_cell_blocks(AsmArgs A, int d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const nksr_level_t& lv = A.hier.lv[d];
    const int c = blockIdx.x * ASM_WAVES + wave;
    if (c >= lv.n) return;
    const int L = A.hier.depth;
    const int T = (L - d) * 27;
    const int j = lane & 31, half = lane >> 5;
    asm_f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    int total = 0;
    if (LM && asm_is_row_list(A)) {
        const int k0 = A.sets[0].start[d][c], k1 = A.sets[0].end[d][c];
        total = k1 > k0 ? k1 - k0 : 0;
        asm_accumulate_lm<NT>(A, d, k0, k1, lane, acc);
        cell_blocks_finalize<NT>(A, d, c, lane, acc, total);
        return;
    }
    for (int si = 0; si < A.nsets; ++si) {
        const nksr_siteset_t& S = A.sets[si];
        const int k0 = S.start[d][c], k1 = S.end[d][c];
        if (k0 >= k1) continue;
        total += k1 - k0;
        const int64_t q0 = (int64_t)k0 * S.ncomp;
        const int nrows = (k1 - k0) * S.ncomp;
        // The set weight multiplies the A operand.  Callers that need BITWISE symmetric same-level blocks (the
        // row fill emits same-level lower entries from the row's own frame) pass rows and targets
        // pre-multiplied by sqrt(w) and weight 1 (nksr_amd/fields/kernel_field.py does): the products
        // (sw r_s)(sw r_t) then commute exactly.  Keep this loop as it is -- variants that dropped the multiply,
        // added a second code path (88 VGPRs) or called sqrtf here all measured 20 % slower.
        const float w = S.weight;
        if (!LM) {
            // site-major rows (the assembled solve): one row pair per trip.  Fine cells hold ~3 rows; batching the loads of two
            // pairs (as below) measured 25 % slower here
            for (int m0 = 0; m0 < nrows; m0 += 2) {
                const bool valid = m0 + half < nrows;
                const int64_t q = q0 + m0 + half;
                const float* ra = S.val + (q * L + d) * 27;
                float b[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = 32 * n + j;
                    b[n] = 0.f;
                    if (valid) {
                        if (col < T) b[n] = ra[col];
                        else if (col == T && S.target) b[n] = S.target[q];
                    }
                }
                const float a = (j < 27) ? w * b[0] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
            }
            continue;
        }
        // level-major rows (coarse block of the preconditioner: cells of 60+ rows): ASM_TRIP row pairs per trip, loads issued together
        for (int m0 = 0; m0 < nrows; m0 += 2 * ASM_TRIP) {
            float b[ASM_TRIP][NT];
#pragma unroll
            for (int u = 0; u < ASM_TRIP; ++u) {
                const int m = m0 + 2 * u + half;
                const bool valid = m < nrows;
                const int64_t q = q0 + (valid ? m : 0);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int col = 32 * n + j;
                    float v = 0.f;
                    if (col < T) v = asm_row_value(S, q, L, d, col);
                    else if (col == T && S.target) v = S.target[q];
                    b[u][n] = valid ? v : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < ASM_TRIP; ++u) {
                const float a = (j < 27) ? w * b[u][0] : 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][n], acc[n], 0, 0, 0);
            }
        }
    }
    cell_blocks_finalize<NT>(A, d, c, lane, acc, total);
}

// ---- phase 2a: structure.  colmap[row slot] = column of every structural slot ---------------------------
// One wavefront walks RC_RUN Morton-consecutive rows.  Such rows share their coarse ancestors, hence the
// 5^3 column frames of the coarser levels: a frame (125 hash lookups) is fetched when the ancestor changes
// and stays in REGISTERS -- lane l owns frame slots l and l + 64, together with the number of rows of the
// run that coupled to them, which becomes ONE global integer atomic per touched column when the frame is
// retired (instead of one per structural entry).  No LDS, no barriers, no float atomics; integer
// atomics are order-independent, so the result is deterministic.
// Frame slots run x fastest (slot = (z * 5 + y) * 5 + x): the unknowns of a level are Morton-ordered with x in the lowest bit, so the
// voxels (2k, y, z), (2k + 1, y, z) have CONSECUTIVE indices and land on adjacent frame slots -- a row's entries then come out with
// consecutive columns next to each other (the SpMV's x gathers of one 64-entry group then touch fewer 64-byte segments).
// colmap encoding: upper neighbour -> column, same-level lower neighbour -> -2 - column (emitted by the
// row itself: bitwise equal to the transposed entry), diagonal / absent / non-overlapping -> -1.
#define RC_RUN 32
__global__ void __launch_bounds__(ASM_WAVES * 64) k_row_count(AsmArgs A, int32_t* __restrict__ rowcount,
                                                              int32_t* __restrict__ crosscount, int32_t* __restrict__ samelow,
                                                              int32_t* __restrict__ indeg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const nksr_hier_t& h = A.hier;
    const int L = h.depth;
    const int row0 = (blockIdx.x * ASM_WAVES + wave) * RC_RUN;
    if (row0 >= A.M) return;
    constexpr int F = NKSR_MAX_DEPTH - 1;
    int cf[F][2], nf[F][2];                 // frame dd (index dd - 1): columns and coupling counts of slots lane, lane + 64
    int kd[F], kx[F], ky[F], kz[F];         // ancestor the cached frame belongs to (kd < 0: none)
#pragma unroll
    for (int f = 0; f < F; ++f) { kd[f] = -1; kx[f] = ky[f] = kz[f] = 0; cf[f][0] = cf[f][1] = -1; nf[f][0] = nf[f][1] = 0; }
    const int r1 = lane + 64 < 125 ? lane + 64 : 124;       // second slot of the lane (lanes 61..63 idle there)
    const bool has1 = lane + 64 < 125;

    for (int row = row0; row < row0 + RC_RUN && row < A.M; ++row) {
        const int d = row_level(h, row);
        const nksr_level_t& lv0 = h.lv[d];
        const int i = row - lv0.offset;
        const int ix = lv0.ijk[i * 3], iy = lv0.ijk[i * 3 + 1], iz = lv0.ijk[i * 3 + 2];
        const int nslots = (L - d) * 125;
        int32_t* cm = A.colmap[d] + (int64_t)i * nslots;
        int cnt = 0, lower = 0, cross = 0;

        // coarser levels: (re)load the frames whose ancestor changed, retiring the old ones
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int dd = f + 1;
            if (d + dd >= L) continue;
            const int ax = ix >> dd, ay = iy >> dd, az = iz >> dd;
            if (kd[f] == d && kx[f] == ax && ky[f] == ay && kz[f] == az) continue;
            if (kd[f] >= 0) {
                if (nf[f][0] > 0) atomicAdd(&indeg[cf[f][0]], nf[f][0]);
                if (nf[f][1] > 0) atomicAdd(&indeg[cf[f][1]], nf[f][1]);
            }
            const nksr_level_t& lc = h.lv[d + dd];
            const int bias = NKSR_BIAS0 >> (d + dd);
            int j0 = hash_find(lc.hkeys, lc.hvals, lc.hcap, morton_biased(ax + lane % 5 - 2, ay + (lane / 5) % 5 - 2, az + lane / 25 - 2, bias));
            int j1 = has1 ? hash_find(lc.hkeys, lc.hvals, lc.hcap, morton_biased(ax + r1 % 5 - 2, ay + (r1 / 5) % 5 - 2, az + r1 / 25 - 2, bias)) : -1;
            cf[f][0] = j0 < 0 ? -1 : lc.offset + j0;
            cf[f][1] = j1 < 0 ? -1 : lc.offset + j1;
            nf[f][0] = nf[f][1] = 0;
            kd[f] = d; kx[f] = ax; ky[f] = ay; kz[f] = az;
        }

        // same level: the 5^3 frame through the 27-neighbour table -- slot (dx,dy,dz), |d| <= 2, is neighbour
        // (d - e) of neighbour e = clamp(d, -1, 1): two reads inside 108-byte table rows that Morton-adjacent
        // matrix rows share, instead of a random hash probe per slot; the hash is only consulted when the
        // intermediate voxel does not exist.  Every slot of this frame overlaps (|dI| <= 2).
        const int nb_lane = (lane < 27) ? lv0.nbr[(int64_t)i * 27 + lane] : -1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = u == 0 ? lane : r1;
            const bool on = u == 0 || has1;
            const int dx = r % 5 - 2, dy = (r / 5) % 5 - 2, dz = r / 25 - 2;          // frame slot r = (z, y, x), x fastest
            const int ex = dx < -1 ? -1 : (dx > 1 ? 1 : dx), ey = dy < -1 ? -1 : (dy > 1 ? 1 : dy), ez = dz < -1 ? -1 : (dz > 1 ? 1 : dz);
            const int n1 = __shfl(nb_lane, (ex + 1) * 9 + (ey + 1) * 3 + (ez + 1));
            int col = -1;
            if (on) {
                int j;
                if (n1 >= 0) j = lv0.nbr[(int64_t)n1 * 27 + (dx - ex + 1) * 9 + (dy - ey + 1) * 3 + (dz - ez + 1)];
                else j = hash_find(lv0.hkeys, lv0.hvals, lv0.hcap, morton_biased(ix + dx, iy + dy, iz + dz, NKSR_BIAS0 >> d));
                col = j < 0 ? -1 : lv0.offset + j;
            }
            const bool lo = col >= 0 && col < row;
            lower += __popcll(__ballot(lo));
            if (col == row) col = -1;
            cnt += __popcll(__ballot(col > row));
            if (on) cm[r] = lo ? -2 - col : col;
        }

        // coarser levels: integer support-overlap test  |(2I+1) - (2J+1) 2^dd| < 3 (1 + 2^dd)  on every axis
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int dd = f + 1;
            if (d + dd >= L) continue;
            const int lim = 3 * (1 + (1 << dd));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = u == 0 ? lane : r1;
                const bool on = u == 0 || has1;
                const int x = (ix >> dd) + r % 5 - 2, y = (iy >> dd) + (r / 5) % 5 - 2, z = (iz >> dd) + r / 25 - 2;
                const int ox = (2 * ix + 1) - ((2 * x + 1) << dd), oy = (2 * iy + 1) - ((2 * y + 1) << dd),
                          oz = (2 * iz + 1) - ((2 * z + 1) << dd);
                int col = -1;
                if (on && abs(ox) < lim && abs(oy) < lim && abs(oz) < lim) col = cf[f][u];
                if (on) cm[dd * 125 + r] = col;
                if (col >= 0) nf[f][u] += 1;
                cross += __popcll(__ballot(col >= 0));
            }
        }
        if (lane == 0) {
            rowcount[row] = cnt + cross;
            crosscount[row] = cross;
            samelow[row] = lower;
        }
    }
    // retire the frames that are still cached
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (kd[f] < 0) continue;
        if (nf[f][0] > 0) atomicAdd(&indeg[cf[f][0]], nf[f][0]);
        if (nf[f][1] > 0) atomicAdd(&indeg[cf[f][1]], nf[f][1]);
    }
}

// NQ consecutive floats per lane from a 4-byte aligned block row: ONE vector-memory instruction per row
// (a row fill issues ~100 of them per matrix row and each costs ~16 clocks in the CU's address unit)
template <int NQ> struct cols_u { float v[NQ]; } __attribute__((packed, aligned(4)));
template <int NQ>
__device__ __forceinline__ void load_cols(const float* __restrict__ row, int lane, int T, float out[NQ]) {
    const int t0 = NQ * lane;
    if (t0 + NQ <= T) {
        const cols_u<NQ> c = *reinterpret_cast<const cols_u<NQ>*>(row + t0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[q] = c.v[q];
    } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[q] = (t0 + q < T) ? row[t0 + q] : 0.f;
    }
}

// ---- phase 2b: one wavefront per row: gather block rows into the slot frame, emit COO ----------------
template <int NQ>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_row_fill(AsmArgs A, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ indeg, const int32_t* __restrict__ samelow,
                                                             const int32_t* __restrict__ mir_off,
                                                             int32_t* __restrict__ cols_out, float* __restrict__ vals_out,
                                                             float* __restrict__ diag_out, uint64_t* __restrict__ mir_keys,
                                                             float* __restrict__ mir_vals,
                                                             float* __restrict__ b_out, int row_begin, int row_end) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = row_begin + blockIdx.x * ASM_WAVES + wave;
    if (row >= row_end) return;
    const nksr_hier_t& h = A.hier;
    const int L = h.depth;
    const int d = row_level(h, row);
    const nksr_level_t& lv = h.lv[d];
    const int i = row - lv.offset;
    const int ix = lv.ijk[i * 3], iy = lv.ijk[i * 3 + 1], iz = lv.ijk[i * 3 + 2];
    const int nslots = (L - d) * 125;
    const int T = (L - d) * 27;
    float* acc = lds + wave * (NKSR_MAX_DEPTH * 125);
    for (int t = lane; t < nslots; t += 64) acc[t] = 0.f;

    // neighbour cells and their site counts, one per lane
    const int cme = (lane < 27) ? lv.nbr[(int64_t)i * 27 + lane] : -1;
    const int nse = (cme >= 0) ? A.nsites[d][cme] : 0;
    unsigned long long act = __ballot(nse > 0);
    float bl = (nse > 0) ? A.bvec[d][(int64_t)cme * 27 + (26 - lane)] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bl += __shfl_xor(bl, o);
    if (lane == 0) b_out[row] = bl;

    // per-lane entry decomposition (independent of the neighbour cell)
    int edd[NQ], esx[NQ], esy[NQ], esz[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {      // lane l holds the NQ consecutive block columns NQ l .. NQ l + NQ - 1 (one vector load)
        const int t = NQ * lane + q, s = t % 27;
        edd[q] = t / 27;
        esx[q] = s / 9 - 1; esy[q] = (s / 3) % 3 - 1; esz[q] = s % 3 - 1;
    }
    // neighbour cells in batches of RF_BATCH: the block rows of a batch are requested together, and the next
    // batch is requested before the current one is accumulated in LDS (ablation: the block-row loads are
    // 3.2 of this kernel's 5.7 ms and latency-, not pattern-bound; one row in flight was too few, all 27 too many)
    constexpr int RF_BATCH = 4;
    float vb[2][RF_BATCH][NQ];
    int sp[2][RF_BATCH];
    auto fetch = [&](int buf) {
#pragma unroll
        for (int u = 0; u < RF_BATCH; ++u) {
            sp[buf][u] = -1;
            if (act) {
                const int spc = __ffsll((long long)act) - 1;
                act &= act - 1;
                sp[buf][u] = spc;
                const int c = __builtin_amdgcn_readlane(cme, spc);
                load_cols<NQ>(A.blocks[d] + ((int64_t)c * 27 + (26 - spc)) * T, lane, T, vb[buf][u]);
            }
        }
    };
    auto accumulate = [&](int buf) {
#pragma unroll
        for (int u = 0; u < RF_BATCH; ++u) {
            const int spc = sp[buf][u];
            if (spc < 0) continue;
            const int cx = ix + spc / 9 - 1, cy = iy + (spc / 3) % 3 - 1, cz = iz + spc % 3 - 1;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (NQ * lane + q < T) {
                    const int dd = edd[q];
                    const int rx = ((cx >> dd) + esx[q]) - (ix >> dd) + 2, ry = ((cy >> dd) + esy[q]) - (iy >> dd) + 2,
                              rz = ((cz >> dd) + esz[q]) - (iz >> dd) + 2;
                    acc[dd * 125 + (rz * 5 + ry) * 5 + rx] += vb[buf][u][q];
                }
            }
        }
    };
    fetch(0);
    while (sp[0][0] >= 0) {
        fetch(1);
        accumulate(0);
        if (sp[1][0] < 0) break;
        fetch(0);
        accumulate(1);
    }

    // emission: structure comes from the count pass (no hashing here)
    const int32_t* cm = A.colmap[d] + (int64_t)i * nslots;
    // row = [cross-level mirrors (from finer rows)][same-level lower][own upper][diagonal].  Same-level lower and
    // own upper entries go straight to their final CSR slot; only the cross-level upper entries have a
    // mirrored copy, which goes to a list keyed by destination row, sorted afterwards
    int64_t lpos = (int64_t)rowptr[row] + indeg[row];
    int64_t opos = lpos + samelow[row];
    int64_t mpos = (int64_t)mir_off[row];
    for (int t0 = 0; t0 < nslots; t0 += 64) {
        const int t = t0 + lane;
        const int cv = (t < nslots) ? cm[t] : -1;
        const bool up = cv >= 0, low = cv <= -2, mir = up && t >= 125;
        const unsigned long long mu = __ballot(up), ml = __ballot(low), mm = __ballot(mir);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (up | low) {
            const float v = acc[t];
            const int64_t k = up ? opos + __popcll(mu & below) : lpos + __popcll(ml & below);
            const int64_t ph = csr_phys(k, A.col_format);
            cols_out[ph] = up ? cv : -2 - cv;
            vals_out[ph] = v;
            if (mir) {
                const int64_t m = mpos + __popcll(mm & below);
                mir_keys[m] = ((uint64_t)row << A.col_bits) | (uint64_t)cv;   // low bits = destination row
                mir_vals[m] = v;
            }
        }
        opos += __popcll(mu);
        lpos += __popcll(ml);
        mpos += __popcll(mm);
    }
    if (lane == 0) {
        const float dv = acc[62] + A.reg;  // slot (dd=0, rel=(2,2,2))
        const int64_t ph = csr_phys(opos, A.col_format);
        cols_out[ph] = row;
        vals_out[ph] = dv;
        diag_out[row] = dv;
    }
}

// mirrored entries, stably sorted by destination row (sources ascending): final CSR slot
__global__ void k_place_mirrors(const uint64_t* __restrict__ keys, const float* __restrict__ vals, int64_t n, int col_bits,
                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ mirptr,
                                int32_t* __restrict__ cols_out, float* __restrict__ vals_out, int fmt) {
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const uint64_t key = keys[m];
    const int r = (int)(key & (((uint64_t)1 << col_bits) - 1));
    const int src = (int)(key >> col_bits);
    const int64_t ph = csr_phys((int64_t)rowptr[r] + (m - (int64_t)mirptr[r]), fmt);
    cols_out[ph] = src;
    vals_out[ph] = vals[m];
}

static int fill_args(AsmArgs& A, const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                     void* workspace) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 0 || nsets > 3) return nksr_set_error(NKSR_ERR_ARG, "at most 3 site sets");
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    for (int s = 0; s < nsets; ++s) {
        if (!(sets[s].weight >= 0.f)) return nksr_set_error(NKSR_ERR_ARG, "site-set weights must be >= 0 (set %d: %g)", s, (double)sets[s].weight);
        A.sets[s] = sets[s];
    }
    A.nsets = nsets;
    A.M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    A.col_bits = col_bits;
    A.reg = reg;
    if (col_bits < 1 || col_bits > 32 || ((int64_t)1 << col_bits) < A.M) return nksr_set_error(NKSR_ERR_ARG, "col_bits too small");
    char* p = (char*)workspace;
    for (int d = 0; d < h->depth; ++d) {
        const size_t n = (size_t)h->lv[d].n, T = (size_t)(h->depth - d) * 27;
        A.blocks[d] = (float*)p; p += (n * 27 * T * sizeof(float) + 255) / 256 * 256;
        A.bvec[d] = (float*)p; p += (n * 27 * sizeof(float) + 255) / 256 * 256;
        A.nsites[d] = (int32_t*)p; p += (n * sizeof(int32_t) + 255) / 256 * 256;
        A.colmap[d] = (int32_t*)p; p += (n * (size_t)(h->depth - d) * 125 * sizeof(int32_t) + 255) / 256 * 256;
    }
    return NKSR_OK;
}

extern "C" size_t nksr_assemble_workspace_bytes(const nksr_hier_t* h) {
    size_t tot = 256;
    for (int d = 0; d < h->depth; ++d) {
        const size_t n = (size_t)h->lv[d].n, T = (size_t)(h->depth - d) * 27;
        tot += (n * 27 * T * sizeof(float) + 255) / 256 * 256 + (n * 27 * sizeof(float) + 255) / 256 * 256 +
               (n * sizeof(int32_t) + 255) / 256 * 256 + (n * (size_t)(h->depth - d) * 125 * sizeof(int32_t) + 255) / 256 * 256;
    }
    return tot;
}

// Levels that split (asm_level_parts) run the parallel schedule -- one wavefront per (cell, part), tiles through scratch -- while
// that gives at most ~64 k wavefronts and the scratch fits; the sequential schedule (same bits) otherwise.
static bool asm_parallel_parts(int n, int parts, size_t NT, size_t scratch_bytes) {
    return parts > 1 && (int64_t)n * parts <= 65536 && (size_t)n * parts * NT * 16 * 64 * sizeof(float) <= scratch_bytes;
}
extern "C" size_t nksr_assemble_split_bytes(const nksr_hier_t* h, int64_t total_rows) {
    (void)total_rows;
    size_t best = 0;
    for (int d = 0; d < h->depth; ++d) {
        const int n = h->lv[d].n, ns = asm_level_parts(d);
        if (n <= 0 || ns <= 1 || (int64_t)n * ns > 65536) continue;
        const size_t NT = ((size_t)(h->depth - d) * 27 + 32) / 32, b = (size_t)n * ns * NT * 16 * 64 * sizeof(float);
        if (b > best) best = b;
    }
    return best;
}

extern "C" int nksr_assemble_count(const nksr_hier_t* h, void* workspace, int32_t* rowcount, int32_t* crosscount, int32_t* samelow,
                                   int32_t* indeg, void* stream) {
    AsmArgs A;
    int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int cb = 1;
    while (((int64_t)1 << cb) < M) ++cb;
    int rc = fill_args(A, h, nullptr, 0, 0.f, cb, workspace);
    if (rc) return rc;
    if (A.M <= 0) return NKSR_OK;
    hipLaunchKernelGGL(k_row_count, dim3(nksr_blocks(A.M, ASM_WAVES * RC_RUN)), dim3(ASM_WAVES * 64), 0, (hipStream_t)stream, A, rowcount,
                       crosscount, samelow, indeg);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_assemble(const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                             void* workspace, const int32_t* rowptr, const int32_t* indeg, const int32_t* samelow,
                             const int32_t* mir_off, int col_format,
                             int32_t* cols_out, float* vals_out, float* diag_out, uint64_t* mir_keys, float* mir_vals,
                             float* b_out, void* split_scratch, size_t split_scratch_bytes, void* stream) {
    AsmArgs A;
    int rc = fill_args(A, h, sets, nsets, reg, col_bits, workspace);
    if (rc) return rc;
    if (col_format < 0 || col_format > 2) return nksr_set_error(NKSR_ERR_ARG, "col_format must be 0, 1 or 2");
    A.col_format = col_format;
    if (A.M <= 0) return NKSR_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(ASM_WAVES * 64);
    const size_t lds = (size_t)ASM_WAVES * NKSR_MAX_DEPTH * 125 * sizeof(float);
    int64_t total_rows = 0;
    bool lm = false;
    for (int si = 0; si < nsets; ++si) {
        total_rows += sets[si].n * sets[si].ncomp;
        if (si > 0 && (sets[si].level_stride != 0) != lm) return nksr_set_error(NKSR_ERR_ARG, "site sets mix site-major and level-major rows");
        if (sets[si].level_base < 0 || sets[si].level_base >= h->depth || (sets[si].level_base > 0 && (!sets[si].level_stride || sets[si].compact_nbr32)))
            return nksr_set_error(NKSR_ERR_ARG, "site set: level_base needs dense level-major rows and a level of the hierarchy");
        lm = sets[si].level_stride != 0;
        for (int d = 0; d < sets[si].level_base; ++d)
            if (h->lv[d].n > 0) return nksr_set_error(NKSR_ERR_ARG, "site set: rows start at level %d but level %d of the hierarchy has cells", (int)sets[si].level_base, d);
    }
    for (int d = 0; d < h->depth; ++d) {
        const int n = h->lv[d].n;
        if (n <= 0) continue;
        const int T = (h->depth - d) * 27, NT = (T + 32) / 32;      // column tiles incl. the right-hand-side column
        const dim3 grid(nksr_blocks(n, ASM_WAVES));
        // the part structure is a property of the level (and the cell); the schedule is picked by size -- same bits either way
        const int nsplit = asm_level_parts(d);
        if (nsplit > 1) {
            const bool par = split_scratch && asm_parallel_parts(n, nsplit, (size_t)NT, split_scratch_bytes);
            const dim3 gp(nksr_blocks((int64_t)n * nsplit, ASM_WAVES));
            float* sc = (float*)split_scratch;
#define CELLSPLIT(N) if (par) { hipLaunchKernelGGL((k_cell_blocks_part<N>), gp, blk, 0, st, A, d, nsplit, sc); \
                                hipLaunchKernelGGL((k_cell_blocks_reduce<N>), grid, blk, 0, st, A, d, nsplit, (const float*)sc); } \
                     else hipLaunchKernelGGL((k_cell_blocks_seq<N>), grid, blk, 0, st, A, d, nsplit)
            switch (NT) {
                case 1: CELLSPLIT(1); break;
                case 2: CELLSPLIT(2); break;
                case 3: CELLSPLIT(3); break;
                case 4: CELLSPLIT(4); break;
                case 5: CELLSPLIT(5); break;
                default: CELLSPLIT(6); break;
            }
        } else {
            switch (NT) {
                case 1: if (lm) hipLaunchKernelGGL((k_cell_blocks<1, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<1, false>), grid, blk, 0, st, A, d); break;
                case 2: if (lm) hipLaunchKernelGGL((k_cell_blocks<2, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<2, false>), grid, blk, 0, st, A, d); break;
                case 3: if (lm) hipLaunchKernelGGL((k_cell_blocks<3, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<3, false>), grid, blk, 0, st, A, d); break;
                case 4: if (lm) hipLaunchKernelGGL((k_cell_blocks<4, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<4, false>), grid, blk, 0, st, A, d); break;
                case 5: if (lm) hipLaunchKernelGGL((k_cell_blocks<5, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<5, false>), grid, blk, 0, st, A, d); break;
                default: if (lm) hipLaunchKernelGGL((k_cell_blocks<6, true>), grid, blk, 0, st, A, d); else hipLaunchKernelGGL((k_cell_blocks<6, false>), grid, blk, 0, st, A, d); break;
            }
        }
        NKSR_CHECK_LAUNCH();
    }
    for (int d = 0; d < h->depth; ++d) {
        const int n = h->lv[d].n;
        if (n <= 0) continue;
        const int T = (h->depth - d) * 27;
        const int r0 = h->lv[d].offset, r1 = r0 + n;
        const dim3 grid(nksr_blocks(n, ASM_WAVES));
#define ROWFILL(NQ) hipLaunchKernelGGL((k_row_fill<NQ>), grid, blk, lds, st, A, rowptr, indeg, samelow, mir_off, cols_out, vals_out, diag_out, \
                                       mir_keys, mir_vals, b_out, r0, r1)
        if (T > 128) ROWFILL(3); else if (T > 64) ROWFILL(2); else ROWFILL(1);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}

extern "C" int nksr_place_mirrors(const uint64_t* keys_sorted, const float* vals_sorted, int64_t n, int col_bits,
                                  const int32_t* rowptr, const int32_t* mirptr, int col_format, int32_t* cols_out, float* vals_out,
                                  void* stream) {
    if (n <= 0) return NKSR_OK;
    if (col_format < 0 || col_format > 2) return nksr_set_error(NKSR_ERR_ARG, "col_format must be 0, 1 or 2");
    hipLaunchKernelGGL(k_place_mirrors, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, keys_sorted, vals_sorted, n,
                       col_bits, rowptr, mirptr, cols_out, vals_out, col_format);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
