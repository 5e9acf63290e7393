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

#define FZ_RC 32
// (round 3: reading the kernel rows with the non-temporal hint made the sweep 9-13 % SLOWER -- a 108-byte row shares its cache lines
// with its neighbours in the list, which the four rows of a trip and the next trip fetch again)
#define FZ_ROW_LOAD(p) (*(p))
#define FZ_BLOCK 256

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

// partial blocks of a cell: one per 32-row item its rows touch
__global__ void k_fz_block_counts(int M, const int32_t* __restrict__ first, const int32_t* __restrict__ last, int32_t* __restrict__ counts) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > M) return;
    counts[j] = (j < M && first[j] >= 0) ? last[j] / FZ_RC - first[j] / FZ_RC + 1 : 0;
}

__device__ __forceinline__ int fz_level(const nksr_hier_t& h, int j) {
    int d = 0;
    while (d + 1 < h.depth && j >= h.lv[d + 1].offset) ++d;
    return d;
}

// nbr32[j][0..26]: global unknown index of the neighbour voxels or -1;  [27]: (first block of j) - (first item of j), so that the
// block of item i is nbr32[j][27] + i;  [28]: 1 if the cell owns exactly one block
__global__ void k_fz_tables(nksr_hier_t hier, int M, const int32_t* __restrict__ offsets, const int32_t* __restrict__ first,
                            int32_t* __restrict__ nbr32) {
    const int64_t lin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= (int64_t)M * 32) return;
    const int j = (int)(lin >> 5), s = (int)(lin & 31);
    int v = 0;
    if (s < 27) {
        const int d = fz_level(hier, j), c = j - hier.lv[d].offset;
        const int nb = hier.lv[d].nbr[(int64_t)c * 27 + s];
        v = nb >= 0 ? nb + hier.lv[d].offset : -1;
    } else if (s == 27) {
        v = offsets[j] - (first[j] >= 0 ? first[j] / FZ_RC : 0);
    } else if (s == 28) {
        v = offsets[j + 1] - offsets[j] == 1;     // the cell's only block: the operator writes it straight into the per-cell sums
    }
    nbr32[lin] = v;
}

// ---- the operator ----------------------------------------------------------------------------------------------------------
struct FusedArgs {               // uniform scalars and base pointers only
    const float* rows_all;       // [depth][rows_total][27]
    const float* targets_all;    // [rows_total]
    const int32_t* row_cells;    // [depth][rows_total]
    const int32_t* nbr32;        // [M][32]
    const int32_t* offsets;      // [M + 1] blocks of a cell
    const int32_t* multi;        // cells with more than one block: the n_big cells with more than FZ_BIG blocks first
    int n_multi, n_big, M, depth;
    int hw_total;                // half-waves = items, rounded up to whole wavefronts
    int64_t rows_total, nblocks;
    unsigned long long* nnz_counter;   // the set-up pass (MODE 1) adds the non-zero slots it sees (may be NULL)
    const int32_t* item_seg;     // [hw_total] segment of every 32-row item, [M] segment of every unknown (batched chunks; may be NULL)
    const int32_t* unknown_seg;
    const int* seg_done;         // device flags of the PCG: segment c finished <=> seg_done[c * seg_stride] != 0 (may be NULL)
    int seg_stride;
    const int* seg_done_count;   // device: finished segments so far (may be NULL); the per-cell sums and the gather look at their
    int seg_gate;                // unknowns' segments only once >= seg_gate have finished (the look-up costs ~20 % of those passes)
};
__device__ __forceinline__ bool fz_seg_done(const FusedArgs& A, const int32_t* seg_of, int64_t i) {
    return A.seg_done && seg_of && A.seg_done[(int64_t)seg_of[i] * A.seg_stride] != 0;
}
// true when the FZ_GI (four) consecutive unknowns i0 .. of this half-wave all belong to finished segments (gated, see seg_gate);
// `idx` non-NULL: the unknowns are idx[i0 ..] (the cell list of the per-cell sums).  n: valid entries from i0 on.
__device__ __forceinline__ bool fz_group_done(const FusedArgs& A, const int32_t* idx, int64_t i0, int n, int lane32, bool upper) {
    if (!A.seg_done || !A.unknown_seg || !A.seg_done_count || *A.seg_done_count < A.seg_gate) return false;
    bool dn = true;
    if (lane32 < 4 && lane32 < n) {
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

// where the block of the current cell goes: the cell's block base (then block = base + item; the base may be negative) or, in
// MODE 0 for a cell with a single block, the cell's row of the per-cell sums (direct = true, base = the cell)
template <int MODE>
__device__ __forceinline__ int fz_block_base(int nbrow, int cell, bool& direct) {
    direct = MODE == 0 && __shfl(nbrow, 28, 32) != 0;
    return direct ? cell : __shfl(nbrow, 27, 32);
}

// MODE 0: the operator (t from x).  MODE 1: the set-up pass -- right-hand side (t = target) into `part`, Jacobi diagonal
// (P2 += rows^2) into `part2`, and the count of non-zero slots, all in one sweep over the rows.
// U rows per trip.  Levels < NG (the fine ones, where a cell holds a handful of rows) fetch the stencil of EVERY row (neighbour
// row, then 27 x values: the loads of a trip go out together, nothing to decide); levels >= NG keep the stencil of their current
// cell in registers and refresh it on the rare trip that crosses a cell boundary -- that trip is processed in two parts, before
// and after the refresh.  Which rows change cell / have a cell at all is known up front as two 32-bit masks per level.
template <int MODE, int D, int U, int NG>
__global__ void __launch_bounds__(FZ_BLOCK) k_fz_sweep(FusedArgs A, const float* __restrict__ x, float* __restrict__ part,
                                                      float* __restrict__ part2, float* __restrict__ cellp, const int* __restrict__ done) {
    if (done && *done) return;
    constexpr int G = NG < D ? NG : D;                               // levels that gather per row
    const int item = (blockIdx.x * FZ_BLOCK + threadIdx.x) >> 5;
    if (item >= A.hw_total) return;                                  // whole wavefronts: hw_total is even
    const int64_t R0 = (int64_t)item * FZ_RC;
    const int64_t left = A.rows_total - R0;
    // (the rows of a segment whose conjugate gradients have finished are skipped: an item lies inside ONE segment -- segments
    // are padded to whole items -- and a half-wave without rows just idles next to its partner)
    const int nrows = (MODE == 0 && left > 0 && fz_seg_done(A, A.item_seg, item)) ? 0 : (left >= FZ_RC ? FZ_RC : (left > 0 ? (int)left : 0));
    const int s = threadIdx.x & 31;
    const bool act = s < 27, upper = (threadIdx.x & 32) != 0;
    const int sh = upper ? 32 : 0;
    // the cells (and targets) of the item's rows, one row per lane: a single coalesced load each
    int cells[D];
    unsigned chg[D], pos[D];         // bit l: row l lies in another cell than row l - 1 / lies in a cell at all
#pragma unroll
    for (int d = 0; d < D; ++d) {
        cells[d] = s < nrows ? A.row_cells[(int64_t)d * A.rows_total + R0 + s] : -1;
        const int before = __shfl_up(cells[d], 1, 32);
        chg[d] = (unsigned)(__ballot(cells[d] != (s ? before : -1)) >> sh);
        pos[d] = (unsigned)(__ballot(cells[d] >= 0) >> sh);
    }
    const float tg = (MODE == 1 && A.targets_all && s < nrows) ? A.targets_all[R0 + s] : 0.f;
    int fb[D];                       // block base of the current cell (or the cell itself: direct[d])
    bool direct[D];
    float P[D], P2[MODE == 1 ? D : 1], xs[D];
    bool have[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { fb[d] = 0; direct[d] = false; P[d] = 0.f; P2[MODE == 1 ? d : 0] = 0.f; xs[d] = 0.f; have[d] = false; }
    // coarse levels: the stencils of the item's first row, all levels in one round trip (then one more for x)
    {
        int c0[D], nb0[D];
        float x0[D];
#pragma unroll
        for (int d = G; d < D; ++d) {
            c0[d] = half_lane_i(cells[d], 0, upper);
            nb0[d] = A.nbr32[(int64_t)(c0[d] >= 0 ? c0[d] : 0) * 32 + s];
        }
#pragma unroll
        for (int d = G; d < D; ++d) x0[d] = MODE == 0 ? x[(act && nb0[d] >= 0) ? nb0[d] : 0] : 0.f;
#pragma unroll
        for (int d = G; d < D; ++d) {
            have[d] = c0[d] >= 0;
            fb[d] = fz_block_base<MODE>(nb0[d], c0[d], direct[d]);
            xs[d] = (have[d] && act && nb0[d] >= 0) ? x0[d] : 0.f;
            chg[d] &= ~1u;
        }
    }
    const int nlo = __builtin_amdgcn_readlane(nrows, 0), nhi = __builtin_amdgcn_readlane(nrows, 32);
    const int nmax = nlo > nhi ? nlo : nhi;
    int nnz = 0;                                                     // MODE 1: this lane's non-zero slots (stored entries of G and Q)
    // the neighbour rows of the fine levels are requested one trip ahead: with them in hand all loads of a trip (kernel rows, x
    // stencils, the next trip's neighbour rows) are independent -- one memory round trip per trip instead of two
    // (all loads of the row loop are UNCONDITIONAL -- clamped addresses, results masked afterwards: a load under a branch makes
    // the compiler lose count of the outstanding loads and wait for all of them)
    const int sc = act ? s : 26;
    const int lastrow = nrows > 0 ? nrows - 1 : 0;
    int nbn[U][G > 0 ? G : 1];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int d = 0; d < G; ++d) {
            const int cj = half_lane_i(cells[d], u, upper);
            const int v = A.nbr32[(int64_t)(cj >= 0 ? cj : 0) * 32 + s];
            nbn[u][d] = cj >= 0 ? v : -1;
        }
    for (int rr = 0; rr < nmax; rr += U) {
        float w[U][D], xg[U][G > 0 ? G : 1];
        int nb[U][G > 0 ? G : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = rr + u < nrows ? rr + u : lastrow;
#pragma unroll
            for (int d = 0; d < D; ++d) w[u][d] = FZ_ROW_LOAD(A.rows_all + ((int64_t)d * A.rows_total + R0 + row) * 27 + sc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int d = 0; d < G; ++d) {
                nb[u][d] = nbn[u][d];
                xg[u][d] = MODE == 0 ? x[(act && nb[u][d] >= 0) ? nb[u][d] : 0] : 0.f;      // (lanes 27.. of a neighbour row are not x indices)
            }
        {
            const int nr = rr + U < 32 ? rr + U : 0;                  // (the last trip's request is a harmless repeat)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int d = 0; d < G; ++d) {
                    const int cj = half_lane_i(cells[d], (nr + u) & 31, upper);
                    const int v = A.nbr32[(int64_t)(cj >= 0 ? cj : 0) * 32 + s];
                    nbn[u][d] = cj >= 0 ? v : -1;
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = rr + u < nrows && act;
#pragma unroll
            for (int d = 0; d < D; ++d) w[u][d] = ok ? w[u][d] : 0.f;
#pragma unroll
            for (int d = 0; d < G; ++d) xg[u][d] = (ok && nb[u][d] >= 0) ? xg[u][d] : 0.f;
            if (MODE == 1) {
#pragma unroll
                for (int d = 0; d < D; ++d) nnz += w[u][d] != 0.f ? 1 : 0;
            }
        }
        // rows of the trip that cross a cell boundary of a level >= NG: the trip is cut there
        unsigned cut = 0;
#pragma unroll
        for (int d = G; d < D; ++d) cut |= chg[d] >> rr;
        cut &= (1u << U) - 1u;
        int from = 0;
        while (true) {
            // refresh the coarse stencils that change at row `from`
            if ((cut >> from) & 1u) {
#pragma unroll
                for (int d = G; d < D; ++d)
                    if ((chg[d] >> (rr + from)) & 1u) {
                        if (have[d]) { (direct[d] ? cellp + (int64_t)fb[d] * 32 : part + ((int64_t)fb[d] + item) * 32)[s] = P[d]; if (MODE == 1) part2[((int64_t)fb[d] + item) * 32 + s] = P2[d]; }
                        P[d] = 0.f;
                        if (MODE == 1) P2[d] = 0.f;
                        xs[d] = 0.f;
                        have[d] = (pos[d] >> (rr + from)) & 1u;
                        if (have[d]) {
                            const int nbv = A.nbr32[(int64_t)__shfl(cells[d], (rr + from) & 31, 32) * 32 + s];      // `from` differs between the halves: no scalar lane read here
                            fb[d] = fz_block_base<MODE>(nbv, __shfl(cells[d], (rr + from) & 31, 32), direct[d]);
                            if (MODE == 0 && act && nbv >= 0) xs[d] = x[nbv];
                        }
                    }
                cut &= ~(1u << from);
            }
            const int to = cut ? __builtin_ctz(cut) : U;             // rows [from, to) see the same coarse cells
            float t[U];
            if (MODE == 0) {
                float prod[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    prod[u] = 0.f;
                    const bool in = u >= from && u < to;
#pragma unroll
                    for (int d = 0; d < D; ++d) prod[u] = fmaf(in ? w[u][d] : 0.f, d < G ? xg[u][d < G ? d : 0] : xs[d], prod[u]);
                }
                if (U == 4) {
                    const float r = half_sum4(prod[0], prod[1], prod[2 % U], prod[3 % U], s);
#pragma unroll
                    for (int u = 0; u < U; ++u) t[u] = half_lane_f(r, 8 * u, upper);
                } else {
                    const float r = half_sum2(prod[0], prod[1], s);
#pragma unroll
                    for (int u = 0; u < U; ++u) t[u] = half_lane_f(r, 16 * u, upper);
                }
            } else if (MODE == 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = half_lane_f(tg, (rr + u) & 31, upper);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = u >= from && u < to && rr + u < nrows;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (d < G && in && ((chg[d] >> (rr + u)) & 1u)) {      // fine levels: the block leaves with its cell
                        if (have[d]) { (direct[d] ? cellp + (int64_t)fb[d] * 32 : part + ((int64_t)fb[d] + item) * 32)[s] = P[d]; if (MODE == 1) part2[((int64_t)fb[d] + item) * 32 + s] = P2[d]; }
                        P[d] = 0.f;
                        if (MODE == 1) P2[d] = 0.f;
                        have[d] = (pos[d] >> (rr + u)) & 1u;
                        if (have[d]) fb[d] = fz_block_base<MODE>(nb[u][d < G ? d : 0], half_lane_i(cells[d], (rr + u) & 31, upper), direct[d]);
                    }
                    if (in && have[d]) {
                        P[d] = fmaf(w[u][d], t[u], P[d]);
                        if (MODE == 1) P2[d] = fmaf(w[u][d], w[u][d], P2[d]);
                    }
                }
            }
            if (to >= U) break;
            from = to;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (have[d]) { (direct[d] ? cellp + (int64_t)fb[d] * 32 : part + ((int64_t)fb[d] + item) * 32)[s] = P[d]; if (MODE == 1) part2[((int64_t)fb[d] + item) * 32 + s] = P2[d]; }
    if (MODE == 1 && A.nnz_counter) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o);
        if ((threadIdx.x & 63) == 0 && nnz) atomicAdd(A.nnz_counter, (unsigned long long)nnz);           // integer: order-free
    }
}

// C[c][s] = sum of the partial blocks of cell c (one block per 32-row item the cell's rows touch: mostly one, hundreds for a
// coarse cell), so that the gather below reads exactly one block per neighbour.  LIST: only the cells with more than one block
// (A.multi) -- inside the operator the sweep writes single-block cells straight into C.  A half-wave takes FOUR cells at once
// (lane = slot): the pass is latency-bound, the first two blocks of the four cells are requested together.  Fixed order.
#define FZ_GI 4
template <bool LIST>
__global__ void __launch_bounds__(256) k_fz_cellsum(FusedArgs A, const float* __restrict__ part, float* __restrict__ cellp,
                                                   const int* __restrict__ done) {
    if (done && *done) return;
    const int s = threadIdx.x & 31;
    if (LIST && (int)blockIdx.x < A.n_big) {
        // a coarse cell with many blocks (A.multi[0 .. n_big)): the whole workgroup sums it -- half-wave h takes blocks h, h + 8, ...
        // (one serial chain over hundreds of blocks was the critical path of the pass), then the eight partial sums in order
        __shared__ float partial[8][32];
        const int cell = A.multi[blockIdx.x], h = threadIdx.x >> 5;
        if (A.seg_done && A.unknown_seg && A.seg_done_count && *A.seg_done_count >= A.seg_gate &&
            A.seg_done[(int64_t)A.unknown_seg[cell] * A.seg_stride] != 0) return;          // uniform over the workgroup
        const int b0 = A.offsets[cell], n = A.offsets[cell + 1] - b0;
        const float* p = part + (int64_t)(b0 + h) * 32 + s;
        float acc = 0.f;
        int b = h;
        for (; b + 24 < n; b += 32, p += 1024) acc += (p[0] + p[256]) + (p[512] + p[768]);
        for (; b < n; b += 8, p += 256) acc += p[0];
        partial[h][s] = acc;
        __syncthreads();
        if (h == 0) {
            float t = partial[0][s];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += partial[k][s];
            cellp[(int64_t)cell * 32 + s] = t;
        }
        return;
    }
    const int first = LIST ? A.n_big : 0, ncell = (LIST ? A.n_multi : A.M) - first;
    const int i0 = (((LIST ? (int)blockIdx.x - A.n_big : (int)blockIdx.x) * 256 + (int)threadIdx.x) >> 5) * FZ_GI;
    if (i0 >= ncell) return;
    if (fz_group_done(A, LIST ? A.multi + first : nullptr, i0, ncell - i0, s, (threadIdx.x & 32) != 0)) return;
    int cell[FZ_GI], b0[FZ_GI], n[FZ_GI];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) {
        const int i = first + (i0 + k < ncell ? i0 + k : ncell - 1);
        cell[k] = LIST ? A.multi[i] : i;
    }
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
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) {
        acc[k] += a1[k];
        const float* p = part + (int64_t)(b0[k] + 2) * 32 + s;
        int b = 2;
        for (; b + 4 <= n[k]; b += 4, p += 128) acc[k] += (p[0] + p[32]) + (p[64] + p[96]);
        for (; b < n[k]; ++b, p += 32) acc[k] += p[0];
        if (i0 + k < ncell) cellp[(int64_t)cell[k] * 32 + s] = acc[k];
    }
}

// y_j = (MODE 0: reg x_j, 1: 0, 2: reg) + sum over the 27 neighbour cells c of j:  C[c][26 - s']
// A half-wave takes four consecutive unknowns, lane = neighbour slot; one transposing butterfly sums the four.  Workgroups are
// dealt to the XCDs round-robin by the hardware: workgroup 8 i + k takes the i-th group of the k-th EIGHTH of the unknowns, so
// that an XCD's L2 holds one contiguous (Morton-ordered) part of C instead of every XCD fetching all of it.
template <int MODE>
__global__ void __launch_bounds__(256) k_fz_gather(FusedArgs A, int per_xcd, const float* __restrict__ cellp, const float* __restrict__ x,
                                                  float reg, float* __restrict__ y, const int* __restrict__ done) {
    if (done && *done) return;
    const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
    const int local = (grp * 8 + (threadIdx.x >> 5)) * FZ_GI;        // first unknown of this half-wave inside its eighth
    if (local >= per_xcd) return;
    const int j0 = xcd * per_xcd + local;
    if (j0 >= A.M) return;
    const int sp = threadIdx.x & 31;
    if (fz_group_done(A, nullptr, j0, A.M - j0, sp, (threadIdx.x & 32) != 0)) return;
    int c[FZ_GI];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) c[k] = (sp < 27 && j0 + k < A.M) ? A.nbr32[(int64_t)(j0 + k) * 32 + sp] : -1;
    float v[FZ_GI];
#pragma unroll
    for (int k = 0; k < FZ_GI; ++k) v[k] = c[k] >= 0 ? cellp[(int64_t)c[k] * 32 + (26 - sp)] : 0.f;
    const float r = half_sum4(v[0], v[1], v[2], v[3], sp);           // lanes 8 k .. 8 k + 7 hold the total of unknown k
    const int k = sp >> 3;
    if ((sp & 7) == 0 && j0 + k < A.M) y[j0 + k] = r + (MODE == 0 ? reg * x[j0 + k] : (MODE == 2 ? reg : 0.f));
}
static void fz_gather_dims(int M, dim3& grid, int& per_xcd) {
    per_xcd = ((M + 7) / 8 + 8 * FZ_GI - 1) / (8 * FZ_GI) * (8 * FZ_GI);        // whole workgroups (8 half-waves x FZ_GI unknowns)
    grid = dim3((unsigned)(per_xcd / (8 * FZ_GI) * 8));
}

static size_t fz_align(size_t v) { return (v + 255) / 256 * 256; }

struct FusedWork { float* part; float* part2; float* cellp; };   // [nblocks][32] partial blocks (x 2: the set-up pass makes two), [M][32] per-cell sums
static size_t fz_blocks_bytes(int64_t nblocks) { return fz_align((size_t)(nblocks > 0 ? nblocks : 1) * 32 * sizeof(float)); }
extern "C" size_t nksr_fused_workspace_bytes(int64_t nblocks) { return 2 * fz_blocks_bytes(nblocks); }
static FusedWork fz_carve(const nksr_fused_op_t* op) {
    FusedWork w;
    w.part = (float*)op->workspace;
    w.part2 = (float*)((char*)op->workspace + fz_blocks_bytes(op->nblocks));
    w.cellp = op->cell_sums;
    return w;
}

extern "C" int nksr_fused_block_counts(int32_t depth, int32_t M, int64_t rows_total, const int32_t* row_cells, int32_t* span_out,
                                       int32_t* counts_out, void* stream) {
    if (depth < 1 || depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", depth);
    if (M <= 0) return NKSR_OK;
    if (rows_total < 0 || rows_total >= ((int64_t)1 << 31) - 64) return nksr_set_error(NKSR_ERR_CAPACITY, "too many kernel rows");
    if (!span_out || !counts_out || (rows_total > 0 && !row_cells)) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipStream_t st = (hipStream_t)stream;
    NKSR_CHECK_HIP(hipMemsetAsync(span_out, 0xFF, (size_t)2 * M * sizeof(int32_t), st));
    if (rows_total > 0)
        hipLaunchKernelGGL(k_fz_spans, dim3(nksr_blocks(rows_total * depth, 256)), dim3(256), 0, st, depth, rows_total, row_cells, span_out, span_out + M);
    hipLaunchKernelGGL(k_fz_block_counts, dim3(nksr_blocks((int64_t)M + 1, 256)), dim3(256), 0, st, M, (const int32_t*)span_out,
                       (const int32_t*)(span_out + M), counts_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_fused_tables(const nksr_hier_t* h, const int32_t* offsets, const int32_t* span, int32_t* nbr32_out, void* stream) {
    if (!h || h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad hierarchy");
    const int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    if (M <= 0) return NKSR_OK;
    if (!offsets || !span || !nbr32_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipLaunchKernelGGL(k_fz_tables, dim3(nksr_blocks((int64_t)M * 32, 256)), dim3(256), 0, (hipStream_t)stream, *h, M, offsets, span, nbr32_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

static int fz_args(FusedArgs& A, const nksr_fused_op_t* op) {
    if (!op) return nksr_set_error(NKSR_ERR_ARG, "operator is NULL");
    if (op->depth < 1 || op->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", op->depth);
    if (op->M > 0 && (!op->rows_all || !op->row_cells || !op->nbr32 || !op->offsets || !op->workspace || !op->cell_sums ||
                      (op->n_multi > 0 && !op->multi) || op->n_big < 0 || op->n_big > op->n_multi))
        return nksr_set_error(NKSR_ERR_ARG, "operator has NULL arrays");
    if (op->rows_total < 0 || op->rows_total >= ((int64_t)1 << 31) - 64 || op->nblocks >= ((int64_t)1 << 31) - ((int64_t)1 << 26))
        return nksr_set_error(NKSR_ERR_CAPACITY, "operator too large");
    memset(&A, 0, sizeof(A));
    A.rows_all = op->rows_all; A.targets_all = op->targets_all; A.row_cells = op->row_cells; A.nbr32 = op->nbr32; A.offsets = op->offsets;
    A.multi = op->multi; A.n_multi = op->n_multi; A.n_big = op->n_big;
    A.M = op->M; A.depth = op->depth; A.rows_total = op->rows_total; A.nblocks = op->nblocks;
    A.nnz_counter = (unsigned long long*)op->nnz_counter;
    A.item_seg = op->item_seg; A.unknown_seg = op->unknown_seg;
    const int64_t items = (op->rows_total + FZ_RC - 1) / FZ_RC;
    A.hw_total = (int)((items + 1) / 2 * 2);
    return NKSR_OK;
}

// rows per trip / per-row-gather levels.  Measured on the bench workload: (4, 1) 587 us per application,
// (2, 1) 600, (4, 2) 649, (2, 2) 647; the tree_depth-5 chunks of configs[4] rank the same way.
#define FZ_ROWS_PER_TRIP 4
#define FZ_GATHER_LEVELS 1

template <int MODE, int U, int NG>
static void fz_sweep_v(const FusedArgs& A, const float* x, float* part, float* part2, float* cellp, const int* done, hipStream_t st) {
    const dim3 grid(nksr_blocks((int64_t)A.hw_total * 32, FZ_BLOCK)), blk(FZ_BLOCK);
    switch (A.depth) {
        case 1: hipLaunchKernelGGL((k_fz_sweep<MODE, 1, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
        case 2: hipLaunchKernelGGL((k_fz_sweep<MODE, 2, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
        case 3: hipLaunchKernelGGL((k_fz_sweep<MODE, 3, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
        case 4: hipLaunchKernelGGL((k_fz_sweep<MODE, 4, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
        case 5: hipLaunchKernelGGL((k_fz_sweep<MODE, 5, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
        default: hipLaunchKernelGGL((k_fz_sweep<MODE, 6, U, NG>), grid, blk, 0, st, A, x, part, part2, cellp, done); break;
    }
}

template <int MODE>
static void fz_sweep(const FusedArgs& A, const float* x, float* part, float* part2, float* cellp, const int* done, hipStream_t st) {
    if (A.hw_total <= 0) return;
    fz_sweep_v<MODE, FZ_ROWS_PER_TRIP, FZ_GATHER_LEVELS>(A, x, part, part2, cellp, done, st);
}

static int fz_apply(const FusedArgs& A, float reg, const FusedWork& w, const float* x, float* y, const int* done, hipStream_t st) {
    dim3 gg;
    int per_xcd;
    fz_gather_dims(A.M, gg, per_xcd);
    fz_sweep<0>(A, x, w.part, nullptr, w.cellp, done, st);
    if (A.n_multi > 0)
        hipLaunchKernelGGL(k_fz_cellsum<true>, dim3(A.n_big + nksr_blocks(((int64_t)(A.n_multi - A.n_big) + FZ_GI - 1) / FZ_GI * 32, 256)), dim3(256), 0, st, A,
                           (const float*)w.part, w.cellp, done);
    hipLaunchKernelGGL((k_fz_gather<0>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.cellp, x, reg, y, done);
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
    const dim3 gm(nksr_blocks(((int64_t)A.M + FZ_GI - 1) / FZ_GI * 32, 256));
    dim3 gg;
    int per_xcd;
    fz_gather_dims(A.M, gg, per_xcd);
    const float* nof = nullptr;
    const int* nod = nullptr;
    if (b_out && !A.targets_all) return nksr_set_error(NKSR_ERR_ARG, "targets_all is NULL");
    if (!b_out && !diag_out) return NKSR_OK;
    // one sweep over the rows makes the blocks of both (and counts the stored entries)
    if (A.nnz_counter) (void)hipMemsetAsync(A.nnz_counter, 0, sizeof(unsigned long long), st);
    fz_sweep<1>(A, nof, w.part, w.part2, nullptr, nod, st);
    if (b_out) {
        hipLaunchKernelGGL(k_fz_cellsum<false>, gm, dim3(256), 0, st, A, (const float*)w.part, w.cellp, nod);
        hipLaunchKernelGGL((k_fz_gather<1>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.cellp, nof, reg, b_out, nod);
    }
    if (diag_out) {
        hipLaunchKernelGGL(k_fz_cellsum<false>, gm, dim3(256), 0, st, A, (const float*)w.part2, w.cellp, nod);
        hipLaunchKernelGGL((k_fz_gather<2>), gg, dim3(256), 0, st, A, per_xcd, (const float*)w.cellp, nof, reg, diag_out, nod);
    }
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
        // Physical: every dense slot once (zeros included), the row -> cell map, partial blocks written + read, one neighbour
        // row per block (sweep), the per-cell sums written + read and one neighbour row per unknown (gather), x and y.
        // SURVEY.md section 8d's formula prices an index per entry and both products: 2 x 8 bytes per stored entry + 12 M + 4.
        const double slots = 27.0 * A.depth * (double)A.rows_total;
        unsigned long long nnz = 0;                                  // (only reached with nksr_pcg_profile on, after a stream sync)
        if (A.nnz_counter) (void)hipMemcpy(&nnz, A.nnz_counter, sizeof(nnz), hipMemcpyDeviceToHost);
        const double stored = nnz > 0 ? (double)nnz : slots;
        *alg = 4.0 * stored + 4.0 * A.depth * (double)A.rows_total + (108.0 + 8.0) * A.M + 4.0;
        *phys = 4.0 * slots + 4.0 * A.depth * (double)A.rows_total + 3.0 * 128.0 * (double)A.nblocks + (3.0 * 128.0 + 8.0 + 12.0) * A.M;
        *survey = 2.0 * 8.0 * stored + 12.0 * A.M + 4.0;
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
