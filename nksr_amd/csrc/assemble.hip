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
//  2. row gather.  Row i (voxel at level d) adds, for each of its 27 neighbour cells c, the
//     block row B[d][c][slot of i in c's stencil] into a structured (L-d) x 5^3 slot frame in
//     LDS, then emits the upper triangle (coarser level, or same level and col > row) mirrored
//     into a COO list at offsets obtained from an exclusive scan of per-row counts.  The
//     (row,col) keys are unique, so the following radix sort yields a deterministic, exactly
//     symmetric CSR.  A slot is structural iff the column voxel exists and the two B-spline
//     supports overlap (integer test), so count and fill agree without looking at values.
#include "common.h"

#define ASM_WAVES 4

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
};

__device__ __forceinline__ int row_level(const nksr_hier_t& h, int row) {
    int d = 0;
    while (d + 1 < h.depth && row >= h.lv[d + 1].offset) ++d;
    return d;
}

__device__ __forceinline__ int rel_slot(int cx, int cy, int cz, int ix, int iy, int iz, int dd, int s) {
    int rx = ((cx >> dd) + s / 9 - 1) - (ix >> dd) + 2;
    int ry = ((cy >> dd) + (s / 3) % 3 - 1) - (iy >> dd) + 2;
    int rz = ((cz >> dd) + s % 3 - 1) - (iz >> dd) + 2;
    return dd * 125 + (rx * 5 + ry) * 5 + rz;
}

// ---- phase 1: one wavefront per (level, cell) -------------------------------------------------------
template <bool TWO_COLS>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_cell_blocks(AsmArgs A, int d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const nksr_level_t& lv = A.hier.lv[d];
    const int c = blockIdx.x * ASM_WAVES + wave;
    if (c >= lv.n) return;
    const int L = A.hier.depth;
    const int T = (L - d) * 27;
    float acc0[27], acc1[TWO_COLS ? 27 : 1];
#pragma unroll
    for (int s = 0; s < 27; ++s) { acc0[s] = 0.f; if (TWO_COLS) acc1[s] = 0.f; }
    float bacc = 0.f;
    int total = 0;
    const bool has0 = lane < T, has1 = TWO_COLS && (lane + 64 < T);
    for (int si = 0; si < A.nsets; ++si) {
        const nksr_siteset_t& S = A.sets[si];
        const int k0 = S.start[d][c], k1 = S.end[d][c];
        const int ncomp = S.ncomp;
        const float w = S.weight;
        total += k1 - k0;
        for (int k = k0; k < k1; ++k) {
            for (int a = 0; a < ncomp; ++a) {
                const float* ra = S.val + (((int64_t)k * ncomp + a) * L + d) * 27;
                const float v0 = has0 ? ra[lane] : 0.f;
                const float v1 = has1 ? ra[lane + 64] : 0.f;
                if (S.target && lane < 27) bacc = fmaf(w * v0, S.target[(int64_t)k * ncomp + a], bacc);
#pragma unroll
                for (int s = 0; s < 27; ++s) {
                    const float gs = w * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), s));
                    acc0[s] = fmaf(gs, v0, acc0[s]);
                    if (TWO_COLS) acc1[s] = fmaf(gs, v1, acc1[s]);
                }
            }
        }
    }
    if (lane == 0) A.nsites[d][c] = total;
    if (total == 0) return;
    float* out = A.blocks[d] + (int64_t)c * 27 * T;
#pragma unroll
    for (int s = 0; s < 27; ++s) {
        if (has0) out[s * T + lane] = acc0[s];
        if (has1) out[s * T + lane + 64] = acc1[s];
    }
    if (lane < 27) A.bvec[d][(int64_t)c * 27 + lane] = bacc;
}

// ---- structural test shared by count and fill -----------------------------------------------------
// slot t of row i (level d, coords ix,iy,iz): column voxel index (global) or -1
__device__ __forceinline__ int slot_column(const nksr_hier_t& h, int d, int ix, int iy, int iz, int t) {
    const int dd = t / 125, r = t % 125;
    const nksr_level_t& lc = h.lv[d + dd];
    const int x = (ix >> dd) + r / 25 - 2, y = (iy >> dd) + (r / 5) % 5 - 2, z = (iz >> dd) + r % 5 - 2;
    // B-spline supports overlap  <=>  |(2I+1) - (2J+1) 2^dd| < 3 (1 + 2^dd)  on every axis
    const int lim = 3 * (1 + (1 << dd));
    const int ax = (2 * ix + 1) - ((2 * x + 1) << dd), ay = (2 * iy + 1) - ((2 * y + 1) << dd),
              az = (2 * iz + 1) - ((2 * z + 1) << dd);
    if (abs(ax) >= lim || abs(ay) >= lim || abs(az) >= lim) return -1;
    const int j = hash_find(lc.hkeys, lc.hvals, lc.hcap, morton_biased(x, y, z, NKSR_BIAS0 >> (d + dd)));
    return j < 0 ? -1 : lc.offset + j;
}

// ---- phase 2: one wavefront per row -----------------------------------------------------------------
template <bool COUNT_ONLY>
__global__ void __launch_bounds__(ASM_WAVES * 64) k_row_gather(AsmArgs A, int32_t* __restrict__ rowcount,
                                                               const int32_t* __restrict__ rowoff,
                                                               uint64_t* __restrict__ coo_keys, float* __restrict__ coo_vals,
                                                               float* __restrict__ b_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * ASM_WAVES + wave;
    if (row >= A.M) return;
    const nksr_hier_t& h = A.hier;
    const int L = h.depth;
    const int d = row_level(h, row);
    const nksr_level_t& lv = h.lv[d];
    const int i = row - lv.offset;
    const int ix = lv.ijk[i * 3], iy = lv.ijk[i * 3 + 1], iz = lv.ijk[i * 3 + 2];
    const int nslots = (L - d) * 125;
    float* acc = lds + wave * (NKSR_MAX_DEPTH * 125);

    if (!COUNT_ONLY) {
        for (int t = lane; t < nslots; t += 64) acc[t] = 0.f;
        const int T = (L - d) * 27;
        float bsum = 0.f;
        for (int sp = 0; sp < 27; ++sp) {
            const int c = lv.nbr[(int64_t)i * 27 + sp];
            if (c < 0 || A.nsites[d][c] == 0) continue;
            const int s_i = 26 - sp;
            const int cx = ix + sp / 9 - 1, cy = iy + (sp / 3) % 3 - 1, cz = iz + sp % 3 - 1;
            const float* brow = A.blocks[d] + ((int64_t)c * 27 + s_i) * T;
            for (int t = lane; t < T; t += 64) {
                const int sl = rel_slot(cx, cy, cz, ix, iy, iz, t / 27, t % 27);
                acc[sl] += brow[t];
            }
            bsum += A.bvec[d][(int64_t)c * 27 + s_i];
        }
        if (lane == 0) b_out[row] = bsum;
    }

    int64_t wpos = COUNT_ONLY ? 0 : (int64_t)rowoff[row];
    int cnt = 0;
    for (int t0 = 0; t0 < nslots; t0 += 64) {
        const int t = t0 + lane;
        const int col = (t < nslots) ? slot_column(h, d, ix, iy, iz, t) : -1;
        const bool keep = col > row;
        const unsigned long long mask = __ballot(keep);
        if (!COUNT_ONLY && keep) {
            const int64_t pos = wpos + 2 * __popcll(mask & ((1ull << lane) - 1ull));
            const float v = acc[t];
            coo_keys[pos] = ((uint64_t)row << A.col_bits) | (uint64_t)col;
            coo_vals[pos] = v;
            coo_keys[pos + 1] = ((uint64_t)col << A.col_bits) | (uint64_t)row;
            coo_vals[pos + 1] = v;
        }
        const int n = __popcll(mask);
        wpos += 2 * n;
        cnt += n;
    }
    if (lane == 0) {
        if (COUNT_ONLY) rowcount[row] = 2 * cnt + 1;
        else {
            coo_keys[wpos] = ((uint64_t)row << A.col_bits) | (uint64_t)row;
            coo_vals[wpos] = acc[62] + A.reg;  // slot (dd=0, rel=(2,2,2))
        }
    }
}

static int fill_args(AsmArgs& A, const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                     void* workspace) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 0 || nsets > 3) return nksr_set_error(NKSR_ERR_ARG, "at most 3 site sets");
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    for (int s = 0; s < nsets; ++s) A.sets[s] = sets[s];
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
    }
    return NKSR_OK;
}

extern "C" size_t nksr_assemble_workspace_bytes(const nksr_hier_t* h) {
    size_t tot = 256;
    for (int d = 0; d < h->depth; ++d) {
        const size_t n = (size_t)h->lv[d].n, T = (size_t)(h->depth - d) * 27;
        tot += (n * 27 * T * sizeof(float) + 255) / 256 * 256 + (n * 27 * sizeof(float) + 255) / 256 * 256 +
               (n * sizeof(int32_t) + 255) / 256 * 256;
    }
    return tot;
}

extern "C" int nksr_assemble_count(const nksr_hier_t* h, int32_t* rowcount, void* stream) {
    AsmArgs A;
    int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int cb = 1;
    while (((int64_t)1 << cb) < M) ++cb;
    int rc = fill_args(A, h, nullptr, 0, 0.f, cb, nullptr);
    if (rc) return rc;
    if (A.M <= 0) return NKSR_OK;
    hipLaunchKernelGGL((k_row_gather<true>), dim3(nksr_blocks(A.M, ASM_WAVES)), dim3(ASM_WAVES * 64), 0, (hipStream_t)stream, A,
                       rowcount, (const int32_t*)nullptr, (uint64_t*)nullptr, (float*)nullptr, (float*)nullptr);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_assemble(const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                             void* workspace, const int32_t* rowoff, uint64_t* coo_keys, float* coo_vals, float* b_out,
                             void* stream) {
    AsmArgs A;
    int rc = fill_args(A, h, sets, nsets, reg, col_bits, workspace);
    if (rc) return rc;
    if (A.M <= 0) return NKSR_OK;
    hipStream_t st = (hipStream_t)stream;
    for (int d = 0; d < h->depth; ++d) {
        const int n = h->lv[d].n;
        if (n <= 0) continue;
        const int T = (h->depth - d) * 27;
        if (T > 128) return nksr_set_error(NKSR_ERR_ARG, "tree_depth - level > 4 not supported by the block kernel");
        if (T > 64) hipLaunchKernelGGL((k_cell_blocks<true>), dim3(nksr_blocks(n, ASM_WAVES)), dim3(ASM_WAVES * 64), 0, st, A, d);
        else hipLaunchKernelGGL((k_cell_blocks<false>), dim3(nksr_blocks(n, ASM_WAVES)), dim3(ASM_WAVES * 64), 0, st, A, d);
        NKSR_CHECK_LAUNCH();
    }
    size_t lds = (size_t)ASM_WAVES * NKSR_MAX_DEPTH * 125 * sizeof(float);
    hipLaunchKernelGGL((k_row_gather<false>), dim3(nksr_blocks(A.M, ASM_WAVES)), dim3(ASM_WAVES * 64), lds, st, A,
                       (int32_t*)nullptr, rowoff, coo_keys, coo_vals, b_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- sorted COO -> CSR ------------------------------------------------------------------------
__global__ void k_coo_rowptr(const uint64_t* __restrict__ keys, int64_t nnz, int M, int col_bits, int32_t* __restrict__ rowptr) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > M) return;
    uint64_t target = (uint64_t)r << col_bits;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
}

__global__ void k_coo_cols(const uint64_t* __restrict__ keys, const float* __restrict__ vals, int64_t nnz, int col_bits,
                           int32_t* __restrict__ cols, float* __restrict__ vals_out, float* __restrict__ diag) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    uint64_t key = keys[k];
    int col = (int)(key & (((uint64_t)1 << col_bits) - 1));
    int row = (int)(key >> col_bits);
    // physical layout: 256-entry tiles, logical entry m of a tile is stored at 4*(m%64) + m/64, so a
    // wavefront's 16-byte loads deliver 64 CONSECUTIVE logical entries per vector component
    // (csrc/pcg.hip: x-gather instructions then touch few cache lines)
    const int64_t m = k & 255;
    const int64_t phys = (k & ~(int64_t)255) + 4 * (m & 63) + (m >> 6);
    cols[phys] = col;
    vals_out[phys] = vals[k];
    if (row == col) diag[row] = vals[k];
}

extern "C" int nksr_coo_to_csr(const uint64_t* keys_sorted, const float* vals, int64_t nnz, int32_t M, int col_bits,
                               int32_t* rowptr, int32_t* cols, float* vals_out, float* diag, void* stream) {
    if (nnz >= ((int64_t)1 << 31)) return nksr_set_error(NKSR_ERR_CAPACITY, "nnz %lld exceeds int32 row pointers; use chunking", (long long)nnz);
    hipLaunchKernelGGL(k_coo_rowptr, dim3(nksr_blocks((int64_t)M + 1, 256)), dim3(256), 0, (hipStream_t)stream, keys_sorted, nnz, M, col_bits, rowptr);
    NKSR_CHECK_LAUNCH();
    if (nnz > 0) {
        hipLaunchKernelGGL(k_coo_cols, dim3(nksr_blocks(nnz, 256)), dim3(256), 0, (hipStream_t)stream, keys_sorted, vals, nnz, col_bits, cols, vals_out, diag);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}
