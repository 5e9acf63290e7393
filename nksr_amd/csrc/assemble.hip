// Normal-equation assembly  A = sum_s w_s R_s^T R_s + reg I,  b = sum_s w_s R_s^T t_s
// (KernelField.solve_non_fused, reference call site models/nksr_net.py:105-112).
//
// Design: gather, not scatter.  One wavefront owns one row i (a voxel
// at level d).  Every site (input point / normal sample) whose level-d cell is one of the 27
// neighbours of i touches the row; sites are Morton-sorted, so each neighbour cell is one
// contiguous range of dense-slot rows (val[site][comp][level][27]).  Lanes run over the
// (level' >= d, slot) entries of the site's row -- a contiguous, coalesced 4*(L-d)*27 byte read
// -- and accumulate  g_i * g_j  into a structured (L-d) x 5^3 slot block in LDS (no atomics:
// inside one neighbour cell distinct lanes map to distinct slots).  The upper triangle is then
// appended, mirrored, to a COO list whose (row,col) keys are unique, so the following radix
// sort yields a deterministic, exactly symmetric CSR.
#include "common.h"

#define ASM_WAVES 4
#define ASM_MAXQ 3  // ceil(5*27/64)

struct AsmArgs {
    nksr_hier_t hier;
    nksr_siteset_t sets[3];
    int nsets;
    int M;
    int col_bits;
    float reg;
};

__device__ __forceinline__ int row_level(const nksr_hier_t& h, int row) {
    int d = 0;
    while (d + 1 < h.depth && row >= h.lv[d + 1].offset) ++d;
    return d;
}

__global__ void __launch_bounds__(ASM_WAVES * 64) k_assemble(AsmArgs A, uint64_t* __restrict__ coo_keys,
                                                             float* __restrict__ coo_vals, long long cap,
                                                             unsigned long long* __restrict__ count,
                                                             float* __restrict__ b_out, int count_only) {
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
    float bsum = 0.f;

    if (!count_only) {
        for (int t = lane; t < nslots; t += 64) acc[t] = 0.f;
        const int nitems = (L - d) * 27;
        for (int si = 0; si < A.nsets; ++si) {
            const nksr_siteset_t& S = A.sets[si];
            const int ncomp = S.ncomp;
            const int64_t rowstride = (int64_t)ncomp * L * 27;
            const int32_t* st = S.start[d];
            const int32_t* en = S.end[d];
            float bset = 0.f;
            for (int sp = 0; sp < 27; ++sp) {
                const int c = lv.nbr[(int64_t)i * 27 + sp];
                if (c < 0) continue;
                const int k0 = st[c], k1 = en[c];
                if (k0 >= k1) continue;
                const int s_i = 26 - sp;
                const int cx = ix + sp / 9 - 1, cy = iy + (sp / 3) % 3 - 1, cz = iz + sp % 3 - 1;
                int slot[ASM_MAXQ];
                float reg[ASM_MAXQ];
#pragma unroll
                for (int q = 0; q < ASM_MAXQ; ++q) {
                    int t = lane + 64 * q;
                    reg[q] = 0.f;
                    slot[q] = -1;
                    if (t < nitems) {
                        int dd = t / 27, s = t % 27;
                        int rx = ((cx >> dd) + s / 9 - 1) - (ix >> dd) + 2;
                        int ry = ((cy >> dd) + (s / 3) % 3 - 1) - (iy >> dd) + 2;
                        int rz = ((cz >> dd) + s % 3 - 1) - (iz >> dd) + 2;
                        slot[q] = dd * 125 + (rx * 5 + ry) * 5 + rz;
                    }
                }
                for (int k = k0; k < k1; ++k) {
                    const float* rp = S.val + (int64_t)k * rowstride;
                    for (int a = 0; a < ncomp; ++a) {
                        const float* ra = rp + (a * L + d) * 27;
                        const float gi = ra[s_i];
#pragma unroll
                        for (int q = 0; q < ASM_MAXQ; ++q)
                            if (slot[q] >= 0) reg[q] = fmaf(gi, ra[lane + 64 * q], reg[q]);
                        if (S.target) bset = fmaf(gi, S.target[(int64_t)k * ncomp + a], bset);
                    }
                }
#pragma unroll
                for (int q = 0; q < ASM_MAXQ; ++q)
                    if (slot[q] >= 0) acc[slot[q]] = fmaf(S.weight, reg[q], acc[slot[q]]);
            }
            bsum = fmaf(S.weight, bset, bsum);
        }
        if (lane == 0) b_out[row] = bsum;
    }

    // ---- emission: upper triangle (coarser level, or same level with col > row), mirrored ----
    for (int t0 = 0; t0 < nslots; t0 += 64) {
        const int t = t0 + lane;
        bool keep = false;
        int col = -1;
        float v = 0.f;
        if (t < nslots) {
            const int dd = t / 125, r = t % 125;
            const nksr_level_t& lc = h.lv[d + dd];
            const int x = (ix >> dd) + r / 25 - 2, y = (iy >> dd) + (r / 5) % 5 - 2, z = (iz >> dd) + r % 5 - 2;
            const int j = hash_find(lc.hkeys, lc.hvals, lc.hcap, morton_biased(x, y, z, NKSR_BIAS0 >> (d + dd)));
            if (j >= 0) {
                col = lc.offset + j;
                if (count_only) keep = col > row;
                else { v = acc[t]; keep = (col > row) && (v != 0.f); }
            }
        }
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, 2ull * (unsigned long long)__popcll(mask));
        base = __shfl(base, 0);
        if (keep && !count_only) {
            const unsigned long long pos = base + 2ull * (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
            if ((long long)pos + 2 <= cap) {
                coo_keys[pos] = ((uint64_t)row << A.col_bits) | (uint64_t)col;
                coo_vals[pos] = v;
                coo_keys[pos + 1] = ((uint64_t)col << A.col_bits) | (uint64_t)row;
                coo_vals[pos + 1] = v;
            }
        }
    }
    if (lane == 0) {
        const unsigned long long pos = atomicAdd(count, 1ull);
        if (!count_only && (long long)pos + 1 <= cap) {
            coo_keys[pos] = ((uint64_t)row << A.col_bits) | (uint64_t)row;
            coo_vals[pos] = acc[62] + A.reg;  // slot (dd=0, rel=(2,2,2))
        }
    }
}

static int launch_assemble(const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                           uint64_t* coo_keys, float* coo_vals, int64_t capacity, int64_t* d_count, float* b_out,
                           int count_only, void* stream) {
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (nsets < 0 || nsets > 3) return nksr_set_error(NKSR_ERR_ARG, "at most 3 site sets");
    AsmArgs A;
    memset(&A, 0, sizeof(A));
    A.hier = *h;
    for (int s = 0; s < nsets; ++s) A.sets[s] = sets[s];
    A.nsets = nsets;
    A.M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    A.col_bits = col_bits;
    A.reg = reg;
    if (A.M <= 0) return NKSR_OK;
    if (col_bits < 1 || col_bits > 32 || ((int64_t)1 << col_bits) < A.M) return nksr_set_error(NKSR_ERR_ARG, "col_bits too small");
    size_t lds = (size_t)ASM_WAVES * NKSR_MAX_DEPTH * 125 * sizeof(float);
    hipLaunchKernelGGL(k_assemble, dim3(nksr_blocks(A.M, ASM_WAVES)), dim3(ASM_WAVES * 64), lds, (hipStream_t)stream, A,
                       coo_keys, coo_vals, (long long)capacity, (unsigned long long*)d_count, b_out, count_only);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_assemble_count(const nksr_hier_t* h, int64_t* d_count, void* stream) {
    int M = h->lv[h->depth - 1].offset + h->lv[h->depth - 1].n;
    int cb = 1;
    while (((int64_t)1 << cb) < M) ++cb;
    return launch_assemble(h, nullptr, 0, 0.f, cb, nullptr, nullptr, 0, d_count, nullptr, 1, stream);
}

extern "C" int nksr_assemble(const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                             uint64_t* coo_keys, float* coo_vals, int64_t capacity, int64_t* d_count, float* b_out,
                             void* stream) {
    return launch_assemble(h, sets, nsets, reg, col_bits, coo_keys, coo_vals, capacity, d_count, b_out, 0, stream);
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
                           int32_t* __restrict__ cols, float* __restrict__ diag) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    uint64_t key = keys[k];
    int col = (int)(key & (((uint64_t)1 << col_bits) - 1));
    int row = (int)(key >> col_bits);
    cols[k] = col;
    if (row == col) diag[row] = vals[k];
}

extern "C" int nksr_coo_to_csr(const uint64_t* keys_sorted, const float* vals, int64_t nnz, int32_t M, int col_bits,
                               int32_t* rowptr, int32_t* cols, float* diag, void* stream) {
    if (nnz >= ((int64_t)1 << 31)) return nksr_set_error(NKSR_ERR_CAPACITY, "nnz %lld exceeds int32 row pointers; use chunking", (long long)nnz);
    hipLaunchKernelGGL(k_coo_rowptr, dim3(nksr_blocks((int64_t)M + 1, 256)), dim3(256), 0, (hipStream_t)stream, keys_sorted, nnz, M, col_bits, rowptr);
    NKSR_CHECK_LAUNCH();
    if (nnz > 0) {
        hipLaunchKernelGGL(k_coo_cols, dim3(nksr_blocks(nnz, 256)), dim3(256), 0, (hipStream_t)stream, keys_sorted, vals, nnz, col_bits, cols, diag);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}
