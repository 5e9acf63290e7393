// Sparse voxel hierarchy: key generation, hash build/query, 27-neighbour tables, site ranges.
// Serves SparseFeatureHierarchy.build_point_splatting / grids[d] (reference call sites
// models/nksr_net.py:57-62, models/loss.py:33-46).  All kernels are HBM-bound integer work:
// one thread per element, coalesced key streams, hash probes served from L2.
#include "common.h"
// The oracle rounds every fp32 product before it is used (numpy).  Device code contracts a * b + c into one fma by default --
// x * inv_w - centre then keeps the unrounded product, the trilinear weights move by an ulp of p and a splat whose normals nearly
// cancel amplifies that to 1e-4 in the unit target (measured in round 3) -- so contraction is off in this file; explicit fmaf stays.
#pragma clang fp contract(off)

__global__ void k_splat_keys(const float* __restrict__ xyz, int64_t n, float inv_w0, int level, int mode,
                             int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p;
    int H[3];
    for (int a = 0; a < 3; ++a) H[a] = half_index(xyz[i * 3 + a], inv_w0, p) >> level;
    int bias = NKSR_BIAS0 >> level;
    if (mode == 0) {
        int bx = (H[0] - 1) >> 1, by = (H[1] - 1) >> 1, bz = (H[2] - 1) >> 1;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            out[i * 8 + c] = morton_biased(bx + (c >> 2), by + ((c >> 1) & 1), bz + (c & 1), bias);
    } else {
        int cx = H[0] >> 1, cy = H[1] >> 1, cz = H[2] >> 1;
        for (int s = 0; s < 27; ++s)
            out[i * 27 + s] = morton_biased(cx + s / 9 - 1, cy + (s / 3) % 3 - 1, cz + s % 3 - 1, bias);
    }
}

__global__ void k_point_keys(const float* __restrict__ xyz, int64_t n, float inv_w0, int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p;
    int cx = half_index(xyz[i * 3 + 0], inv_w0, p) >> 1;
    int cy = half_index(xyz[i * 3 + 1], inv_w0, p) >> 1;
    int cz = half_index(xyz[i * 3 + 2], inv_w0, p) >> 1;
    out[i] = morton_biased(cx, cy, cz, NKSR_BIAS0);
}

__global__ void k_decode_keys(const int64_t* __restrict__ keys, int64_t n, int bias, int32_t* __restrict__ ijk) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    morton_decode_biased(keys[i], bias, x, y, z);
    ijk[i * 3 + 0] = x;
    ijk[i * 3 + 1] = y;
    ijk[i * 3 + 2] = z;
}

__global__ void k_encode_keys(const int32_t* __restrict__ ijk, int64_t n, int bias, int64_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = morton_biased(ijk[i * 3], ijk[i * 3 + 1], ijk[i * 3 + 2], bias);
}

__global__ void k_hash_build(const int64_t* __restrict__ keys, int n, int64_t* hkeys, int32_t* hvals, int hcap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t key = keys[i];
    uint32_t slot = hash_slot(key, hcap);
    for (int probe = 1; probe <= hcap; ++probe) {
        unsigned long long prev = atomicCAS((unsigned long long*)&hkeys[slot], 0xFFFFFFFFFFFFFFFFull,
                                            (unsigned long long)key);
        if (prev == 0xFFFFFFFFFFFFFFFFull || prev == (unsigned long long)key) {
            hvals[slot] = i;  // keys are unique: exactly one writer per slot
            return;
        }
        slot = hash_next(slot, probe, hcap);
    }
}

__global__ void k_hash_query(const int64_t* __restrict__ q, int64_t nq, const int64_t* __restrict__ hkeys,
                             const int32_t* __restrict__ hvals, int hcap, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    out[i] = hash_find(hkeys, hvals, hcap, q[i]);
}

// one thread per (voxel, slot): consecutive lanes write consecutive nbr entries
__global__ void k_build_nbr(const int32_t* __restrict__ ijk, int n, int bias, const int64_t* __restrict__ hkeys,
                            const int32_t* __restrict__ hvals, int hcap, int32_t* __restrict__ nbr) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * 27) return;
    int i = (int)(t / 27), s = (int)(t % 27);
    int x = ijk[i * 3] + s / 9 - 1, y = ijk[i * 3 + 1] + (s / 3) % 3 - 1, z = ijk[i * 3 + 2] + s % 3 - 1;
    nbr[t] = (s == 13) ? i : hash_find(hkeys, hvals, hcap, morton_biased(x, y, z, bias));
}

// The same table from the NEXT-COARSER level instead of 27 hash probes per voxel (round 5).  Every voxel's parent is active in the
// hierarchies this package builds, so the neighbour q = i + o of a fine voxel exists only under a coarse voxel
// that is a neighbour of parent(i): its index comes out of the parent's OWN neighbour row, and the child inside it from two small
// per-coarse-voxel tables (8-bit child mask, index of the first child: the children of a voxel are one run of the sorted fine keys,
// in octant order) -- three reads from L2-resident arrays against a probe of a hash table of 12 bytes x 4 n.  Should any voxel
// lack its parent (a foreign key list), the device flag sends every thread down the hash path: same result either way.
__global__ void k_child_tables(const int64_t* __restrict__ fkeys, int nf, const int32_t* __restrict__ parent, int32_t* __restrict__ mask,
                               int32_t* __restrict__ first, int* __restrict__ orphan) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const int p = parent[i];
    if (p < 0) { atomicOr(orphan, 1); return; }
    atomicOr(&mask[p], 1 << (int)(fkeys[i] & 7));                      // (integer OR: order-free)
    if (i == 0 || parent[i - 1] != p) first[p] = i;
}
// Round 6: one 32-lane half-wave per COARSE voxel p instead of one thread per (fine voxel, slot).  Lane s' < 27 holds p's neighbour
// N[s'] with its child mask and first child (one coalesced row + two gathers per PARENT); the children of p -- one run of the fine
// voxels, in octant order -- then take their 27 neighbours out of those registers: the neighbour of child octant o in direction s lies
// under parent slot sp(o, s), octant oct(o, s), both lane constants per octant -- two lane reads and a popcount per entry, and the
// eight children's rows leave as ONE contiguous run of 8 x 108 bytes.  (A thread per entry walked parent -> the parent's row ->
// child tables as four dependent scattered loads: 9.4 M wavefronts at 0.7 TB/s, 5.2 ms per scene step.)
__global__ void __launch_bounds__(256) k_build_nbr_parent(int n_coarse, const int32_t* __restrict__ nbr_c, const int32_t* __restrict__ mask_c,
                                                          const int32_t* __restrict__ first_c, const int* __restrict__ orphan, int32_t* __restrict__ nbr) {
    if (*orphan) return;                                       // (a fine voxel without its parent: k_build_nbr_orphans takes the hash path)
    const int p = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5), s = threadIdx.x & 31;
    if (p >= n_coarse) return;
    const int mp = mask_c[p];
    if (mp == 0) return;                                       // (no children)
    const int fp = first_c[p];
    const int N = s < 27 ? nbr_c[(int64_t)p * 27 + s] : -1;
    const int m = N >= 0 ? mask_c[N] : 0, f = N >= 0 ? first_c[N] : 0;
    const int dx = s / 9 - 1, dy = (s / 3) % 3 - 1, dz = s % 3 - 1;
    int child = fp;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (!((mp >> o) & 1)) continue;                        // (uniform over the half-wave)
        // the neighbour of child octant o = (x, y, z bits) in direction (dx, dy, dz): coordinates q in {-1 .. 2} relative to the parent's corner
        const int qx = (o & 1) + dx, qy = ((o >> 1) & 1) + dy, qz = ((o >> 2) & 1) + dz;
        const int sp = ((qx >> 1) + 1) * 9 + ((qy >> 1) + 1) * 3 + ((qz >> 1) + 1);
        const int oct = (qx & 1) | ((qy & 1) << 1) | ((qz & 1) << 2);
        const int ms = __shfl(m, sp, 32), fs = __shfl(f, sp, 32);
        int r = ((ms >> oct) & 1) ? fs + __popc(ms & ((1 << oct) - 1)) : -1;
        if (s == 13) r = child;
        if (s < 27) nbr[(int64_t)child * 27 + s] = r;
        ++child;
    }
}
// the same table through the hash when a fine voxel lacks its parent (a foreign key list): a fixed grid that has nothing to do otherwise
__global__ void __launch_bounds__(256) k_build_nbr_orphans(const int32_t* __restrict__ ijk, int n, int bias, const int64_t* __restrict__ hkeys,
                                                           const int32_t* __restrict__ hvals, int hcap, const int* __restrict__ orphan,
                                                           int32_t* __restrict__ nbr) {
    if (!*orphan) return;
    const int64_t total = (int64_t)n * 27;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t / 27), s = (int)(t % 27);
        const int x = ijk[i * 3] + s / 9 - 1, y = ijk[i * 3 + 1] + (s / 3) % 3 - 1, z = ijk[i * 3 + 2] + s % 3 - 1;
        nbr[t] = (s == 13) ? i : hash_find(hkeys, hvals, hcap, morton_biased(x, y, z, bias));
    }
}

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_site_ranges(const int64_t* __restrict__ site_keys, int64_t ns, const int64_t* __restrict__ vkeys,
                              int n, int shift, int32_t* __restrict__ start, int32_t* __restrict__ end) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t k = vkeys[i];
    start[i] = (int32_t)lower_bound_i64(site_keys, ns, k << shift);
    end[i] = (int32_t)lower_bound_i64(site_keys, ns, (k + 1) << shift);
}

__global__ void k_sorted_lookup(const int64_t* __restrict__ sorted, int64_t n, const int64_t* __restrict__ q,
                                int64_t nq, int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int64_t v = q[i];
    int64_t pos = lower_bound_i64(sorted, n, v);
    out[i] = (pos < n && sorted[pos] == v) ? (int32_t)pos : -1;
}

// rank of every element of the ASCENDING list q in the ascending list `sorted`: out[i] = #{ sorted < q[i] } (upper == 0) or
// #{ sorted <= q[i] } (upper != 0).  The ranks of a block's first and last query bound the window of `sorted` all its queries fall
// into (q is ascending): two full bisections per 256 queries, then ~8 steps inside a window the whole block shares -- a bisection
// per query from scratch is 25 dependent, uncoalesced loads (torch.searchsorted on 2e7 keys: ~10 ms; this: < 1 ms).
__device__ __forceinline__ int64_t bound_i64(const int64_t* __restrict__ a, int64_t lo, int64_t hi, int64_t v, bool upper) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t x = a[mid];
        if (upper ? (x <= v) : (x < v)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ void __launch_bounds__(256) k_rank_sorted(const int64_t* __restrict__ sorted, int64_t n, const int64_t* __restrict__ q, int64_t nq,
                                                     int upper, int32_t* __restrict__ out) {
    __shared__ int64_t win[2];
    const int64_t i0 = (int64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
    const int64_t last = (i0 + 255 < nq ? i0 + 255 : nq - 1);
    if (threadIdx.x == 0) win[0] = bound_i64(sorted, 0, n, q[i0], upper != 0);
    if (threadIdx.x == 64) win[1] = bound_i64(sorted, 0, n, q[last], upper != 0);
    __syncthreads();
    if (i < nq) out[i] = (int32_t)bound_i64(sorted, win[0], win[1], q[i], upper != 0);
}

#define LAUNCH1D(kern, n, stream, ...)                                                              \
    do {                                                                                            \
        if ((n) > 0) {                                                                              \
            hipLaunchKernelGGL(kern, dim3(nksr_blocks((n), 256)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__); \
            NKSR_CHECK_LAUNCH();                                                                    \
        }                                                                                           \
    } while (0)

extern "C" int nksr_splat_keys(const float* xyz, int64_t n, float inv_w0, int level, int mode, int64_t* keys_out,
                               void* stream) {
    if (level < 0 || level >= NKSR_MAX_DEPTH || (mode != 0 && mode != 1)) return nksr_set_error(NKSR_ERR_ARG, "bad level/mode");
    LAUNCH1D(k_splat_keys, n, stream, xyz, n, inv_w0, level, mode, keys_out);
    return NKSR_OK;
}
extern "C" int nksr_point_keys(const float* xyz, int64_t n, float inv_w0, int64_t* keys_out, void* stream) {
    LAUNCH1D(k_point_keys, n, stream, xyz, n, inv_w0, keys_out);
    return NKSR_OK;
}
extern "C" int nksr_decode_keys(const int64_t* keys, int64_t n, int level, int32_t* ijk_out, void* stream) {
    int bias = level < 0 ? NKSR_BIAS0 : (NKSR_BIAS0 >> level);
    LAUNCH1D(k_decode_keys, n, stream, keys, n, bias, ijk_out);
    return NKSR_OK;
}
extern "C" int nksr_encode_keys(const int32_t* ijk, int64_t n, int level, int64_t* keys_out, void* stream) {
    int bias = level < 0 ? NKSR_BIAS0 : (NKSR_BIAS0 >> level);
    LAUNCH1D(k_encode_keys, n, stream, ijk, n, bias, keys_out);
    return NKSR_OK;
}
extern "C" int nksr_hash_build(const int64_t* keys, int32_t n, int64_t* hkeys, int32_t* hvals, int32_t hcap, void* stream) {
    // (>= 8: a line of the table holds the eight children of a cell, common.h hash_slot / hash_next)
    if (hcap < 8 || (hcap & (hcap - 1)) || hcap < 2 * n) return nksr_set_error(NKSR_ERR_ARG, "hash capacity must be a power of two >= max(8, 2n)");
    LAUNCH1D(k_hash_build, n, stream, keys, n, hkeys, hvals, hcap);
    return NKSR_OK;
}
extern "C" int nksr_hash_query(const int64_t* q, int64_t nq, const int64_t* hkeys, const int32_t* hvals, int32_t hcap,
                               int32_t* idx_out, void* stream) {
    if (nq > 0 && (hcap < 8 || (hcap & (hcap - 1)))) return nksr_set_error(NKSR_ERR_ARG, "hash capacity must be a power of two >= 8");
    LAUNCH1D(k_hash_query, nq, stream, q, nq, hkeys, hvals, hcap, idx_out);
    return NKSR_OK;
}
extern "C" int nksr_build_nbr(const int32_t* ijk, int32_t n, int level, const int64_t* hkeys, const int32_t* hvals,
                              int32_t hcap, int32_t* nbr_out, void* stream) {
    LAUNCH1D(k_build_nbr, (int64_t)n * 27, stream, ijk, n, NKSR_BIAS0 >> level, hkeys, hvals, hcap, nbr_out);
    return NKSR_OK;
}
extern "C" int nksr_build_nbr_from_parent(const int32_t* ijk, const int64_t* keys, int32_t n, int level, const int64_t* hkeys, const int32_t* hvals,
                                          int32_t hcap, const int32_t* parent_idx, const int32_t* coarse_nbr, int32_t n_coarse, int32_t* work,
                                          int32_t* nbr_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!ijk || !keys || !hkeys || !hvals || !parent_idx || !coarse_nbr || !work || !nbr_out || n_coarse <= 0)
        return nksr_set_error(NKSR_ERR_ARG, "build_nbr_from_parent: NULL arrays");
    // work: [n_coarse] child masks (ZEROED by the caller) | [n_coarse] first child | [1] orphan flag (ZEROED by the caller)
    int32_t* mask = work;
    int32_t* first = work + n_coarse;
    int* orphan = work + 2 * (int64_t)n_coarse;
    hipLaunchKernelGGL(k_child_tables, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, n, parent_idx, mask, first, orphan);
    hipLaunchKernelGGL(k_build_nbr_parent, dim3(nksr_blocks((int64_t)n_coarse * 32, 256)), dim3(256), 0, (hipStream_t)stream, n_coarse, coarse_nbr,
                       (const int32_t*)mask, (const int32_t*)first, (const int*)orphan, nbr_out);
    hipLaunchKernelGGL(k_build_nbr_orphans, dim3(1024), dim3(256), 0, (hipStream_t)stream, ijk, n, NKSR_BIAS0 >> level, hkeys, hvals, hcap,
                       (const int*)orphan, nbr_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_site_ranges(const int64_t* site_keys, int64_t ns, const int64_t* vox_keys, int32_t n, int level,
                                int32_t* start_out, int32_t* end_out, void* stream) {
    LAUNCH1D(k_site_ranges, n, stream, site_keys, ns, vox_keys, n, 3 * level, start_out, end_out);
    return NKSR_OK;
}
extern "C" int nksr_rank_sorted(const int64_t* sorted, int64_t n, const int64_t* q, int64_t nq, int upper, int32_t* rank_out, void* stream) {
    if (nq <= 0) return NKSR_OK;
    if (n < 0 || n >= ((int64_t)1 << 31) || !q || !rank_out || (n > 0 && !sorted)) return nksr_set_error(NKSR_ERR_ARG, "rank_sorted: bad arguments");
    hipLaunchKernelGGL(k_rank_sorted, dim3(nksr_blocks(nq, 256)), dim3(256), 0, (hipStream_t)stream, sorted, n, q, nq, upper, rank_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_sorted_lookup(const int64_t* sorted, int64_t n, const int64_t* q, int64_t nq, int32_t* idx_out,
                                  void* stream) {
    LAUNCH1D(k_sorted_lookup, nq, stream, sorted, n, q, nq, idx_out);
    return NKSR_OK;
}

// ---- trilinear splat-mean of per-point features onto the voxels of one level ------------------
// Gather form (deterministic, no float atomics): voxel j visits the points of its 27 neighbour
// cells (contiguous ranges of the Morton-sorted cloud) and weights them with the hat function
// prod_a max(0, 1 - |x_a/w - (j_a + 1/2)|).  Used by the point encoder (network.encoder,
// reference call site models/nksr_net.py:73) for the input-normal skip path.
// One 32-lane half-wave per voxel, lane = neighbour cell: the 27 chains cell -> point range -> coordinates -> features run side by
// side (a thread per voxel walked them one after the other: latency-bound, 0.68 ms at 7.8e5 voxels), then one fixed-tree sum per
// channel over the lanes.
__device__ __forceinline__ double half_sum_d(double p) {      // sum over the 32 lanes of this half-wave, fixed tree
    p += __shfl_xor(p, 16, 32);
    p += __shfl_xor(p, 8, 32);
    p += __shfl_xor(p, 4, 32);
    p += __shfl_xor(p, 2, 32);
    p += __shfl_xor(p, 1, 32);
    return p;
}
__global__ void __launch_bounds__(256) k_splat_trilinear(const float* __restrict__ xyz, const float* __restrict__ feat, int C,
                                                         const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                                         const int32_t* __restrict__ nbr, const int32_t* __restrict__ ijk, int n,
                                                         float inv_w, float* __restrict__ out, float* __restrict__ wsum_out) {
    __shared__ int lk[8][SPLAT_LIST];
    __shared__ float lw[8][SPLAT_LIST];
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 5, s = threadIdx.x & 31, h = threadIdx.x >> 5;
    const bool live = j < n;
    const int jc = live ? j : n - 1;
    // fp64 accumulators: the splat of the input normals is normalised afterwards, and where the normals of a voxel's points nearly
    // cancel (both sides of a thin sheet) fp32 sums left 1e-4 of noise in the unit target (round 3: sites with |sum w n| ~ 0.1 sum w
    // differed by 7e-5 from the fp64 oracle); the products w * f are exact in fp64, so only the order of the additions is left.
    // The points that weigh at the voxel are listed first (common.h: splat_for_each_point, lane = neighbour cell), then lane =
    // channel sums them in list order.
    double acc = 0.0, wsum = 0.0;
    const float cx = (float)ijk[jc * 3] + 0.5f, cy = (float)ijk[jc * 3 + 1] + 0.5f, cz = (float)ijk[jc * 3 + 2] + 0.5f;
    const int c = (live && s < 27) ? nbr[(int64_t)jc * 27 + s] : -1;
    const int ch = s < C ? s : 0;
    splat_for_each_point(xyz, start, end, c, cx, cy, cz, inv_w, s, lk[h], lw[h], [&](const int (&kq)[4], const float (&wq)[4]) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = feat[(int64_t)kq[i] * C + ch];
#pragma unroll
        for (int i = 0; i < 4; ++i) { wsum += (double)wq[i]; acc = fma((double)wq[i], (double)f[i], acc); }
    });
    if (live && s < C) out[(int64_t)j * C + s] = (float)acc;
    if (live && s == 0) wsum_out[j] = (float)wsum;
}

extern "C" int nksr_splat_trilinear(const float* xyz_sorted, const float* feat_sorted, int C, const int32_t* start,
                                    const int32_t* end, const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w,
                                    float* out, float* wsum_out, void* stream) {
    if (C < 1 || C > 8) return nksr_set_error(NKSR_ERR_ARG, "splat supports 1..8 channels");
    LAUNCH1D(k_splat_trilinear, (int64_t)n * 32, stream, xyz_sorted, feat_sorted, C, start, end, nbr, ijk, n, inv_w, out, wsum_out);
    return NKSR_OK;
}

// ---- footprints generated from the unique CELLS that contain points (not from every point) ----------
// mode 0: the 8 voxel centres at level+1 nearest to any point inside a level-`level` cell: the level
//         (l+1) half index of a point IS its level-l cell index, so base = (I - 1) >> 1  (8 keys/cell)
// mode 1: the cell itself and its 26 neighbours at the same level                        (27 keys/cell)
__global__ void k_cell_footprint_keys(const int64_t* __restrict__ cell_keys, int64_t nc, int level, int mode,
                                      int64_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    int x, y, z;
    morton_decode_biased(cell_keys[i], NKSR_BIAS0 >> level, x, y, z);
    if (mode == 0) {
        const int bias = NKSR_BIAS0 >> (level + 1);
        const int bx = (x - 1) >> 1, by = (y - 1) >> 1, bz = (z - 1) >> 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) out[i * 8 + c] = morton_biased(bx + (c >> 2), by + ((c >> 1) & 1), bz + (c & 1), bias);
    } else {
        const int bias = NKSR_BIAS0 >> level;
        for (int s = 0; s < 27; ++s) out[i * 27 + s] = morton_biased(x + s / 9 - 1, y + (s / 3) % 3 - 1, z + s % 3 - 1, bias);
    }
}

// ---- bounding box of a cloud: exact min / max per axis (two fixed-order stages).  A non-finite coordinate anywhere turns out6[0]
// into NaN: the one readback of the box is also the "is the input finite" check (torch.isfinite(...).all() was a pass + a sync each) --
#define BB_BLOCKS 1024
__device__ __forceinline__ void bb_wave(float (&lo)[3], float (&hi)[3]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
}
__global__ void __launch_bounds__(256) k_bbox(const float* __restrict__ xyz, int64_t n, const float* __restrict__ part_in, float* __restrict__ out) {
    __shared__ float sm[4][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    if (part_in) {       // second stage: BB_BLOCKS partial boxes
        for (int i = threadIdx.x; i < BB_BLOCKS; i += 256) {
            bad = bad || part_in[i * 6] != part_in[i * 6];
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], part_in[i * 6 + a]); hi[a] = fmaxf(hi[a], part_in[i * 6 + 3 + a]); }
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)BB_BLOCKS * 256)
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = xyz[i * 3 + a]; bad = bad || !(fabsf(v) <= 3.402823466e38f); lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
    bad = __syncthreads_or(bad ? 1 : 0) != 0;
    bb_wave(lo, hi);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[threadIdx.x >> 6][a] = lo[a]; sm[threadIdx.x >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = sm[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, sm[w][a]) : fmaxf(v, sm[w][a]);
        out[(part_in ? 0 : (int64_t)blockIdx.x * 6) + a] = (bad && a == 0) ? NAN : v;
    }
}
// out6 = (min x, y, z, max x, y, z);  work: BB_BLOCKS * 6 floats
extern "C" int nksr_bbox(const float* xyz, int64_t n, float* work, float* out6, void* stream) {
    if (n <= 0 || !xyz || !work || !out6) return nksr_set_error(NKSR_ERR_ARG, "nksr_bbox: empty cloud or NULL arrays");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bbox, dim3(BB_BLOCKS), dim3(256), 0, st, xyz, n, (const float*)nullptr, work);
    hipLaunchKernelGGL(k_bbox, dim3(1), dim3(256), 0, st, xyz, n, (const float*)work, out6);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int64_t nksr_bbox_work_floats(void) { return (int64_t)BB_BLOCKS * 6; }

// ---- the same footprints with the duplicates of a workgroup's neighbourhood removed before they reach HBM -------------------
// Points and cells arrive in Morton order, so the 8 / 27 keys of consecutive elements repeat each other several times over: a
// workgroup inserts the keys of its elements into an LDS hash set and emits the distinct ones (order within the stream is free:
// the caller sorts it).  The level-0 streams shrink 4-7x, and so do the radix sort and the unique pass behind them.
#define DD_SLOTS 4096
template <int SRC>   // 0: points (k_splat_keys)   1: cells (k_cell_footprint_keys)
__global__ void __launch_bounds__(256) k_footprint_dedup(const float* __restrict__ xyz, const int64_t* __restrict__ cell_keys, int64_t n, float inv_w0,
                                                         int level, int mode, int epb, int64_t* __restrict__ out,
                                                         unsigned long long* __restrict__ counter) {
    __shared__ unsigned long long tab[DD_SLOTS];
    __shared__ int wsum[4];
    __shared__ unsigned long long gbase;
    const unsigned long long EMPTY = ~0ull;
    for (int t = threadIdx.x; t < DD_SLOTS; t += 256) tab[t] = EMPTY;
    __syncthreads();
    const int per = mode == 1 ? 27 : mode == 3 ? 1 : 8;
    const int64_t e0 = (int64_t)blockIdx.x * epb;
    const int ne = (int)(n - e0 < epb ? n - e0 : epb);
    for (int t = threadIdx.x; t < ne * per; t += 256) {
        const int e = t / per, sl = t - e * per;
        const int64_t i = e0 + e;
        int x, y, z, bias;
        if (SRC == 0) {
            float p;
            const int hx = half_index(xyz[i * 3 + 0], inv_w0, p) >> level, hy = half_index(xyz[i * 3 + 1], inv_w0, p) >> level,
                      hz = half_index(xyz[i * 3 + 2], inv_w0, p) >> level;
            bias = NKSR_BIAS0 >> level;
            if (mode == 0) { x = ((hx - 1) >> 1) + (sl >> 2); y = ((hy - 1) >> 1) + ((sl >> 1) & 1); z = ((hz - 1) >> 1) + (sl & 1); }
            else { x = (hx >> 1) + sl / 9 - 1; y = (hy >> 1) + (sl / 3) % 3 - 1; z = (hz >> 1) + sl % 3 - 1; }
        } else {
            if (mode == 3) { x = y = z = bias = 0; }                                                    // the keys themselves
            else morton_decode_biased(cell_keys[i], NKSR_BIAS0 >> level, x, y, z);
            if (mode == 3) {}
            else if (mode == 2) { bias = NKSR_BIAS0; x += sl >> 2; y += (sl >> 1) & 1; z += sl & 1; }       // lattice corners of a dual cell
            else if (mode == 0) { bias = NKSR_BIAS0 >> (level + 1); x = ((x - 1) >> 1) + (sl >> 2); y = ((y - 1) >> 1) + ((sl >> 1) & 1); z = ((z - 1) >> 1) + (sl & 1); }
            else { bias = NKSR_BIAS0 >> level; x += sl / 9 - 1; y += (sl / 3) % 3 - 1; z += sl % 3 - 1; }
        }
        const unsigned long long key = (SRC == 1 && mode == 3) ? (unsigned long long)cell_keys[e0 + t] : (unsigned long long)morton_biased(x, y, z, bias);
        unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 52);
        for (;;) {
            const unsigned long long prev = atomicCAS(&tab[h], EMPTY, key);
            if (prev == EMPTY || prev == key) break;
            h = (h + 1) & (DD_SLOTS - 1);
        }
    }
    __syncthreads();
    // ordered emission: thread t owns slots [16 t, 16 t + 16)
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < DD_SLOTS / 256; ++q) cnt += tab[threadIdx.x * (DD_SLOTS / 256) + q] != EMPTY;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) gbase = atomicAdd(counter, (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
    __syncthreads();
    int64_t pos = (int64_t)gbase + incl - cnt;
    for (int w = 0; w < wave; ++w) pos += wsum[w];
#pragma unroll
    for (int q = 0; q < DD_SLOTS / 256; ++q) {
        const unsigned long long k = tab[threadIdx.x * (DD_SLOTS / 256) + q];
        if (k != EMPTY) out[pos++] = (int64_t)k;
    }
}

// Distinct-per-workgroup footprint keys of points (xyz != NULL: nksr_splat_keys) or of cells (cell_keys != NULL:
// nksr_cell_footprint_keys; mode 2: the 8 lattice corners of dual-grid cells, nksr_cell_corner_keys; mode 3: cell_keys is the
// stream itself, any non-negative keys).  keys_out needs room for every key (n * 8 or n * 27); *count_out = number written.
extern "C" int nksr_footprint_keys_dedup(const float* xyz, const int64_t* cell_keys, int64_t n, float inv_w0, int level, int mode,
                                         int64_t* keys_out, int64_t* count_out, void* stream) {
    if ((xyz != nullptr) == (cell_keys != nullptr)) return nksr_set_error(NKSR_ERR_ARG, "exactly one of xyz / cell_keys");
    if (level < 0 || level + (cell_keys && mode == 0) >= NKSR_MAX_DEPTH || mode < 0 || mode > 3 || (mode >= 2 && (!cell_keys || level != 0)))
        return nksr_set_error(NKSR_ERR_ARG, "bad level/mode");
    if (!keys_out || !count_out) return nksr_set_error(NKSR_ERR_ARG, "NULL output");
    hipStream_t st = (hipStream_t)stream;
    NKSR_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), st));
    if (n <= 0) return NKSR_OK;
    const int epb = mode == 1 ? 64 : mode == 3 ? 2048 : 256;       // <= 2048 / 1728 keys per 4096-slot set
    const dim3 grid((unsigned)((n + epb - 1) / epb));
    if (xyz) hipLaunchKernelGGL(k_footprint_dedup<0>, grid, dim3(256), 0, st, xyz, cell_keys, n, inv_w0, level, mode, epb, keys_out, (unsigned long long*)count_out);
    else hipLaunchKernelGGL(k_footprint_dedup<1>, grid, dim3(256), 0, st, xyz, cell_keys, n, inv_w0, level, mode, epb, keys_out, (unsigned long long*)count_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_cell_footprint_keys(const int64_t* cell_keys, int64_t nc, int level, int mode, int64_t* keys_out,
                                        void* stream) {
    if (level < 0 || level + (mode == 0) >= NKSR_MAX_DEPTH || (mode != 0 && mode != 1)) return nksr_set_error(NKSR_ERR_ARG, "bad level/mode");
    LAUNCH1D(k_cell_footprint_keys, nc, stream, cell_keys, nc, level, mode, keys_out);
    return NKSR_OK;
}
