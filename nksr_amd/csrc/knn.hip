// Grid-hash k-nearest-neighbour kernels:
//   k_knn_pca_normals  kNN-PCA normal estimation -- nksr.get_estimate_normal_preprocess_fn(knn, deg)
//                      (reference call sites examples/recons_waymo.py:36, gis_app.py:41; CPU recipe
//                      examples/recons_waymo_cpu.py:21-41)
//   k_nearest_index    nearest input point of every query -- fields.PCNNField colour lookup
//                      (examples/recons_colored_mesh.py:28)
//   k_sdf_pyramid      signed distance of arbitrary queries to an oriented cloud from their k nearest reference points -- the
//                      training ground truth ext.sdfgen.sdf_from_points (ext/sdfgen/sdf_from_points.cu:32-140 on top of the
//                      kd-tree of ext/common/kdtree_cuda.cu; call sites models/loss.py:85, dataset/av_gt_geometry.py:72):
//                      sign vote or IMLS; k_knn_mean_dist_pyramid = its adaptive_knn radius.  k <= 32:
//                      candidates sorted in registers, an octree over the grid searched in one launch (second half of this file);
//                      k_sdf_from_points / k_knn_mean_dist = the same on ONE grid by bisection (any k; what the host falls back to)
// The cloud is Morton-sorted by a uniform grid (cell size chosen by the host so that a 3^3 block
// holds a few times k points); cells are contiguous point ranges found through the voxel hash.
// EXACT selection without a per-thread heap: the k-th smallest squared distance is found by a
// 32-step bisection on the float bit pattern (positive floats order like their bit patterns), each
// step re-counting the candidates (L1/L2 resident).  fp32 distances decide everything except the few
// candidates within a couple of ulps of that radius: those are ranked by their fp64 distance, so the
// neighbour SET is the one an fp64 kd-tree search returns (an fp32-only rule swaps near-equidistant
// neighbours against it: normals off by ~1e-3 at a few points, measured in round 2).  The covariance
// is then accumulated in fp64 over that set.  One thread per query, queries in Morton order
// (neighbouring lanes scan the same cells).  (The normals keep the bisection: their queries ARE the cloud -- always near -- and
// the fp64 tie rule needs the candidates AT the k-th distance, which a fixed-size list does not hold.)
#include "common.h"

struct KnnGrid {
    const float* xyz;          // [N,3] Morton-sorted points
    const int32_t* start;      // [ncell] point range of every occupied cell
    const int32_t* end;
    const int64_t* hkeys;      // hash: cell key -> cell index
    const int32_t* hvals;
    int hcap;
    float inv_cell;            // fp32 1 / cell size
    float cell;
};

__device__ __forceinline__ void cell_of(const KnnGrid& g, const float q[3], int c[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float p;
        c[a] = half_index(q[a], g.inv_cell, p) >> 1;
    }
}

// squared length with a FIXED rounding sequence: the bisection (count_within) and the later visits of the same candidates must
// agree on every bit of it.  Left to the compiler (device code contracts a*b+c into fmas, and __fmul_rn / __fadd_rn are plain
// operators in HIP) one inlined site fused and the other did not: a neighbour AT the k-th distance was counted but not visited
// (2 - 10 % of the queries of k_sdf_from_points with large cells).  Explicit fmas cannot be re-associated.
__device__ __forceinline__ float knn_d2(float ex, float ey, float ez) {
    return fmaf(ez, ez, fmaf(ey, ey, ex * ex));
}

// number of candidates with squared distance <= r2 inside the (2R+1)^3 block around cell c
__device__ __forceinline__ int count_within(const KnnGrid& g, const float q[3], const int c[3], int R, float r2) {
    int n = 0;
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -R; dz <= R; ++dz) {
                const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                if (ci < 0) continue;
                for (int k = g.start[ci]; k < g.end[ci]; ++k) {
                    const float ex = g.xyz[k * 3] - q[0], ey = g.xyz[k * 3 + 1] - q[1], ez = g.xyz[k * 3 + 2] - q[2];
                    n += knn_d2(ex, ey, ez) <= r2;
                }
            }
    return n;
}

// smallest eigenvector of a symmetric 3x3 matrix (cyclic Jacobi, fp64)
__device__ void smallest_eigvec(double a[3][3], float nrm[3]) {
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off < 1e-30) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(a[p][q]) < 1e-300) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    nrm[0] = (float)v[0][m];
    nrm[1] = (float)v[1][m];
    nrm[2] = (float)v[2][m];
}

__global__ void __launch_bounds__(128) k_knn_pca_normals(KnnGrid g, int64_t n, int k, int max_ring,
                                                         float* __restrict__ normal, float* __restrict__ radius2,
                                                         int32_t* __restrict__ valid, const int32_t* __restrict__ todo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (todo && !todo[i]) return;          // (fallback pass: only the queries the wave-per-query kernel handed back)
    const float q[3] = {g.xyz[i * 3], g.xyz[i * 3 + 1], g.xyz[i * 3 + 2]};
    int c[3];
    cell_of(g, q, c);
    // smallest ring whose inscribed ball already holds k points (then the k nearest are inside it)
    int R = 1;
    float rmax2 = 0.f;
    bool ok = false;
    for (; R <= max_ring; ++R) {
        // the ball of radius R*cell around q lies inside the (2R+1)^3 block around q's cell
        const float rr = (float)R * g.cell;
        rmax2 = rr * rr;
        if (count_within(g, q, c, R, rmax2) >= k) { ok = true; break; }
    }
    if (!ok) {   // fewer than k points within max_ring cells: isolated point
        valid[i] = 0;
        normal[i * 3] = normal[i * 3 + 1] = 0.f;
        normal[i * 3 + 2] = 1.f;
        radius2[i] = 0.f;
        return;
    }
    // exact k-th smallest squared distance: bisection on the bit pattern
    unsigned lo = 0u, hi = __float_as_uint(rmax2);   // count(lo) may be < k, count(hi) >= k
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (count_within(g, q, c, R, __uint_as_float(mid)) >= k) hi = mid; else lo = mid + 1;
    }
    const float r2 = __uint_as_float(lo);
    // candidates with an fp32 distance well below r2 are neighbours; the ones within a few ulps of it are ranked in fp64
    const float r2_lo = r2 * (1.0f - 2e-6f), r2_hi = r2 * (1.0f + 2e-6f);
    constexpr int AMB = 12;
    double amb[AMB];
    int n_amb = 0, n_in = 0;
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -R; dz <= R; ++dz) {
                const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                if (ci < 0) continue;
                for (int kk = g.start[ci]; kk < g.end[ci]; ++kk) {
                    const float px = g.xyz[kk * 3], py = g.xyz[kk * 3 + 1], pz = g.xyz[kk * 3 + 2];
                    const float ex = px - q[0], ey = py - q[1], ez = pz - q[2];
                    const float d32 = ex * ex + ey * ey + ez * ez;
                    if (d32 < r2_lo) { ++n_in; continue; }
                    if (d32 > r2_hi) continue;
                    const double fx = (double)px - (double)q[0], fy = (double)py - (double)q[1], fz = (double)pz - (double)q[2];
                    double d64 = fx * fx + fy * fy + fz * fz;
                    if (n_amb < AMB) {          // insertion into the (tiny) ascending list
                        int j = n_amb++;
                        while (j > 0 && amb[j - 1] > d64) { amb[j] = amb[j - 1]; --j; }
                        amb[j] = d64;
                    } else if (d64 < amb[AMB - 1]) {
                        int j = AMB - 1;
                        while (j > 0 && amb[j - 1] > d64) { amb[j] = amb[j - 1]; --j; }
                        amb[j] = d64;
                    }
                }
            }
    int need = k - n_in;                         // how many of the near-threshold candidates belong to the k nearest
    if (need > n_amb) need = n_amb;
    const double thr64 = need > 0 ? amb[need - 1] : -1.0;
    // mean and covariance of the neighbour set
    double m[3] = {0, 0, 0};
    int cnt = 0;
    for (int pass = 0; pass < 2; ++pass) {
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int dx = -R; dx <= R; ++dx)
            for (int dy = -R; dy <= R; ++dy)
                for (int dz = -R; dz <= R; ++dz) {
                    const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                    if (ci < 0) continue;
                    for (int kk = g.start[ci]; kk < g.end[ci]; ++kk) {
                        const float px = g.xyz[kk * 3], py = g.xyz[kk * 3 + 1], pz = g.xyz[kk * 3 + 2];
                        const float ex = px - q[0], ey = py - q[1], ez = pz - q[2];
                        const float d32 = ex * ex + ey * ey + ez * ez;
                        if (d32 > r2_hi) continue;
                        if (d32 >= r2_lo) {
                            const double fx = (double)px - (double)q[0], fy = (double)py - (double)q[1], fz = (double)pz - (double)q[2];
                            if (fx * fx + fy * fy + fz * fz > thr64) continue;
                        }
                        if (pass == 0) { m[0] += px; m[1] += py; m[2] += pz; ++cnt; }
                        else {
                            const double d0 = px - m[0], d1 = py - m[1], d2 = pz - m[2];
                            cov[0][0] += d0 * d0; cov[0][1] += d0 * d1; cov[0][2] += d0 * d2;
                            cov[1][1] += d1 * d1; cov[1][2] += d1 * d2; cov[2][2] += d2 * d2;
                        }
                    }
                }
        if (pass == 0) { m[0] /= cnt; m[1] /= cnt; m[2] /= cnt; }
        else {
            cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
            float nv[3];
            smallest_eigvec(cov, nv);
            normal[i * 3] = nv[0]; normal[i * 3 + 1] = nv[1]; normal[i * 3 + 2] = nv[2];
        }
    }
    radius2[i] = r2;
    valid[i] = 1;
}

// ---- one WAVEFRONT per query (the default) -------------------------------------------------------------------------------------
// The thread-per-query kernel above walks ~36 times over the ~400 candidates of its block, one dependent load after the other:
// fine when a million queries hide each other's latency, 52 ms for the 10 000-point scan of examples/recons_waymo_cpu.py (40
// wavefronts on 256 CUs).  Here the 64 lanes of a wavefront share one query: the candidates of the (2R+1)^3 block are gathered ONCE
// into LDS as (squared distance, point index) -- lane = cell of the block, offsets from a wave prefix sum over the cells' point
// counts, so the candidate order is the cell order: deterministic -- and every later pass (ring count, the 32-step bit-pattern
// bisection, the fp64 ranking of the candidates within a few ulps of the k-th distance, mean, covariance) strides over that list
// with ballot / fixed-tree reductions.  Same selection rule, same r2, same neighbour SET as the thread-per-query kernel (which
// stays as the fallback for a query whose block holds more than KW_CAP candidates: `todo`).
#define KW_CAP 2048
#define KW_WAVES 4
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(64 * KW_WAVES) k_knn_pca_wave(KnnGrid g, int64_t n, int k, int max_ring, float* __restrict__ normal,
                                                               float* __restrict__ radius2, int32_t* __restrict__ valid,
                                                               int32_t* __restrict__ todo) {
    __shared__ float cd[KW_WAVES][KW_CAP];
    __shared__ int32_t ci[KW_WAVES][KW_CAP];
    __shared__ double amb[KW_WAVES][16];
    // 64.5 KB of static LDS per workgroup: written for the 160 KB per CU of gfx950 (two workgroups = eight wavefronts per CU); a
    // 64 KB-LDS part (gfx942 / gfx90a) would need KW_WAVES 2
    static_assert(sizeof(float) * KW_WAVES * KW_CAP + sizeof(int32_t) * KW_WAVES * KW_CAP + sizeof(double) * KW_WAVES * 16 <= 80 * 1024,
                  "k_knn_pca_wave: two workgroups must fit the 160 KB LDS of a gfx950 CU");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * KW_WAVES + wave;
    if (i >= n) return;
    float* d2 = cd[wave];
    int32_t* idx = ci[wave];
    const float q[3] = {g.xyz[i * 3], g.xyz[i * 3 + 1], g.xyz[i * 3 + 2]};
    int c[3];
    cell_of(g, q, c);
    int ncand = 0, R = 1;
    float rmax2 = 0.f;
    bool ok = false, overflow = false;
    for (; R <= max_ring; ++R) {
        const int side = 2 * R + 1, ncell = side * side * side;
        ncand = 0;
        for (int base = 0; base < ncell && !overflow; base += 64) {
            const int cc = base + lane;
            int s0 = 0, s1 = 0;
            if (cc < ncell) {
                const int dx = cc / (side * side) - R, dy = (cc / side) % side - R, dz = cc % side - R;
                const int cell = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                if (cell >= 0) { s0 = g.start[cell]; s1 = g.end[cell]; }
            }
            int cnt = s1 - s0, off = cnt;                          // inclusive prefix sum over the lanes (= the cells, in block order)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(off, o);
                if (lane >= o) off += t;
            }
            const int total = __shfl(off, 63);
            if (ncand + total > KW_CAP) { overflow = true; break; }
            int w = ncand + off - cnt;
            for (int p = s0; p < s1; ++p, ++w) {
                const float ex = g.xyz[(int64_t)p * 3] - q[0], ey = g.xyz[(int64_t)p * 3 + 1] - q[1], ez = g.xyz[(int64_t)p * 3 + 2] - q[2];
                d2[w] = knn_d2(ex, ey, ez);
                idx[w] = p;
            }
            ncand += total;
        }
        if (overflow) break;
        __builtin_amdgcn_wave_barrier();
        const float rr = (float)R * g.cell;
        rmax2 = rr * rr;
        int cnt = 0;
        for (int j = lane; j - lane < ncand; j += 64) cnt += __popcll(__ballot(j < ncand && d2[j] <= rmax2));
        if (cnt >= k) { ok = true; break; }
    }
    if (overflow) {                 // too many candidates for the LDS list: the thread-per-query kernel takes this query
        if (lane == 0) todo[i] = 1;
        return;
    }
    if (lane == 0) todo[i] = 0;
    if (!ok) {   // fewer than k points within max_ring cells: isolated point
        if (lane == 0) {
            valid[i] = 0;
            normal[i * 3] = normal[i * 3 + 1] = 0.f;
            normal[i * 3 + 2] = 1.f;
            radius2[i] = 0.f;
        }
        return;
    }
    // exact k-th smallest squared distance: bisection on the bit pattern
    unsigned lo = 0u, hi = __float_as_uint(rmax2);
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        const float m = __uint_as_float(mid);
        int cnt = 0;
        for (int j = lane; j - lane < ncand; j += 64) cnt += __popcll(__ballot(j < ncand && d2[j] <= m));
        if (cnt >= k) hi = mid; else lo = mid + 1;
    }
    const float r2 = __uint_as_float(lo);
    // candidates with an fp32 distance well below r2 are neighbours; the ones within a few ulps of it are ranked in fp64
    const float r2_lo = r2 * (1.0f - 2e-6f), r2_hi = r2 * (1.0f + 2e-6f);
    int n_in = 0, n_amb = 0;
    for (int j = lane; j - lane < ncand; j += 64) {
        const bool in = j < ncand;
        const float d = in ? d2[j] : 3.4e38f;
        n_in += __popcll(__ballot(in && d < r2_lo));
        const bool am = in && d >= r2_lo && d <= r2_hi;
        const unsigned long long mask = __ballot(am);
        if (am) {
            const int slot = n_amb + __popcll(mask & ((1ull << lane) - 1ull));
            if (slot < 16) {
                const int p = idx[j];
                const double fx = (double)g.xyz[(int64_t)p * 3] - (double)q[0], fy = (double)g.xyz[(int64_t)p * 3 + 1] - (double)q[1],
                             fz = (double)g.xyz[(int64_t)p * 3 + 2] - (double)q[2];
                amb[wave][slot] = fx * fx + fy * fy + fz * fz;
            }
        }
        n_amb += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    if (n_amb > 16) n_amb = 16;                  // (more than 16 candidates within 2e-6 of the k-th distance: degenerate input)
    int need = k - n_in;
    if (need > n_amb) need = n_amb;
    double thr64 = -1.0;
    if (need > 0) {                              // the need-th smallest of <= 16 values: every lane ranks them (uniform result)
        double best = -1.0;
        for (int a = 0; a < n_amb; ++a) {
            const double v = amb[wave][a];
            int rank = 0;
            for (int b = 0; b < n_amb; ++b) rank += (amb[wave][b] < v) || (amb[wave][b] == v && b < a);
            if (rank == need - 1) best = v;
        }
        thr64 = best;
    }
    // mean and covariance of the neighbour set (fp64, lane-strided partial sums + a fixed reduction tree)
    double m[3] = {0, 0, 0};
    int cntn = 0;
    for (int pass = 0; pass < 2; ++pass) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
        int cl = 0;
        for (int j = lane; j < ncand; j += 64) {
            const float d = d2[j];
            if (d > r2_hi) continue;
            const int p = idx[j];
            const float px = g.xyz[(int64_t)p * 3], py = g.xyz[(int64_t)p * 3 + 1], pz = g.xyz[(int64_t)p * 3 + 2];
            if (d >= r2_lo) {
                const double fx = (double)px - (double)q[0], fy = (double)py - (double)q[1], fz = (double)pz - (double)q[2];
                if (fx * fx + fy * fy + fz * fz > thr64) continue;
            }
            if (pass == 0) { a0 += px; a1 += py; a2 += pz; ++cl; }
            else {
                const double e0 = px - m[0], e1 = py - m[1], e2 = pz - m[2];
                a0 += e0 * e0; a1 += e0 * e1; a2 += e0 * e2; a3 += e1 * e1; a4 += e1 * e2; a5 += e2 * e2;
            }
        }
        a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
        if (pass == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cl += __shfl_xor(cl, o);
            cntn = cl;
            m[0] = a0 / cntn; m[1] = a1 / cntn; m[2] = a2 / cntn;
        } else {
            a3 = wave_sum_d(a3); a4 = wave_sum_d(a4); a5 = wave_sum_d(a5);
            if (lane == 0) {
                double cov[3][3] = {{a0, a1, a2}, {a1, a3, a4}, {a2, a4, a5}};
                float nv[3];
                smallest_eigvec(cov, nv);
                normal[i * 3] = nv[0]; normal[i * 3 + 1] = nv[1]; normal[i * 3 + 2] = nv[2];
                radius2[i] = r2;
                valid[i] = 1;
            }
        }
    }
}

__global__ void __launch_bounds__(128) k_nearest_index(KnnGrid g, const float* __restrict__ query, int64_t nq, int max_ring,
                                                       int32_t* __restrict__ index) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float q[3] = {query[i * 3], query[i * 3 + 1], query[i * 3 + 2]};
    int c[3];
    cell_of(g, q, c);
    int best = -1;
    float bd = 3.4e38f;
    for (int R = 1; R <= max_ring; ++R) {
        // scan the shell |offset|_inf == R (R = 1: the full 3^3 block)
        for (int dx = -R; dx <= R; ++dx)
            for (int dy = -R; dy <= R; ++dy)
                for (int dz = -R; dz <= R; ++dz) {
                    if (R > 1 && abs(dx) < R && abs(dy) < R && abs(dz) < R) continue;
                    const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                    if (ci < 0) continue;
                    for (int k = g.start[ci]; k < g.end[ci]; ++k) {
                        const float ex = g.xyz[k * 3] - q[0], ey = g.xyz[k * 3 + 1] - q[1], ez = g.xyz[k * 3 + 2] - q[2];
                        const float d = ex * ex + ey * ey + ez * ez;
                        if (d < bd || (d == bd && k < best)) { bd = d; best = k; }
                    }
                }
        // everything closer than R*cell has been seen
        const float rr = (float)R * g.cell;
        if (best >= 0 && bd <= rr * rr) break;
    }
    index[i] = best;
}

// ---- ext.sdfgen.sdf_from_points ---------------------------------------------------------------------------------------------
// Ring search + bit-pattern bisection as above give the k-th smallest squared distance r2 of a query; the neighbour set is
// every candidate closer than r2 plus, of the ones AT r2, as many as still fit (scan order): exactly k, like a kd-tree search
// (ties have measure zero on real data).  The two estimators then need one pass (vote) or two (IMLS) over that set -- no index
// lists, no sort.
//   vote (ComputeSDFKernel, sdf_from_points.cu:83-140): the nearest neighbour decides the magnitude -- |n.(x-p)| if x lies within
//        stdv * ref_std[p] of it, else |x-p| -- and the majority of sign(n_k.(x-p_k)) over the k neighbours the sign
//        (positive needs MORE than k/2 votes);
//   IMLS (ComputeIMLSKernel, :32-81): sum_k w_k n_k.(x-p_k) / sum_k w_k,  w_k = exp(-(|x-p_k|^2 - min_j |x-p_j|^2) / stdv^2).
// valid = 0: fewer than k reference points within max_ring cells (the host retries those queries on a coarser grid).
// candidates of the (2R+1)^3 block strictly closer than r2
__device__ __forceinline__ int count_closer(const KnnGrid& g, const float q[3], const int c[3], int R, float r2) {
    int n = 0;
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -R; dz <= R; ++dz) {
                const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                if (ci < 0) continue;
                for (int kk = g.start[ci]; kk < g.end[ci]; ++kk) {
                    const float ex = g.xyz[kk * 3] - q[0], ey = g.xyz[kk * 3 + 1] - q[1], ez = g.xyz[kk * 3 + 2] - q[2];
                    n += knn_d2(ex, ey, ez) < r2;
                }
            }
    return n;
}

template <typename F>
__device__ __forceinline__ void knn_visit(const KnnGrid& g, const float q[3], const int c[3], int R, float r2, int k, F f) {
    int ties_left = k - count_closer(g, q, c, R, r2);        // of the candidates AT r2, as many as still fit
    for (int dx = -R; dx <= R; ++dx)
        for (int dy = -R; dy <= R; ++dy)
            for (int dz = -R; dz <= R; ++dz) {
                const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(c[0] + dx, c[1] + dy, c[2] + dz, NKSR_BIAS0));
                if (ci < 0) continue;
                for (int kk = g.start[ci]; kk < g.end[ci]; ++kk) {
                    const float px = g.xyz[kk * 3], py = g.xyz[kk * 3 + 1], pz = g.xyz[kk * 3 + 2];
                    const float d2 = knn_d2(px - q[0], py - q[1], pz - q[2]);
                    bool take = d2 < r2;
                    if (!take && d2 == r2 && ties_left > 0) { take = true; --ties_left; }
                    if (take) f(kk, q[0] - px, q[1] - py, q[2] - pz, d2);
                }
            }
}

__device__ __forceinline__ bool knn_radius(const KnnGrid& g, const float q[3], const int c[3], int k, int max_ring, int& R, float& r2) {
    float rmax2 = 0.f;
    bool ok = false;
    for (R = 1; R <= max_ring; ++R) {
        const float rr = (float)R * g.cell;          // the ball of radius R*cell around q lies inside the (2R+1)^3 block around q's cell
        rmax2 = rr * rr;
        if (count_within(g, q, c, R, rmax2) >= k) { ok = true; break; }
    }
    if (!ok) return false;
    unsigned lo = 0u, hi = __float_as_uint(rmax2);
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        if (count_within(g, q, c, R, __uint_as_float(mid)) >= k) hi = mid; else lo = mid + 1;
    }
    r2 = __uint_as_float(lo);
    return true;
}

__global__ void __launch_bounds__(128) k_sdf_from_points(KnnGrid g, const float* __restrict__ nrm, const float* __restrict__ ref_std,
                                                         const float* __restrict__ query, int64_t nq, int k, int max_ring, float stdv,
                                                         int imls, float* __restrict__ sdf, float* __restrict__ grad,
                                                         int32_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float q[3] = {query[i * 3], query[i * 3 + 1], query[i * 3 + 2]};
    int c[3], R;
    float r2;
    cell_of(g, q, c);
    if (!knn_radius(g, q, c, k, max_ring, R, r2)) { valid[i] = 0; return; }
    valid[i] = 1;
    float out = 0.f, gr[3] = {0.f, 0.f, 0.f};
    if (imls) {
        float dmin = 3.4e38f;
        knn_visit(g, q, c, R, r2, k, [&](int, float, float, float, float d2) { dmin = fminf(dmin, d2); });
        const float inv_s2 = 1.f / (stdv * stdv);
        const float emin = dmin * inv_s2;
        float wsum = 0.f, acc = 0.f;
        knn_visit(g, q, c, R, r2, k, [&](int kk, float ex, float ey, float ez, float d2) {
            const float nx = nrm[kk * 3], ny = nrm[kk * 3 + 1], nz = nrm[kk * 3 + 2];
            const float w = expf(-d2 * inv_s2 + emin);
            wsum += w;
            acc += (nx * ex + ny * ey + nz * ez) * w;
            gr[0] += nx * w; gr[1] += ny * w; gr[2] += nz * w;
        });
        out = acc / wsum;
        gr[0] /= wsum; gr[1] /= wsum; gr[2] /= wsum;
    } else {
        float dbest = 3.4e38f, sd = 0.f, g0[3] = {0.f, 0.f, 0.f};
        int npos = 0;
        knn_visit(g, q, c, R, r2, k, [&](int kk, float ex, float ey, float ez, float d2) {
            const float nx = nrm[kk * 3], ny = nrm[kk * 3 + 1], nz = nrm[kk * 3 + 2];
            const float d = nx * ex + ny * ey + nz * ez;
            npos += d > 0.f;
            if (d2 < dbest) {                         // the nearest neighbour sets the magnitude
                dbest = d2;
                const float len = sqrtf(d2);
                if (len < stdv * (ref_std ? ref_std[kk] : 1.f)) {
                    sd = fabsf(d);
                    const float sg = d > 0.f ? 1.f : -1.f;
                    g0[0] = sg * nx; g0[1] = sg * ny; g0[2] = sg * nz;
                } else {
                    sd = len;
                    g0[0] = ex / len; g0[1] = ey / len; g0[2] = ez / len;
                }
            }
        });
        const float sg = npos <= k / 2 ? -1.f : 1.f;
        out = sg * sd;
        gr[0] = sg * g0[0]; gr[1] = sg * g0[1]; gr[2] = sg * g0[2];
    }
    sdf[i] = out;
    if (grad) { grad[i * 3] = gr[0]; grad[i * 3 + 1] = gr[1]; grad[i * 3 + 2] = gr[2]; }
}

// mean distance of every (sorted) reference point to its k nearest reference points, itself included (the adaptive_knn radius,
// sdf_from_points.cu:176-184)
__global__ void __launch_bounds__(128) k_knn_mean_dist(KnnGrid g, int64_t n, int k, int max_ring, float* __restrict__ out,
                                                       int32_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float q[3] = {g.xyz[i * 3], g.xyz[i * 3 + 1], g.xyz[i * 3 + 2]};
    int c[3], R;
    float r2;
    cell_of(g, q, c);
    if (!knn_radius(g, q, c, k, max_ring, R, r2)) { valid[i] = 0; out[i] = 0.f; return; }
    float s = 0.f;
    knn_visit(g, q, c, R, r2, k, [&](int, float, float, float, float d2) { s += sqrtf(d2); });
    out[i] = s / (float)k;
    valid[i] = 1;
}

// ---- the same two operations with ONE scan per ring: the k nearest candidates kept in registers -------------------------------------
// The bisection above scans the candidate block ~35 times per query (ring search + 32 bit-steps + the visits).  For k <= 32 a
// thread keeps the k smallest (distance, index) pairs sorted in registers and inserts every candidate once (an unrolled
// compare-and-shift over KMAX slots; a candidate AT the k-th distance does not displace one seen earlier), ring after ring
// (only the new shell of cells is scanned) until the k-th distance lies inside the scanned block.  Measured against the
// REFERENCE'S OWN extension on the same MI355X (tests/sdfgen_vs_ref.py, profiles/r04_sdfgen_vs_reference.md): with the bisection
// 44 ms against the reference's 3.0 ms at 4 000 reference points / 1 000 queries and 2.7 s against 20 ms at 1 M / 1 M; with
// everything below 1.0 ms and 13.8 ms.
template <int KMAX>
struct TopK {
    // The k candidates live in the LAST k slots (ascending); the KMAX - k slots before them hold a sentinel below every distance and
    // are never displaced.  The k-th candidate is therefore always slot KMAX - 1, a fixed register: indexed by k it was a select
    // chain that the compiler turned into an indexed load from a scratch copy of the arrays, written back after every scanned cell.
    float d2[KMAX];
    int idx[KMAX];
    bool any;
    __device__ __forceinline__ void init(int k) {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) { d2[j] = j < KMAX - k ? -1.f : 3.4e38f; idx[j] = -1; }
        any = false;
    }
    __device__ __forceinline__ void insert(float d, int id) {
        if (!(d < d2[KMAX - 1])) return;
        any = true;
#pragma unroll
        for (int j = KMAX - 1; j >= 1; --j) {
            const bool shift = d < d2[j - 1];
            const bool here = !shift && d < d2[j];
            d2[j] = shift ? d2[j - 1] : (here ? d : d2[j]);
            idx[j] = shift ? idx[j - 1] : (here ? id : idx[j]);
        }
        if (d < d2[0]) { d2[0] = d; idx[0] = id; }
    }
    __device__ __forceinline__ float kth() const { return d2[KMAX - 1]; }               // 3.4e38 while fewer than k were seen
    __device__ __forceinline__ bool full() const { return d2[KMAX - 1] < 3.4e38f; }
};
// candidates [s, e) of the sorted cloud, four at a time: the twelve coordinate loads of a group are in flight together (one at a time,
// a cell of 30 points was 30 memory latencies end to end -- the insertion needs each distance before the next)
template <int KMAX>
__device__ __forceinline__ void knn_scan(const float* __restrict__ xyz, int s, int e, const float q[3], TopK<KMAX>& top) {
    int kk = s;
    for (; kk + 4 <= e; kk += 4) {
        float p[12];
#pragma unroll
        for (int t = 0; t < 12; ++t) p[t] = xyz[kk * 3 + t];
#pragma unroll
        for (int t = 0; t < 4; ++t) top.insert(knn_d2(p[3 * t] - q[0], p[3 * t + 1] - q[1], p[3 * t + 2] - q[2]), kk + t);
    }
    for (; kk < e; ++kk) top.insert(knn_d2(xyz[kk * 3] - q[0], xyz[kk * 3 + 1] - q[1], xyz[kk * 3 + 2] - q[2]), kk);
}
// The search itself, for ONE grid scale.  Cells are handed to `visit(cx, cy, cz)` in this order: the 27 cells around q's cell c
// NEAREST FIRST (by the distance of their box from q -- separable, one squared gap per axis and side; q's own cell has gap 0 and
// comes first), stopping at the first cell farther than the current k-th candidate: a query next to the surface looks at 3-5 cells
// instead of 27.  Then, only while the k-th candidate lies outside the scanned block, the shells R = 2 .. max_ring, every cell whose
// BOX is farther than the k-th candidate skipped without a look-up (the gap is shortened by 2 % of a cell against the fp32
// rounding of the box bounds c * cell at large coordinates).  Returns 1: the k nearest are in `top`; 0: fewer than k inside
// max_ring rings; -1: NOTHING within one cell (q is far from the cloud on this scale).
template <int KMAX, class Visit>
__device__ __forceinline__ int knn_rings(float cellsz, const float q[3], const int c[3], int k, int max_ring, bool coarser_exists,
                                         TopK<KMAX>& top, Visit&& visit) {
    top.init(k);
    const float slack = 0.02f * cellsz;
    float gs[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = (float)c[a] * cellsz;
        const float em = fmaxf(q[a] - lo - slack, 0.f), ep = fmaxf(lo + cellsz - q[a] - slack, 0.f);
        gs[a][0] = em * em; gs[a][1] = 0.f; gs[a][2] = ep * ep;
    }
    unsigned visited = 0u;
    int R = 1, dx = 0, dy = 0, dz = 0;
    for (;;) {
        int ox, oy, oz;
        if (R == 1) {
            float best = 3.4e38f;
            int bi = -1;
#pragma unroll
            for (int j = 0; j < 27; ++j) {
                const float gj = gs[0][j / 9] + gs[1][(j / 3) % 3] + gs[2][j % 3];
                if (!((visited >> j) & 1u) && gj < best) { best = gj; bi = j; }
            }
            if (bi < 0 || best > top.kth()) {
                if (top.kth() <= cellsz * cellsz) return 1;
                if (!top.any) return -1;
                // fewer than k candidates so far: nothing prunes the wider shells (98, 218, 386 look-ups) -- a coarser scale, if there
                // is one, reaches the same points in 27
                if (max_ring < 2 || (coarser_exists && !top.full())) return 0;
                R = 2; dx = dy = dz = -2;
                continue;
            }
            visited |= 1u << bi;
            ox = bi / 9 - 1; oy = (bi / 3) % 3 - 1; oz = bi % 3 - 1;
        } else {
            if (dx > R) {                                   // shell R done: the ball of radius R * cell around q lies inside the block
                const float rr = (float)R * cellsz;
                if (top.kth() <= rr * rr) return 1;
                if (++R > max_ring || (coarser_exists && !top.full())) return 0;
                dx = dy = dz = -R;
                continue;
            }
            ox = dx; oy = dy; oz = dz;
            const bool inner = abs(dx) < R && abs(dy) < R;  // (inside the shell only its two z faces are new)
            dz += inner ? 2 * R : 1;
            if (dz > R) { dz = -R; if (++dy > R) { dy = -R; ++dx; } }
            float gap2 = 0.f;
            const int cc[3] = {c[0] + ox, c[1] + oy, c[2] + oz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float lo = (float)cc[a] * cellsz, hi = lo + cellsz;
                const float e = fmaxf(fmaxf(lo - q[a], q[a] - hi) - slack, 0.f);
                gap2 = fmaf(e, e, gap2);
            }
            if (gap2 > top.kth()) continue;
        }
        visit(c[0] + ox, c[1] + oy, c[2] + oz);
    }
}
// one grid: the k nearest reference points of q; false: fewer than k inside max_ring rings
template <int KMAX>
__device__ __forceinline__ bool knn_topk(const KnnGrid& g, const float q[3], int k, int max_ring, TopK<KMAX>& top) {
    int c[3];
    cell_of(g, q, c);
    return knn_rings<KMAX>(g.cell, q, c, k, max_ring, false, top, [&](int cx, int cy, int cz) {
        const int ci = hash_find(g.hkeys, g.hvals, g.hcap, morton_biased(cx, cy, cz, NKSR_BIAS0));
        if (ci < 0) return;
        knn_scan<KMAX>(g.xyz, g.start[ci], g.end[ci], q, top);
    }) == 1;
}

// the two estimators over the k nearest neighbours in `top` (nearest first)
template <int KMAX>
__device__ __forceinline__ void sdf_estimate(const float* __restrict__ xyz, const float* __restrict__ nrm, const float* __restrict__ ref_std, const float q[3],
                                             const TopK<KMAX>& top, int k, float stdv, int imls, int64_t i, float* __restrict__ sdf,
                                             float* __restrict__ grad) {
    float out = 0.f, gr[3] = {0.f, 0.f, 0.f};
    if (imls) {
        const float inv_s2 = 1.f / (stdv * stdv);
        float emin = 0.f, wsum = 0.f, acc = 0.f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j < KMAX - k) continue;
            if (j == KMAX - k) emin = top.d2[j] * inv_s2;                       // the nearest neighbour
            const int kk = top.idx[j];
            const float ex = q[0] - xyz[kk * 3], ey = q[1] - xyz[kk * 3 + 1], ez = q[2] - xyz[kk * 3 + 2];
            const float nx = nrm[kk * 3], ny = nrm[kk * 3 + 1], nz = nrm[kk * 3 + 2];
            const float w = expf(-top.d2[j] * inv_s2 + emin);
            wsum += w;
            acc += (nx * ex + ny * ey + nz * ez) * w;
            gr[0] += nx * w; gr[1] += ny * w; gr[2] += nz * w;
        }
        out = acc / wsum;
        gr[0] /= wsum; gr[1] /= wsum; gr[2] /= wsum;
    } else {
        int npos = 0;
        float sd = 0.f, g0[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j < KMAX - k) continue;
            const int kk = top.idx[j];
            const float ex = q[0] - xyz[kk * 3], ey = q[1] - xyz[kk * 3 + 1], ez = q[2] - xyz[kk * 3 + 2];
            const float nx = nrm[kk * 3], ny = nrm[kk * 3 + 1], nz = nrm[kk * 3 + 2];
            const float d = nx * ex + ny * ey + nz * ez;
            npos += d > 0.f;
            if (j == KMAX - k) {                      // the nearest neighbour sets the magnitude
                const float len = sqrtf(top.d2[j]);
                if (len < stdv * (ref_std ? ref_std[kk] : 1.f)) {
                    sd = fabsf(d);
                    const float sg = d > 0.f ? 1.f : -1.f;
                    g0[0] = sg * nx; g0[1] = sg * ny; g0[2] = sg * nz;
                } else {
                    sd = len;
                    g0[0] = ex / len; g0[1] = ey / len; g0[2] = ez / len;
                }
            }
        }
        const float sg = npos <= k / 2 ? -1.f : 1.f;
        out = sg * sd;
        gr[0] = sg * g0[0]; gr[1] = sg * g0[1]; gr[2] = sg * g0[2];
    }
    sdf[i] = out;
    if (grad) { grad[i * 3] = gr[0]; grad[i * 3 + 1] = gr[1]; grad[i * 3 + 2] = gr[2]; }
}

template <int KMAX>
__global__ void __launch_bounds__(128) k_sdf_topk(KnnGrid g, const float* __restrict__ nrm, const float* __restrict__ ref_std,
                                                  const float* __restrict__ query, int64_t nq, int k, int max_ring, float stdv, int imls,
                                                  float* __restrict__ sdf, float* __restrict__ grad, int32_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float q[3] = {query[i * 3], query[i * 3 + 1], query[i * 3 + 2]};
    TopK<KMAX> top;
    if (!knn_topk<KMAX>(g, q, k, max_ring, top)) { valid[i] = 0; return; }
    valid[i] = 1;
    sdf_estimate<KMAX>(g.xyz, nrm, ref_std, q, top, k, stdv, imls, i, sdf, grad);
}

// ---- every scale in one launch: an octree over the same Morton-sorted points ---------------------------------------------------------
// Level l has cells of size cell * 2^l; its cell keys are the level-0 keys >> 3 l, so a level-l cell is a contiguous range of the
// sorted points AND of the level-(l-1) cells (its <= 8 children, in octant order: child[l][i] .. child[l][i + 1], cmask[l][i] = the
// occupied octants).  A query climbs to the first level with anything within one cell of it (27 look-ups per level -- a far query
// leaves the fine levels at once), and searches THAT scale with knn_rings; a cell there is not scanned but DESCENDED: children
// nearest octant first, every child whose box lies farther than the current k-th candidate cut off, points scanned only in
// cells of <= leaf points.  A query at distance d from the cloud therefore costs ~log(d / cell) box tests plus the few leaves that
// face it -- on the single coarse grid of the host's round loop (x4 cell per round, 16x points per cell) it scanned whole cells
// of thousands of points, 14 ms for the last 16 000 queries of 1 M.  The descent keeps one (node, cursor) pair per level in LDS --
// and so does the table of per-level pointers: indexed by a per-lane level IN THE KERNEL ARGUMENTS it was a vector load from the
// kernarg segment (host-visible, uncached) at every step, 1.7 ms for 4 000 queries that now take a fraction of it.
struct KnnPyramid {
    const int32_t* start[NKSR_KNN_LEVELS];
    const int32_t* end[NKSR_KNN_LEVELS];
    const int32_t* child[NKSR_KNN_LEVELS];
    const uint8_t* cmask[NKSR_KNN_LEVELS];
    const int64_t* hkeys[NKSR_KNN_LEVELS];
    const int32_t* hvals[NKSR_KNN_LEVELS];
    int hcap[NKSR_KNN_LEVELS];
    int levels, leaf;
    const float* xyz;
    float cell, inv_cell;
};
#define KNN_PYR_BLOCK 128
template <int KMAX>
__device__ __forceinline__ bool knn_topk_pyramid(const KnnPyramid& P, const float q[3], int k, int max_ring, TopK<KMAX>& top, int* node,
                                                 unsigned char* cursor) {
    int c0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float p;
        c0[a] = half_index(q[a], P.inv_cell, p) >> 1;
    }
    // the finest level with anything in the 27 cells around q: "occupied" is monotone in the level (the block of level l + 1 covers
    // the block of level l), so after level 0 it is bisected -- a far query pays ~4 probes, most of them ended by an early hit
    auto occupied = [&](int l) {
        const int bias = NKSR_BIAS0 >> l;
        for (int j = 0; j < 27; ++j)
            if (hash_find(P.hkeys[l], P.hvals[l], P.hcap[l],
                          morton_biased((c0[0] >> l) + j / 9 - 1, (c0[1] >> l) + (j / 3) % 3 - 1, (c0[2] >> l) + j % 3 - 1, bias)) >= 0)
                return true;
        return false;
    };
    int first = 0;
    if (P.levels > 1 && !occupied(0)) {
        int lo = 0;
        first = P.levels - 1;
        while (first - lo > 1) {
            const int mid = (lo + first) >> 1;
            if (occupied(mid)) first = mid; else lo = mid;
        }
    }
    for (int l = first; l < P.levels; ++l) {
        const float cell_l = P.cell * (float)(1 << l);
        const int c[3] = {c0[0] >> l, c0[1] >> l, c0[2] >> l};
        const int rc = knn_rings<KMAX>(cell_l, q, c, k, max_ring, l + 1 < P.levels, top, [&](int cx, int cy, int cz) {
            int ci = hash_find(P.hkeys[l], P.hvals[l], P.hcap[l], morton_biased(cx, cy, cz, NKSR_BIAS0 >> l));
            if (ci < 0) return;
            int lvl = l;
            bool enter = true;
            for (;;) {
                if (enter) {
                    const int s = P.start[lvl][ci], e = P.end[lvl][ci];
                    if (lvl == 0 || e - s <= P.leaf) {
                        knn_scan<KMAX>(P.xyz, s, e, q, top);
                        if (lvl == l) return;
                        ++lvl; cx >>= 1; cy >>= 1; cz >>= 1;               // back to the parent
                    } else {
                        node[lvl * KNN_PYR_BLOCK] = ci;
                        cursor[lvl * KNN_PYR_BLOCK] = 0;
                    }
                    enter = false;
                    continue;
                }
                // the next child of node[lvl] (cell (cx, cy, cz) of level lvl) worth a visit
                const int nd = node[lvl * KNN_PYR_BLOCK];
                int t = cursor[lvl * KNN_PYR_BLOCK];
                const unsigned mask = P.cmask[lvl][nd];
                const float cl = P.cell * (float)(1 << lvl), ch = 0.5f * cl, slack = 0.02f * ch;
                const unsigned qo = (q[0] >= ((float)cx + 0.5f) * cl ? 1u : 0u) | (q[1] >= ((float)cy + 0.5f) * cl ? 2u : 0u) |
                                    (q[2] >= ((float)cz + 0.5f) * cl ? 4u : 0u);
                bool found = false;
                while (t < 8) {
                    const unsigned o = ((0x76534210u >> (4 * t)) & 7u) ^ qo;      // octants by the number of axes they differ from q's in
                    ++t;
                    if (!((mask >> o) & 1u)) continue;
                    const int x = 2 * cx + (int)(o & 1u), y = 2 * cy + (int)((o >> 1) & 1u), z = 2 * cz + (int)(o >> 2);
                    const float lo[3] = {(float)x * ch, (float)y * ch, (float)z * ch};
                    float gap2 = 0.f;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float e = fmaxf(fmaxf(lo[a] - q[a], q[a] - (lo[a] + ch)) - slack, 0.f);
                        gap2 = fmaf(e, e, gap2);
                    }
                    if (gap2 > top.kth()) continue;
                    ci = P.child[lvl][nd] + __popc(mask & ((1u << o) - 1u));
                    cx = x; cy = y; cz = z;
                    found = true;
                    break;
                }
                if (found) {
                    cursor[lvl * KNN_PYR_BLOCK] = (unsigned char)t;
                    --lvl;
                    enter = true;
                } else {
                    if (lvl == l) return;
                    ++lvl; cx >>= 1; cy >>= 1; cz >>= 1;
                }
            }
        });
        if (rc == 1) return true;
    }
    return false;
}
template <int KMAX>
__global__ void __launch_bounds__(KNN_PYR_BLOCK) k_sdf_pyramid(KnnPyramid Parg, const float* __restrict__ nrm, const float* __restrict__ ref_std,
                                                               const float* __restrict__ query, int64_t nq, int k, int max_ring, float stdv,
                                                               int imls, float* __restrict__ sdf, float* __restrict__ grad,
                                                               int32_t* __restrict__ valid) {
    __shared__ int s_node[NKSR_KNN_LEVELS][KNN_PYR_BLOCK];
    __shared__ unsigned char s_cursor[NKSR_KNN_LEVELS][KNN_PYR_BLOCK];
    __shared__ KnnPyramid P;
    if (threadIdx.x == 0) P = Parg;
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float q[3] = {query[i * 3], query[i * 3 + 1], query[i * 3 + 2]};
    TopK<KMAX> top;
    if (!knn_topk_pyramid<KMAX>(P, q, k, max_ring, top, &s_node[0][threadIdx.x], &s_cursor[0][threadIdx.x])) { valid[i] = 0; return; }
    valid[i] = 1;
    sdf_estimate<KMAX>(P.xyz, nrm, ref_std, q, top, k, stdv, imls, i, sdf, grad);
}
template <int KMAX>
__global__ void __launch_bounds__(KNN_PYR_BLOCK) k_knn_mean_dist_pyramid(KnnPyramid Parg, int64_t n, int k, int max_ring, float* __restrict__ out,
                                                                         int32_t* __restrict__ valid) {
    __shared__ int s_node[NKSR_KNN_LEVELS][KNN_PYR_BLOCK];
    __shared__ unsigned char s_cursor[NKSR_KNN_LEVELS][KNN_PYR_BLOCK];
    __shared__ KnnPyramid P;
    if (threadIdx.x == 0) P = Parg;
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float q[3] = {P.xyz[i * 3], P.xyz[i * 3 + 1], P.xyz[i * 3 + 2]};
    TopK<KMAX> top;
    if (!knn_topk_pyramid<KMAX>(P, q, k, max_ring, top, &s_node[0][threadIdx.x], &s_cursor[0][threadIdx.x])) { valid[i] = 0; out[i] = 0.f; return; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j >= KMAX - k) s += sqrtf(top.d2[j]);
    out[i] = s / (float)k;
    valid[i] = 1;
}
// one level of the octree from the one below: thread p owns parent cell p (keys = sorted unique child keys >> 3)
__global__ void __launch_bounds__(256) k_pyramid_level(const int64_t* __restrict__ ckeys, int32_t nc, const int32_t* __restrict__ cstart,
                                                       const int32_t* __restrict__ cend, const int64_t* __restrict__ pkeys, int32_t np,
                                                       int32_t* __restrict__ child, uint8_t* __restrict__ cmask, int32_t* __restrict__ pstart,
                                                       int32_t* __restrict__ pend) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const int64_t key = pkeys[p];
    int lo = 0, hi = nc;                                    // first child: lower bound of key << 3
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((ckeys[mid] >> 3) < key) lo = mid + 1; else hi = mid;
    }
    unsigned m = 0u;
    int j = lo;
    for (; j < nc && (ckeys[j] >> 3) == key; ++j) m |= 1u << (unsigned)(ckeys[j] & 7);
    child[p] = lo;
    if (p == np - 1) child[np] = nc;
    cmask[p] = (uint8_t)m;
    pstart[p] = cstart[lo];
    pend[p] = cend[j - 1];
}

template <int KMAX>
__global__ void __launch_bounds__(128) k_knn_mean_dist_topk(KnnGrid g, int64_t n, int k, int max_ring, float* __restrict__ out,
                                                            int32_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float q[3] = {g.xyz[i * 3], g.xyz[i * 3 + 1], g.xyz[i * 3 + 2]};
    TopK<KMAX> top;
    if (!knn_topk<KMAX>(g, q, k, max_ring, top)) { valid[i] = 0; out[i] = 0.f; return; }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
        if (j >= KMAX - k) s += sqrtf(top.d2[j]);
    out[i] = s / (float)k;
    valid[i] = 1;
}

static KnnGrid make_grid(const float* xyz_sorted, const int32_t* start, const int32_t* end, const int64_t* hkeys,
                         const int32_t* hvals, int hcap, float cell, float inv_cell) {
    KnnGrid g;
    g.xyz = xyz_sorted; g.start = start; g.end = end; g.hkeys = hkeys; g.hvals = hvals; g.hcap = hcap;
    g.cell = cell;
    g.inv_cell = inv_cell;   // the SAME fp32 reciprocal the host binned the points with
    return g;
}

extern "C" int nksr_knn_pca_normals(const float* xyz_sorted, int64_t n, const int32_t* start, const int32_t* end,
                                    const int64_t* hkeys, const int32_t* hvals, int32_t hcap, float cell, float inv_cell, int k, int max_ring,
                                    float* normal_out, float* radius2_out, int32_t* valid_out, int32_t* todo_work, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (k < 3) return nksr_set_error(NKSR_ERR_ARG, "knn must be >= 3");
    KnnGrid g = make_grid(xyz_sorted, start, end, hkeys, hvals, hcap, cell, inv_cell);
    if (!todo_work || k > KW_CAP) {        // no work array: thread-per-query for all
        hipLaunchKernelGGL(k_knn_pca_normals, dim3(nksr_blocks(n, 128)), dim3(128), 0, (hipStream_t)stream, g, n, k, max_ring,
                           normal_out, radius2_out, valid_out, (const int32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_knn_pca_wave, dim3(nksr_blocks(n, KW_WAVES)), dim3(64 * KW_WAVES), 0, (hipStream_t)stream, g, n, k, max_ring,
                           normal_out, radius2_out, valid_out, todo_work);
        hipLaunchKernelGGL(k_knn_pca_normals, dim3(nksr_blocks(n, 128)), dim3(128), 0, (hipStream_t)stream, g, n, k, max_ring,
                           normal_out, radius2_out, valid_out, (const int32_t*)todo_work);
    }
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_nearest_index(const float* xyz_sorted, const int32_t* start, const int32_t* end, const int64_t* hkeys,
                                  const int32_t* hvals, int32_t hcap, float cell, float inv_cell, const float* query, int64_t nq, int max_ring,
                                  int32_t* index_out, void* stream) {
    if (nq <= 0) return NKSR_OK;
    KnnGrid g = make_grid(xyz_sorted, start, end, hkeys, hvals, hcap, cell, inv_cell);
    hipLaunchKernelGGL(k_nearest_index, dim3(nksr_blocks(nq, 128)), dim3(128), 0, (hipStream_t)stream, g, query, nq, max_ring,
                       index_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_sdf_from_points(const float* xyz_sorted, const float* normal_sorted, const float* ref_std_sorted, const int32_t* start,
                                    const int32_t* end, const int64_t* hkeys, const int32_t* hvals, int32_t hcap, float cell, float inv_cell,
                                    const float* query, int64_t nq, int k, int max_ring, float stdv, int imls, float* sdf_out,
                                    float* grad_out, int32_t* valid_out, void* stream) {
    if (nq <= 0) return NKSR_OK;
    if (k < 1) return nksr_set_error(NKSR_ERR_ARG, "nb_points must be >= 1");
    if (!(stdv > 0.f)) return nksr_set_error(NKSR_ERR_ARG, "stdv must be > 0");
    if (!xyz_sorted || !normal_sorted || !query || !sdf_out || !valid_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    KnnGrid g = make_grid(xyz_sorted, start, end, hkeys, hvals, hcap, cell, inv_cell);
    const dim3 gr(nksr_blocks(nq, 128)), bl(128);
#define SDF_TOPK(KM) hipLaunchKernelGGL((k_sdf_topk<KM>), gr, bl, 0, (hipStream_t)stream, g, normal_sorted, ref_std_sorted, query, nq, k, max_ring, stdv, imls, sdf_out, grad_out, valid_out)
    if (k <= 8) SDF_TOPK(8);
    else if (k <= 16) SDF_TOPK(16);
    else if (k <= 32) SDF_TOPK(32);
    else                                   // more neighbours than fit the register list: the bisection kernel
        hipLaunchKernelGGL(k_sdf_from_points, gr, bl, 0, (hipStream_t)stream, g, normal_sorted, ref_std_sorted, query,
                           nq, k, max_ring, stdv, imls, sdf_out, grad_out, valid_out);
#undef SDF_TOPK
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

static int make_pyramid(KnnPyramid& P, const nksr_knn_pyramid_t* p) {
    if (!p || p->levels < 1 || p->levels > NKSR_KNN_LEVELS || !p->xyz_sorted) return nksr_set_error(NKSR_ERR_ARG, "kNN pyramid: 1..%d levels", NKSR_KNN_LEVELS);
    memset(&P, 0, sizeof(P));
    for (int l = 0; l < p->levels; ++l) {
        if (!p->start[l] || !p->end[l] || !p->hkeys[l] || !p->hvals[l] || p->hcap[l] < 8 || (l > 0 && (!p->child[l] || !p->cmask[l])))
            return nksr_set_error(NKSR_ERR_ARG, "kNN pyramid: level %d incomplete", l);
        P.start[l] = p->start[l]; P.end[l] = p->end[l]; P.child[l] = p->child[l]; P.cmask[l] = p->cmask[l];
        P.hkeys[l] = p->hkeys[l]; P.hvals[l] = p->hvals[l]; P.hcap[l] = p->hcap[l];
    }
    P.levels = p->levels; P.leaf = p->leaf > 0 ? p->leaf : 48; P.xyz = p->xyz_sorted; P.cell = p->cell; P.inv_cell = p->inv_cell;
    return NKSR_OK;
}
extern "C" int nksr_knn_pyramid_level(const int64_t* child_keys, int32_t n_child, const int32_t* child_start, const int32_t* child_end,
                                      const int64_t* keys, int32_t n, int32_t* child_out, uint8_t* cmask_out, int32_t* start_out,
                                      int32_t* end_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!child_keys || !child_start || !child_end || !keys || !child_out || !cmask_out || !start_out || !end_out || n_child < n)
        return nksr_set_error(NKSR_ERR_ARG, "kNN pyramid level: NULL arrays or more parents than children");
    hipLaunchKernelGGL(k_pyramid_level, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, child_keys, n_child, child_start, child_end,
                       keys, n, child_out, cmask_out, start_out, end_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_sdf_from_points_pyramid(const nksr_knn_pyramid_t* pyramid, const float* normal_sorted, const float* ref_std_sorted, const float* query,
                                            int64_t nq, int k, int max_ring, float stdv, int imls, float* sdf_out, float* grad_out,
                                            int32_t* valid_out, void* stream) {
    if (nq <= 0) return NKSR_OK;
    if (k < 1 || k > 32) return nksr_set_error(NKSR_ERR_ARG, "pyramid search: 1 <= nb_points <= 32 (got %d)", k);
    if (!(stdv > 0.f)) return nksr_set_error(NKSR_ERR_ARG, "stdv must be > 0");
    if (!normal_sorted || !query || !sdf_out || !valid_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    KnnPyramid P;
    if (int rc = make_pyramid(P, pyramid)) return rc;
    const dim3 gr(nksr_blocks(nq, 128)), bl(128);
#define SDF_PYR(KM) hipLaunchKernelGGL((k_sdf_pyramid<KM>), gr, bl, 0, (hipStream_t)stream, P, normal_sorted, ref_std_sorted, query, nq, k, max_ring, stdv, imls, sdf_out, grad_out, valid_out)
    if (k <= 8) SDF_PYR(8);
    else if (k <= 16) SDF_PYR(16);
    else SDF_PYR(32);
#undef SDF_PYR
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_knn_mean_dist_pyramid(const nksr_knn_pyramid_t* pyramid, int64_t n, int k, int max_ring, float* out, int32_t* valid_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (k < 1 || k > 32) return nksr_set_error(NKSR_ERR_ARG, "pyramid search: 1 <= k <= 32 (got %d)", k);
    KnnPyramid P;
    if (int rc = make_pyramid(P, pyramid)) return rc;
    const dim3 gr(nksr_blocks(n, 128)), bl(128);
    if (k <= 8) hipLaunchKernelGGL((k_knn_mean_dist_pyramid<8>), gr, bl, 0, (hipStream_t)stream, P, n, k, max_ring, out, valid_out);
    else if (k <= 16) hipLaunchKernelGGL((k_knn_mean_dist_pyramid<16>), gr, bl, 0, (hipStream_t)stream, P, n, k, max_ring, out, valid_out);
    else hipLaunchKernelGGL((k_knn_mean_dist_pyramid<32>), gr, bl, 0, (hipStream_t)stream, P, n, k, max_ring, out, valid_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_knn_mean_dist(const float* xyz_sorted, int64_t n, const int32_t* start, const int32_t* end, const int64_t* hkeys,
                                  const int32_t* hvals, int32_t hcap, float cell, float inv_cell, int k, int max_ring, float* out,
                                  int32_t* valid_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (k < 1) return nksr_set_error(NKSR_ERR_ARG, "k must be >= 1");
    KnnGrid g = make_grid(xyz_sorted, start, end, hkeys, hvals, hcap, cell, inv_cell);
    const dim3 gr(nksr_blocks(n, 128)), bl(128);
    if (k <= 8) hipLaunchKernelGGL((k_knn_mean_dist_topk<8>), gr, bl, 0, (hipStream_t)stream, g, n, k, max_ring, out, valid_out);
    else if (k <= 16) hipLaunchKernelGGL((k_knn_mean_dist_topk<16>), gr, bl, 0, (hipStream_t)stream, g, n, k, max_ring, out, valid_out);
    else if (k <= 32) hipLaunchKernelGGL((k_knn_mean_dist_topk<32>), gr, bl, 0, (hipStream_t)stream, g, n, k, max_ring, out, valid_out);
    else hipLaunchKernelGGL(k_knn_mean_dist, gr, bl, 0, (hipStream_t)stream, g, n, k, max_ring, out, valid_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
