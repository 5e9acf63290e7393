// Grid-hash k-nearest-neighbour kernels:
//   k_knn_pca_normals  kNN-PCA normal estimation -- nksr.get_estimate_normal_preprocess_fn(knn, deg)
//                      (reference call sites examples/recons_waymo.py:36, gis_app.py:41; CPU recipe
//                      examples/recons_waymo_cpu.py:21-41)
//   k_nearest_index    nearest input point of every query -- fields.PCNNField colour lookup
//                      (examples/recons_colored_mesh.py:28)
// The cloud is Morton-sorted by a uniform grid (cell size chosen by the host so that a 3^3 block
// holds a few times k points); cells are contiguous point ranges found through the voxel hash.
// EXACT selection without a per-thread heap: the k-th smallest squared distance is found by a
// 32-step bisection on the float bit pattern (positive floats order like their bit patterns), each
// step re-counting the candidates (L1/L2 resident).  fp32 distances decide everything except the few
// candidates within a couple of ulps of that radius: those are ranked by their fp64 distance, so the
// neighbour SET is the one an fp64 kd-tree search returns (an fp32-only rule swaps near-equidistant
// neighbours against it: normals off by ~1e-3 at a few points, measured in round 2).  The covariance
// is then accumulated in fp64 over that set.  One thread per query, queries in Morton order
// (neighbouring lanes scan the same cells).
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
                    n += (ex * ex + ey * ey + ez * ez) <= r2;
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
                                                         int32_t* __restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
                                    float* normal_out, float* radius2_out, int32_t* valid_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (k < 3) return nksr_set_error(NKSR_ERR_ARG, "knn must be >= 3");
    KnnGrid g = make_grid(xyz_sorted, start, end, hkeys, hvals, hcap, cell, inv_cell);
    hipLaunchKernelGGL(k_knn_pca_normals, dim3(nksr_blocks(n, 128)), dim3(128), 0, (hipStream_t)stream, g, n, k, max_ring,
                       normal_out, radius2_out, valid_out);
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
