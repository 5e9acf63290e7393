// Chunk plumbing of reconstruct(chunk_size=...) (reference call site examples/recons_by_chunk.py:29; nksr_amd/chunking.py):
//   * which chunks does a point belong to (core +- band along the split axes)            -> the batched solve's input
//   * which chunks weigh at a query, with what partition-of-unity weight                  -> the blend  f = sum w_c f_c / sum w_c
//   * the blend itself, chunks in ascending order per query (fixed summation order)
// Plain HBM-bound integer / float work: one thread per point, a count pass and a fill pass around an exclusive scan.  (Round 3: these
// were ~150 torch elementwise / index / nonzero launches per call over 10^7 elements, ~45 ms of the 0.54 s scene step.)
// The arithmetic is the one the oracle states (oracle/chunking.py: weight(), the selection masks): every product is rounded before it
// is used -- contraction off.
#include "common.h"
#pragma clang fp contract(off)

__device__ __forceinline__ int chunk_home(const nksr_chunk_grid_t& G, int a, float x) {
    int i = (int)floorf((x - G.origin[a]) * G.inv_cs);
    return i < 0 ? 0 : (i >= G.grid[a] ? G.grid[a] - 1 : i);
}
__device__ __forceinline__ float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

// MODE 0: membership for the solve (x >= lo_j - band and x < hi_j + band along every split axis, chunk wanted)
// MODE 1: blend weight > 0 and chunk present.  FILL: write the records at offsets[i] + k, else count them.
template <int MODE, bool FILL>
__global__ void __launch_bounds__(256) k_chunk_pairs(nksr_chunk_grid_t G, const float* __restrict__ xyz, int64_t n,
                                                     const int32_t* __restrict__ flag, const int32_t* __restrict__ offsets,
                                                     int32_t* __restrict__ counts, int64_t* __restrict__ pair_point, int32_t* __restrict__ pair_chunk,
                                                     float* __restrict__ pair_w, float* __restrict__ pair_xyz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    int home[3], lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (G.grid[a] > 1) { home[a] = chunk_home(G, a, x[a]); lo[a] = -G.reach; hi[a] = G.reach; }
        else { home[a] = 0; lo[a] = hi[a] = 0; }
    }
    int64_t w = FILL ? offsets[i] : 0;
    int cnt = 0;
    for (int ox = lo[0]; ox <= hi[0]; ++ox)
        for (int oy = lo[1]; oy <= hi[1]; ++oy)
            for (int oz = lo[2]; oz <= hi[2]; ++oz) {          // ascending chunk id: (cx * gy + cy) * gz + cz
                const int o[3] = {ox, oy, oz};
                int j[3];
                bool ok = true;
                float wt = 1.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    j[a] = home[a] + o[a];
                    if (G.grid[a] <= 1) continue;
                    if (j[a] < 0 || j[a] >= G.grid[a]) { ok = false; continue; }
                    if (MODE == 0) ok = ok && x[a] >= G.lo_sel[a][j[a]] && x[a] < G.hi_sel[a][j[a]];
                    else {
                        const float up = clamp01((x[a] - G.lo_w[a][j[a]]) * G.inv_2ov);
                        const float dn = clamp01((G.hi_w[a][j[a]] - x[a]) * G.inv_2ov);
                        wt = (wt * up) * dn;                    // the order of the oracle's weight(): ((w up_x) dn_x) up_y ...
                    }
                }
                if (!ok) continue;
                const int c = (j[0] * G.grid[1] + j[1]) * G.grid[2] + j[2];
                if (flag[c] < 0 || (MODE == 1 && !(wt > 0.f))) continue;
                if (FILL) {
                    pair_point[w] = i;
                    pair_chunk[w] = c;
                    if (MODE == 1) {
                        pair_w[w] = wt;
                        pair_xyz[w * 3] = x[0] + G.shift[c * 3];        // the chunk's slot of the exploded frame: one fp32 rounding
                        pair_xyz[w * 3 + 1] = x[1] + G.shift[c * 3 + 1];
                        pair_xyz[w * 3 + 2] = x[2] + G.shift[c * 3 + 2];
                    }
                    ++w;
                } else {
                    ++cnt;
                }
            }
    if (!FILL) counts[i] = cnt;
}

// f(x) = sum_k w_k f_k / max(sum_k w_k, 1e-20), pairs of a query in ascending chunk order (the fill order)
__global__ void __launch_bounds__(256) k_chunk_blend(int64_t n, const int32_t* __restrict__ offsets, const float* __restrict__ pair_w,
                                                     const float* __restrict__ pair_f, const float* __restrict__ pair_g,
                                                     float* __restrict__ f_out, float* __restrict__ g_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float num = 0.f, den = 0.f, g[3] = {0.f, 0.f, 0.f};
    for (int64_t k = offsets[i]; k < offsets[i + 1]; ++k) {
        const float w = pair_w[k];
        num = num + pair_f[k] * w;
        den = den + w;
        if (pair_g) { g[0] = g[0] + pair_g[k * 3] * w; g[1] = g[1] + pair_g[k * 3 + 1] * w; g[2] = g[2] + pair_g[k * 3 + 2] * w; }
    }
    den = den < 1e-20f ? 1e-20f : den;
    f_out[i] = num / den;
    if (g_out) { g_out[i * 3] = g[0] / den; g_out[i * 3 + 1] = g[1] / den; g_out[i * 3 + 2] = g[2] / den; }
}

static int chunk_check(const nksr_chunk_grid_t* G, int mode) {
    if (!G) return nksr_set_error(NKSR_ERR_ARG, "chunk grid is NULL");
    for (int a = 0; a < 3; ++a) {
        if (G->grid[a] < 1) return nksr_set_error(NKSR_ERR_ARG, "chunk grid must be >= 1 per axis");
        if (G->grid[a] > 1 && (mode == 0 ? (!G->lo_sel[a] || !G->hi_sel[a]) : (!G->lo_w[a] || !G->hi_w[a])))
            return nksr_set_error(NKSR_ERR_ARG, "chunk grid: bounds of a split axis are NULL");
    }
    if (G->reach < 1 || G->reach > 4) return nksr_set_error(NKSR_ERR_ARG, "chunk grid: reach (candidate window of a point: home chunk +- reach, floor(band / chunk_size) + 1) must be 1..4, got %d", G->reach);
    if (mode == 1 && !G->shift) return nksr_set_error(NKSR_ERR_ARG, "chunk grid: shift is NULL");
    return NKSR_OK;
}

extern "C" int nksr_chunk_pair_counts(const nksr_chunk_grid_t* G, int mode, const float* xyz, int64_t n, const int32_t* chunk_flag,
                                      int32_t* counts_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (int rc = chunk_check(G, mode)) return rc;
    if (!xyz || !chunk_flag || !counts_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    const dim3 grid(nksr_blocks(n, 256)), blk(256);
    if (mode == 0) hipLaunchKernelGGL((k_chunk_pairs<0, false>), grid, blk, 0, (hipStream_t)stream, *G, xyz, n, chunk_flag, (const int32_t*)nullptr, counts_out,
                                      (int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL((k_chunk_pairs<1, false>), grid, blk, 0, (hipStream_t)stream, *G, xyz, n, chunk_flag, (const int32_t*)nullptr, counts_out,
                            (int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_chunk_pair_fill(const nksr_chunk_grid_t* G, int mode, const float* xyz, int64_t n, const int32_t* chunk_flag,
                                    const int32_t* offsets, int64_t* pair_point_out, int32_t* pair_chunk_out, float* pair_w_out,
                                    float* pair_xyz_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (int rc = chunk_check(G, mode)) return rc;
    if (!xyz || !chunk_flag || !offsets || !pair_point_out || !pair_chunk_out || (mode == 1 && (!pair_w_out || !pair_xyz_out)))
        return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    const dim3 grid(nksr_blocks(n, 256)), blk(256);
    if (mode == 0) hipLaunchKernelGGL((k_chunk_pairs<0, true>), grid, blk, 0, (hipStream_t)stream, *G, xyz, n, chunk_flag, offsets, (int32_t*)nullptr,
                                      pair_point_out, pair_chunk_out, pair_w_out, pair_xyz_out);
    else hipLaunchKernelGGL((k_chunk_pairs<1, true>), grid, blk, 0, (hipStream_t)stream, *G, xyz, n, chunk_flag, offsets, (int32_t*)nullptr,
                            pair_point_out, pair_chunk_out, pair_w_out, pair_xyz_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_chunk_blend(int64_t n, const int32_t* offsets, const float* pair_w, const float* pair_f, const float* pair_grad,
                                float* f_out, float* grad_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!offsets || !pair_w || !pair_f || !f_out || ((pair_grad == nullptr) != (grad_out == nullptr))) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    hipLaunchKernelGGL(k_chunk_blend, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, n, offsets, pair_w, pair_f, pair_grad, f_out, grad_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- halo selection of a batch of chunks (chunking.ChunkPart.pack_halos) ----------------------------------------------------------------
// One thread per voxel of a level: its chunk = the last key range that starts at or before its key, its centre in the chunk's own frame,
// inside one of the chunk's band intervals (two per axis, thresholds already widened by the level's support) or not.  The same
// comparisons in the same fp32 arithmetic as chunking.pack_field's per-chunk masks (contraction is off in this file); before, ~65
// torch launches per level.
__global__ void __launch_bounds__(256) k_halo_band_flags(const int64_t* __restrict__ keys, const int32_t* __restrict__ ijk, int64_t n,
                                                         const int64_t* __restrict__ klo, int nchunk, const float* __restrict__ shift,
                                                         const float* __restrict__ tlo, const float* __restrict__ thi, float w,
                                                         int32_t* __restrict__ seg_out, int32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t k = keys[i];
    int lo = 0, hi = nchunk;                                  // upper bound: the ranges that start at or before k
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (klo[mid] <= k) lo = mid + 1; else hi = mid;
    }
    const int seg = lo > 0 ? lo - 1 : 0;
    bool in = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ca = ((float)ijk[i * 3 + a] + 0.5f) * w - shift[seg * 3 + a];
#pragma unroll
        for (int q = 0; q < 2; ++q) in = in || (ca >= tlo[(seg * 3 + a) * 2 + q] && ca <= thi[(seg * 3 + a) * 2 + q]);
    }
    seg_out[i] = seg;
    flags[i] = in ? 1 : 0;
}

extern "C" int nksr_halo_band_flags(const int64_t* keys, const int32_t* ijk, int64_t n, const int64_t* klo, int32_t nchunk, const float* shift,
                                    const float* tlo, const float* thi, float w, int32_t* seg_out, int32_t* flags_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (nchunk < 1 || !keys || !ijk || !klo || !shift || !tlo || !thi || !seg_out || !flags_out) return nksr_set_error(NKSR_ERR_ARG, "halo_band_flags: NULL arrays / no chunks");
    hipLaunchKernelGGL(k_halo_band_flags, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, keys, ijk, n, klo, (int)nchunk, shift, tlo, thi, w,
                       seg_out, flags_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- which mesh vertices of a rank's piece can ANOTHER rank emit too (the seam merge on rank 0 groups only those) --------------------------
// A lattice-mesh vertex lies on the lattice edge (vertex g, axis a); the four lattice cells around that edge are the only ones whose
// triangles use it, and a cell is meshed by the rank that owns the chunk holding the centre of its base voxel (chunking.base_cell_mask:
// centre = (floor(cell / R) + 0.5) w0, chunk = floor((centre - origin) * (1 / chunk_size)) clamped -- the same fp32 operations here,
// each rounded: contraction is off in this file).  flag = 1 when one of the four cells is not this rank's.
__device__ __forceinline__ int floor_div_i(int x, int r) { return x >= 0 ? x / r : -((-x + r - 1) / r); }
__global__ void __launch_bounds__(256) k_edge_seam_flags(nksr_chunk_grid_t G, const int64_t* __restrict__ vkey, const int8_t* __restrict__ axis,
                                                         int64_t n, int R, float w0, const int32_t* __restrict__ owner, int rank,
                                                         uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int g[3];
    morton_decode_biased(vkey[i], NKSR_BIAS0, g[0], g[1], g[2]);
    const int a = axis[i], b = (a + 1) % 3, c = (a + 2) % 3;
    bool seam = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int cell[3] = {g[0], g[1], g[2]};
        cell[b] -= q & 1;
        cell[c] -= q >> 1;
        int lin = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float centre = ((float)floor_div_i(cell[k], R) + 0.5f) * w0;
            lin = lin * G.grid[k] + chunk_home(G, k, centre);
        }
        seam = seam || owner[lin] != rank;
    }
    flags[i] = seam ? 1 : 0;
}

extern "C" int nksr_edge_seam_flags(const nksr_chunk_grid_t* grid, const int64_t* vkey, const int8_t* axis, int64_t n, int32_t cells_per_voxel,
                                    float w0, const int32_t* owner, int32_t rank, uint8_t* flags_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!grid || !vkey || !axis || !owner || !flags_out || cells_per_voxel < 1) return nksr_set_error(NKSR_ERR_ARG, "edge_seam_flags: NULL arrays / cells_per_voxel < 1");
    hipLaunchKernelGGL(k_edge_seam_flags, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, *grid, vkey, axis, n, (int)cells_per_voxel, w0, owner,
                       (int)rank, flags_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- which points lie in (or within `reach` of) a core this rank owns (chunking.MultiChunkField.owns_points / near_owned) --------------------
// flag = 1 when owner[chunk of (x + o)] == rank for some offset o in {-reach, 0, +reach}^(split axes); reach = 0: the chunk of x itself.
// The arithmetic of chunk_of: (x + o) rounded, then floor(((x + o) - origin) * (1 / chunk_size)) clamped.  Before: ~100 torch launches per call.
__global__ void __launch_bounds__(256) k_points_owner_flags(nksr_chunk_grid_t G, const float* __restrict__ xyz, int64_t n, float reach,
                                                            const int32_t* __restrict__ owner, int rank, uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int idx[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = xyz[i * 3 + a];
        idx[a][1] = chunk_home(G, a, x);
        const bool split = G.grid[a] > 1 && reach > 0.f;
        idx[a][0] = split ? chunk_home(G, a, x + (-reach)) : idx[a][1];
        idx[a][2] = split ? chunk_home(G, a, x + reach) : idx[a][1];
    }
    bool mine = false;
    for (int q = 0; q < 27; ++q) {
        const int kx = q / 9, ky = (q / 3) % 3, kz = q % 3;
        mine = mine || owner[(idx[0][kx] * G.grid[1] + idx[1][ky]) * G.grid[2] + idx[2][kz]] == rank;
    }
    flags[i] = mine ? 1 : 0;
}

extern "C" int nksr_points_owner_flags(const nksr_chunk_grid_t* grid, const float* xyz, int64_t n, float reach, const int32_t* owner, int32_t rank,
                                       uint8_t* flags_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!grid || !xyz || !owner || !flags_out || !(reach >= 0.f)) return nksr_set_error(NKSR_ERR_ARG, "points_owner_flags: NULL arrays / reach < 0");
    hipLaunchKernelGGL(k_points_owner_flags, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, *grid, xyz, n, reach, owner, (int)rank, flags_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
