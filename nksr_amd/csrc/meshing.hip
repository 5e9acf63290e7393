// Dual marching cubes on the finest level of the hierarchy + MISE refinement
// (field.extract_dual_mesh; reference call sites examples/recons_simple.py:27,
// recons_scannet.py:29, models/nksr_net.py:214,284).  All integer/topology work; the field
// values come from nksr_evaluate_f.  Lattice convention:
//   x = g*h + w0/2,  h = w0/(U 2^m),  lattice keys = Morton(g + 2^20).
// Compaction is ordered (deterministic): per-wave ballot + popcount prefix inside a block,
// block offsets from an exclusive scan.
#include "common.h"
// The oracle rounds every fp32 product before it is used (numpy).  Device code contracts a * b + c into one fma by default --
// x * inv_w - centre then keeps the unrounded product, the trilinear weights move by an ulp of p and a splat whose normals nearly
// cancel amplifies that to 1e-4 in the unit target (measured in round 3) -- so contraction is off in this file; explicit fmaf stays.
#pragma clang fp contract(off)
#include "mc_table.h"

#define CMP_BLOCK 256

// ---- ordered stream compaction ----------------------------------------------------------------
__global__ void __launch_bounds__(CMP_BLOCK) k_flag_block_count(const int32_t* __restrict__ flags, int64_t n,
                                                                 int32_t* __restrict__ block_counts) {
    __shared__ int wsum[CMP_BLOCK / 64];
    int64_t i = (int64_t)blockIdx.x * CMP_BLOCK + threadIdx.x;
    bool f = i < n && flags[i] != 0;
    unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < CMP_BLOCK / 64; ++w) t += wsum[w];
        block_counts[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(CMP_BLOCK) k_flag_block_scatter(const int32_t* __restrict__ flags, int64_t n,
                                                                   const int32_t* __restrict__ block_offsets,
                                                                   int32_t* __restrict__ sel) {
    __shared__ int wsum[CMP_BLOCK / 64];
    int64_t i = (int64_t)blockIdx.x * CMP_BLOCK + threadIdx.x;
    bool f = i < n && flags[i] != 0;
    unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int base = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (f) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
}

extern "C" int nksr_compact_block_counts(const int32_t* flags, int64_t n, int32_t* block_counts, void* stream) {
    if (n <= 0) return NKSR_OK;
    hipLaunchKernelGGL(k_flag_block_count, dim3(nksr_blocks(n, CMP_BLOCK)), dim3(CMP_BLOCK), 0, (hipStream_t)stream, flags, n, block_counts);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_compact_scatter(const int32_t* flags, int64_t n, const int32_t* block_offsets, int32_t* sel, void* stream) {
    if (n <= 0) return NKSR_OK;
    hipLaunchKernelGGL(k_flag_block_scatter, dim3(nksr_blocks(n, CMP_BLOCK)), dim3(CMP_BLOCK), 0, (hipStream_t)stream, flags, n, block_offsets, sel);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- dual cells ---------------------------------------------------------------------------------
__global__ void k_base_cell_flags(const int32_t* __restrict__ nbr, int n, int32_t* __restrict__ flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t* nb = nbr + (int64_t)i * 27;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) ok = ok && nb[(1 + (c >> 2)) * 9 + (1 + ((c >> 1) & 1)) * 3 + (1 + (c & 1))] >= 0;
    flags[i] = ok ? 1 : 0;
}

__global__ void k_base_cell_keys(const int32_t* __restrict__ ijk, const int32_t* __restrict__ sel, int64_t nsel, int U,
                                 int64_t* __restrict__ keys) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int U3 = U * U * U;
    if (t >= nsel * U3) return;
    int64_t c = t / U3;
    int r = (int)(t % U3);
    int i = sel[c];
    int x = ijk[i * 3] * U + r / (U * U), y = ijk[i * 3 + 1] * U + (r / U) % U, z = ijk[i * 3 + 2] * U + r % U;
    keys[t] = morton_biased(x, y, z, NKSR_BIAS0);
}

// lattice cells covered by the dual cell of a level-`level` voxel (level >= 1): the dual cell spans the centres
// (i + 1/2) w_d .. (i + 3/2) w_d, i.e. fine-lattice coordinates 2^d i + 2^(d-1) - 1/2 .. + 2^d; the S = U 2^d lattice cells per axis
// whose centre lies inside start at (2^d i + 2^(d-1) - 1) U.  level 0 reproduces k_base_cell_keys.
__global__ void k_level_cell_keys(const int32_t* __restrict__ ijk, const int32_t* __restrict__ sel, int64_t nsel, int level, int U,
                                  int64_t* __restrict__ keys) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = U << level;
    const int64_t S3 = (int64_t)S * S * S;
    if (t >= nsel * S3) return;
    const int64_t c = t / S3;
    const int r = (int)(t % S3);
    const int i = sel[c];
    const int off = level == 0 ? 0 : (1 << (level - 1)) - 1;
    const int x = ((ijk[i * 3] << level) + off) * U + r / (S * S), y = ((ijk[i * 3 + 1] << level) + off) * U + (r / S) % S,
              z = ((ijk[i * 3 + 2] << level) + off) * U + r % S;
    keys[t] = morton_biased(x, y, z, NKSR_BIAS0);
}

__global__ void k_cell_corner_keys(const int64_t* __restrict__ cell_keys, int64_t ncell, int64_t* __restrict__ ck) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncell * 8) return;
    int x, y, z;
    morton_decode_biased(cell_keys[t >> 3], NKSR_BIAS0, x, y, z);
    int c = (int)(t & 7);
    ck[t] = morton_biased(x + (c >> 2), y + ((c >> 1) & 1), z + (c & 1), NKSR_BIAS0);
}

__global__ void k_cell_children(const int64_t* __restrict__ cell_keys, const int32_t* __restrict__ sel, int64_t nsel,
                                int64_t* __restrict__ child) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsel * 8) return;
    int x, y, z;
    morton_decode_biased(cell_keys[sel[t >> 3]], NKSR_BIAS0, x, y, z);
    int c = (int)(t & 7);
    child[t] = morton_biased(2 * x + (c >> 2), 2 * y + ((c >> 1) & 1), 2 * z + (c & 1), NKSR_BIAS0);
}

__global__ void k_lattice_positions(const int64_t* __restrict__ vkeys, int64_t n, float h, float half_w0,
                                    float* __restrict__ xyz) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    morton_decode_biased(vkeys[i], NKSR_BIAS0, x, y, z);
    // g * h + w0 / 2 with TWO fp32 roundings (contraction is off in this file).  Plain operators on purpose:
    // __fadd_rn(__fmul_rn(..)) are inline header functions compiled with contraction allowed, the pair still fused into one fma and
    // every lattice position sat one ulp off the oracle's -- 3e-7 of field difference at the mesh vertices (found in round 3)
    const float px = (float)x * h, py = (float)y * h, pz = (float)z * h;
    xyz[i * 3] = px + half_w0;
    xyz[i * 3 + 1] = py + half_w0;
    xyz[i * 3 + 2] = pz + half_w0;
}

__global__ void k_cell_config(const int32_t* __restrict__ corner_idx, const float* __restrict__ f, int64_t ncell,
                              int32_t* __restrict__ config, int32_t* __restrict__ ntri) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncell) return;
    int cfg = 0, j[8], lowest = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) { j[c] = corner_idx[i * 8 + c]; lowest = j[c] < lowest ? j[c] : lowest; }
#pragma unroll
    for (int c = 0; c < 8; ++c) cfg |= (f[j[c] < 0 ? 0 : j[c]] > 0.f) ? (1 << c) : 0;
    if (lowest < 0) cfg = 0;                              // (adaptive dual graph: a corner that lacks one of its eight cells emits nothing)
    config[i] = cfg;
    ntri[i] = MC_NTRI[cfg];
}

__global__ void k_cell_active_flags(const int32_t* __restrict__ config, int64_t ncell, int32_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncell) return;
    int c = config[i];
    flags[i] = (c != 0 && c != 255) ? 1 : 0;
}

__global__ void k_mc_emit(const int32_t* __restrict__ corner_idx, const int32_t* __restrict__ config,
                          const int32_t* __restrict__ tri_offset, int64_t ncell, int64_t* __restrict__ edge_keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncell) return;
    const int cfg = config[i];
    const int nt = MC_NTRI[cfg];
    int64_t o = (int64_t)tri_offset[i] * 3;
    for (int t = 0; t < nt * 3; ++t) {
        int e = MC_TRI[cfg][t];
        edge_keys[o + t] = (int64_t)corner_idx[i * 8 + MC_EDGE_LO[e]] * 3 + MC_EDGE_AXIS[e];
    }
}

// (lattice vertices are looked up through an open-addressing hash of their keys: one or two probes instead of the ~21 dependent
// loads of a binary search over millions of sorted keys -- every such load costs a full gather instruction per wavefront)
__global__ void k_mc_vertices(const int64_t* __restrict__ edge_keys, int64_t nedge, const int64_t* __restrict__ vkeys,
                              const int64_t* __restrict__ hkeys, const int32_t* __restrict__ hvals, int hcap,
                              const float* __restrict__ vpos, const float* __restrict__ f, float h,
                              float* __restrict__ verts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nedge) return;
    int64_t ek = edge_keys[i];
    int64_t v0 = ek / 3;
    int axis = (int)(ek % 3);
    int g[3];
    morton_decode_biased(vkeys[v0], NKSR_BIAS0, g[0], g[1], g[2]);
    g[axis] += 1;
    const int64_t k1 = morton_biased(g[0], g[1], g[2], NKSR_BIAS0);
    const int v1 = hash_find(hkeys, hvals, hcap, k1);          // the far end of an emitted edge is a lattice vertex
    float f0 = f[v0], f1 = f[v1 >= 0 ? v1 : v0];
    const float df = f0 - f1;
    const float t = f0 / df;
    float p[3] = {vpos[v0 * 3], vpos[v0 * 3 + 1], vpos[v0 * 3 + 2]};
    const float th = t * h;                 // (rounded product, then rounded sum: contraction is off in this file)
    p[axis] = p[axis] + th;
    verts[i * 3] = p[0];
    verts[i * 3 + 1] = p[1];
    verts[i * 3 + 2] = p[2];
}

#define LAUNCH1D(kern, n, stream, ...)                                                              \
    do {                                                                                            \
        if ((n) > 0) {                                                                              \
            hipLaunchKernelGGL(kern, dim3(nksr_blocks((n), 256)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__); \
            NKSR_CHECK_LAUNCH();                                                                    \
        }                                                                                           \
    } while (0)

extern "C" int nksr_base_cell_flags(const int32_t* nbr, int32_t n, int32_t* flags, void* stream) {
    LAUNCH1D(k_base_cell_flags, (int64_t)n, stream, nbr, n, flags);
    return NKSR_OK;
}
extern "C" int nksr_base_cell_keys(const int32_t* ijk, const int32_t* sel, int64_t nsel, int upsample, int64_t* cell_keys, void* stream) {
    if (upsample < 1 || upsample > 8) return nksr_set_error(NKSR_ERR_ARG, "grid_upsample must be in [1,8]");
    LAUNCH1D(k_base_cell_keys, nsel * upsample * upsample * upsample, stream, ijk, sel, nsel, upsample, cell_keys);
    return NKSR_OK;
}
extern "C" int nksr_level_cell_keys(const int32_t* ijk, const int32_t* sel, int64_t nsel, int level, int upsample, int64_t* cell_keys, void* stream) {
    if (upsample < 1 || upsample > 8) return nksr_set_error(NKSR_ERR_ARG, "grid_upsample must be in [1,8]");
    if (level < 0 || level >= NKSR_MAX_DEPTH || (upsample << level) > 64) return nksr_set_error(NKSR_ERR_ARG, "bad level / upsample");
    const int64_t S = (int64_t)upsample << level;
    LAUNCH1D(k_level_cell_keys, nsel * S * S * S, stream, ijk, sel, nsel, level, upsample, cell_keys);
    return NKSR_OK;
}
extern "C" int nksr_cell_corner_keys(const int64_t* cell_keys, int64_t ncell, int64_t* corner_keys, void* stream) {
    LAUNCH1D(k_cell_corner_keys, ncell * 8, stream, cell_keys, ncell, corner_keys);
    return NKSR_OK;
}
extern "C" int nksr_cell_children(const int64_t* cell_keys, const int32_t* sel, int64_t nsel, int64_t* child_keys, void* stream) {
    LAUNCH1D(k_cell_children, nsel * 8, stream, cell_keys, sel, nsel, child_keys);
    return NKSR_OK;
}
extern "C" int nksr_lattice_positions(const int64_t* vkeys, int64_t n, float h, float half_w0, float* xyz_out, void* stream) {
    LAUNCH1D(k_lattice_positions, n, stream, vkeys, n, h, half_w0, xyz_out);
    return NKSR_OK;
}
extern "C" int nksr_cell_config(const int32_t* corner_idx, const float* f, int64_t ncell, int32_t* config, int32_t* ntri, void* stream) {
    LAUNCH1D(k_cell_config, ncell, stream, corner_idx, f, ncell, config, ntri);
    return NKSR_OK;
}
extern "C" int nksr_cell_active_flags(const int32_t* config, int64_t ncell, int32_t* flags, void* stream) {
    LAUNCH1D(k_cell_active_flags, ncell, stream, config, ncell, flags);
    return NKSR_OK;
}
extern "C" int nksr_mc_emit(const int32_t* corner_idx, const int32_t* config, const int32_t* tri_offset, int64_t ncell,
                            int64_t* edge_keys, void* stream) {
    LAUNCH1D(k_mc_emit, ncell, stream, corner_idx, config, tri_offset, ncell, edge_keys);
    return NKSR_OK;
}
extern "C" int nksr_mc_vertices(const int64_t* edge_keys, int64_t nedge, const int64_t* vkeys, const int64_t* vhash_keys,
                                const int32_t* vhash_vals, int32_t vhash_cap, const float* vpos, const float* f, float h, float* verts_out,
                                void* stream) {
    LAUNCH1D(k_mc_vertices, nedge, stream, edge_keys, nedge, vkeys, vhash_keys, vhash_vals, vhash_cap, vpos, f, h, verts_out);
    return NKSR_OK;
}

// ---- MISE hanging-vertex constraint ------------------------------------------------------------------
// A refined lattice vertex that sits on a coarse edge (one odd coordinate) or coarse face (two odd
// coordinates) is "hanging" unless EVERY coarse cell sharing that edge / face was refined too.  Its
// value is then replaced by the mean of the coarse end points / face corners: an unrefined neighbour
// has equal-sign corners, so no sign change can appear on the shared face and the refined mesh closes
// against it (no T-junction cracks).
__global__ void k_mise_constrain(const int64_t* __restrict__ vkeys_fine, int64_t nv, float* __restrict__ f_fine,
                                 const int64_t* __restrict__ ch_keys, const int32_t* __restrict__ ch_vals, int ch_cap,
                                 const float* __restrict__ f_coarse, const int64_t* __restrict__ ah_keys,
                                 const int32_t* __restrict__ ah_vals, int ah_cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int g[3];
    morton_decode_biased(vkeys_fine[i], NKSR_BIAS0, g[0], g[1], g[2]);
    const int odd[3] = {g[0] & 1, g[1] & 1, g[2] & 1};
    const int k = odd[0] + odd[1] + odd[2];
    if (k == 3) return;
    if (k == 0) {   // coincides with a coarse vertex: inherit its (possibly constrained) value
        const int j = hash_find(ch_keys, ch_vals, ch_cap, morton_biased(g[0] >> 1, g[1] >> 1, g[2] >> 1, NKSR_BIAS0));
        if (j >= 0) f_fine[i] = f_coarse[j];
        return;
    }
    // coarse cells sharing the edge / face: along an ODD axis the cell is fixed ((g-1)/2), along an EVEN
    // axis both cells g/2 - 1 and g/2 touch it
    int lo[3], cnt[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = odd[a] ? (g[a] - 1) >> 1 : (g[a] >> 1) - 1;
        cnt[a] = odd[a] ? 1 : 2;
    }
    bool all_active = true;
    for (int x = 0; x < cnt[0]; ++x)
        for (int y = 0; y < cnt[1]; ++y)
            for (int z = 0; z < cnt[2]; ++z)
                all_active = all_active && hash_find(ah_keys, ah_vals, ah_cap, morton_biased(lo[0] + x, lo[1] + y, lo[2] + z, NKSR_BIAS0)) >= 0;
    if (all_active) return;
    // mean over the coarse vertices spanning the edge / face: odd axes take both (g-1)/2 and (g+1)/2
    float s = 0.f;
    int n = 0;
    for (int x = 0; x <= odd[0]; ++x)
        for (int y = 0; y <= odd[1]; ++y)
            for (int z = 0; z <= odd[2]; ++z) {
                const int cx = odd[0] ? ((g[0] - 1) >> 1) + x : g[0] >> 1, cy = odd[1] ? ((g[1] - 1) >> 1) + y : g[1] >> 1,
                          cz = odd[2] ? ((g[2] - 1) >> 1) + z : g[2] >> 1;
                const int j = hash_find(ch_keys, ch_vals, ch_cap, morton_biased(cx, cy, cz, NKSR_BIAS0));
                if (j >= 0) { s += f_coarse[j]; ++n; }
            }
    if (n == (1 << k)) f_fine[i] = s * (k == 1 ? 0.5f : 0.25f);
}

extern "C" int nksr_mise_constrain(const int64_t* vkeys_fine, int64_t nv, float* f_fine, const int64_t* chash_keys, const int32_t* chash_vals,
                                   int32_t chash_cap, const float* f_coarse, const int64_t* ahash_keys, const int32_t* ahash_vals,
                                   int32_t ahash_cap, void* stream) {
    LAUNCH1D(k_mise_constrain, nv, stream, vkeys_fine, nv, f_fine, chash_keys, chash_vals, chash_cap, f_coarse, ahash_keys, ahash_vals, ahash_cap);
    return NKSR_OK;
}

// ---- marching cubes on the ADAPTIVE dual graph (specification: oracle/dual_adaptive.py) ------------------------------------------------
// One hexahedron per octree corner k, its corners the cells around k; a cell that fills several octants (k on a face or an edge of
// a larger cell) makes the hexahedron degenerate, which the 256-case table handles: equal values on a collapsed edge never cut it.
// All integer work; positions with the two roundings of k_lattice_positions (contraction is off in this file).
__global__ void k_adaptive_corner_keys(const int64_t* __restrict__ cell_keys, int64_t ncell, int lam, int64_t* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncell * 8) return;
    int x, y, z;
    morton_decode_biased(cell_keys[t >> 3], NKSR_BIAS0, x, y, z);
    const int c = (int)(t & 7), s = 1 << lam;
    out[t] = morton_biased((x + (c >> 2)) * s - 1, (y + ((c >> 1) & 1)) * s - 1, (z + (c & 1)) * s - 1, NKSR_BIAS0);
}

__global__ void k_adaptive_dual_cells(const int64_t* __restrict__ corner_keys, int64_t ncorner, nksr_cell_table_t T,
                                      int32_t* __restrict__ cidx) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncorner * 8) return;
    int x, y, z;
    morton_decode_biased(corner_keys[t >> 3], NKSR_BIAS0, x, y, z);
    const int c = (int)(t & 7);
    x += c >> 2; y += (c >> 1) & 1; z += c & 1;
    int id = -1;
    for (int l = 0; l < T.nlev; ++l) {                    // (l is uniform: the table is read through scalar loads)
        if (id >= 0) continue;
        const int sh = T.lam[l];
        const int j = hash_find(T.hkeys[l], T.hvals[l], T.hcap[l], morton_biased(x >> sh, y >> sh, z >> sh, NKSR_BIAS0));
        if (j >= 0) id = T.offset[l] + j;
    }
    cidx[t] = id;
}

__global__ void k_adaptive_positions(const int64_t* __restrict__ cell_keys, const int32_t* __restrict__ cell_lam, int64_t n, float u,
                                     float* __restrict__ xyz) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, z;
    morton_decode_biased(cell_keys[i], NKSR_BIAS0, x, y, z);
    const int s = 1 << cell_lam[i];
    const float half = (float)s * (0.5f * u);             // exact: powers of two times u
    const float px = (float)(x * s) * u, py = (float)(y * s) * u, pz = (float)(z * s) * u;
    xyz[i * 3] = px + half;
    xyz[i * 3 + 1] = py + half;
    xyz[i * 3 + 2] = pz + half;
}

__global__ void k_mc_emit_pairs(const int32_t* __restrict__ corner_idx, const int32_t* __restrict__ config,
                                const int32_t* __restrict__ tri_offset, int64_t ncell, int64_t* __restrict__ pair_keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncell) return;
    const int cfg = config[i];
    const int nt = MC_NTRI[cfg];
    int64_t o = (int64_t)tri_offset[i] * 3;
    for (int t = 0; t < nt * 3; ++t) {
        const int e = MC_TRI[cfg][t];
        const int lo = MC_EDGE_LO[e], axis = MC_EDGE_AXIS[e];
        const int64_t a = corner_idx[i * 8 + lo], b = corner_idx[i * 8 + (lo | (4 >> axis))];
        pair_keys[o + t] = (a << 33) | ((int64_t)axis << 31) | b;
    }
}

__global__ void k_pair_vertices(const int64_t* __restrict__ pair_keys, int64_t npair, const int64_t* __restrict__ cell_keys,
                                const int32_t* __restrict__ cell_lam, const float* __restrict__ cell_pos, const float* __restrict__ f,
                                float u, float* __restrict__ verts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npair) return;
    const int64_t k = pair_keys[i];
    const int64_t a = k >> 33, b = k & 0x7FFFFFFFll;
    int ca[3], cb[3];
    morton_decode_biased(cell_keys[a], NKSR_BIAS0, ca[0], ca[1], ca[2]);
    morton_decode_biased(cell_keys[b], NKSR_BIAS0, cb[0], cb[1], cb[2]);
    const int sa = 1 << cell_lam[a], sb = 1 << cell_lam[b];
    const float fa = f[a], fb = f[b];
    const float t = fa / (fa - fb);
    const float hu = 0.5f * u;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const int d2 = 2 * (cb[x] * sb - ca[x] * sa) + (sb - sa);          // doubled centre difference in fine units: exact
        const float d = (float)d2 * hu;
        const float td = t * d;                                            // (rounded product, then rounded sum)
        verts[i * 3 + x] = cell_pos[a * 3 + x] + td;
    }
}

extern "C" int nksr_adaptive_corner_keys(const int64_t* cell_keys, int64_t ncell, int lam, int64_t* corner_keys_out, void* stream) {
    if (lam < 0 || lam > 20) return nksr_set_error(NKSR_ERR_ARG, "cell size exponent out of range");
    LAUNCH1D(k_adaptive_corner_keys, ncell * 8, stream, cell_keys, ncell, lam, corner_keys_out);
    return NKSR_OK;
}
extern "C" int nksr_adaptive_dual_cells(const int64_t* corner_keys, int64_t ncorner, const nksr_cell_table_t* table, int32_t* cidx_out,
                                        void* stream) {
    if (!table || table->nlev < 1 || table->nlev > NKSR_CELL_SIZES) return nksr_set_error(NKSR_ERR_ARG, "cell table: 1..%d sizes", NKSR_CELL_SIZES);
    for (int l = 0; l < table->nlev; ++l)
        if (!table->hkeys[l] || !table->hvals[l] || table->hcap[l] < 8 || table->lam[l] < 0 || table->lam[l] > 20)
            return nksr_set_error(NKSR_ERR_ARG, "cell table: size %d incomplete", l);
    LAUNCH1D(k_adaptive_dual_cells, ncorner * 8, stream, corner_keys, ncorner, *table, cidx_out);
    return NKSR_OK;
}
extern "C" int nksr_adaptive_positions(const int64_t* cell_keys, const int32_t* cell_lam, int64_t ncell, float u, float* xyz_out, void* stream) {
    LAUNCH1D(k_adaptive_positions, ncell, stream, cell_keys, cell_lam, ncell, u, xyz_out);
    return NKSR_OK;
}
extern "C" int nksr_mc_emit_pairs(const int32_t* corner_idx, const int32_t* config, const int32_t* tri_offset, int64_t ncell,
                                  int64_t* pair_keys, void* stream) {
    LAUNCH1D(k_mc_emit_pairs, ncell, stream, corner_idx, config, tri_offset, ncell, pair_keys);
    return NKSR_OK;
}
extern "C" int nksr_pair_vertices(const int64_t* pair_keys, int64_t npair, const int64_t* cell_keys, const int32_t* cell_lam,
                                  const float* cell_pos, const float* f, float u, float* verts_out, void* stream) {
    LAUNCH1D(k_pair_vertices, npair, stream, pair_keys, npair, cell_keys, cell_lam, cell_pos, f, u, verts_out);
    return NKSR_OK;
}
