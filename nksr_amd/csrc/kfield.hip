// Neural-kernel evaluation on the sparse voxel hierarchy:
//   K_d(x, c_j) = <phi_d(x), psi_j> * B((x - c_j)/w_d),  phi_d = t + MLP_d(t), t = trilerp(feat_d)(x)
// Serves KernelField(...) / evaluate_f (reference call sites models/nksr_net.py:91-96,
// models/loss.py:189-198).  Gather-bound: per (site, level) 8 feature gathers for the
// trilinear stencil and 27 psi gathers through the neighbour table; the interpolator MLP
// weights live in LDS.
#include "kfield_dev.h"
#include <stdlib.h>

// ---- psi_j = feat_j + MLP(feat_j) -----------------------------------------------------------
template <int K, int H>
__global__ void k_voxel_psi(const float* __restrict__ feat, int n, const float* __restrict__ mlp, float* __restrict__ psi) {
    __shared__ float w[MlpView<K, H>::SIZE];
    for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += blockDim.x) w[i] = mlp[i];
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MlpView<K, H> m(w);
    float t[K], phi[K];
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = feat[(int64_t)i * K + k];
    mlp_residual<K, H, false>(m, t, nullptr, phi, nullptr);
#pragma unroll
    for (int k = 0; k < K; ++k) psi[(int64_t)i * K + k] = phi[k];
}

// 27 consecutive floats at a 4-byte aligned address: 6 x 16 bytes + 3 words
__device__ __forceinline__ void store_row27(float* __restrict__ p, const float v[27]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        f32x4_u t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4_u*>(p + 4 * q) = t;
    }
    p[24] = v[24]; p[25] = v[25]; p[26] = v[26];
}

// ---- dense-slot kernel rows -------------------------------------------------------------------
// grid.y = level; one thread per site.
template <int K, int H, bool GRAD, bool JAC>
__global__ void __launch_bounds__(128) k_kernel_rows(nksr_hier_t hier, const float* __restrict__ xyz, int64_t n, float row_scale_,
                              const float* __restrict__ site_scale, int64_t level_stride, const int32_t* __restrict__ row_index, int32_t* __restrict__ row_cells,
                              float* __restrict__ val, float* __restrict__ dval, uint32_t level_map) {
    const int d = (level_map >> (4 * blockIdx.y)) & 15, L = hier.depth;
    const nksr_level_t& lv = hier.lv[d];
    __shared__ float w[MlpView<K, H>::SIZE];
    for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += blockDim.x) w[i] = lv.mlp[i];
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    const float row_scale = site_scale ? site_scale[i] : row_scale_;      // per-site factors (batched chunks: sqrt of the chunk's weight)
    SiteCell sc = locate_site(lv, d, hier.inv_w0, x);
    // site-major [n, (3,) L, 27] for the assembly, level-major [L, stride, 27] (row = site * ncomp + component) for the matrix-free solve
    // (level-major rows can be scattered: row_index[i] = first row of site i, so that several site sets share one Morton-ordered row list)
    const int64_t vr = row_index ? row_index[i] : i, gr = row_index ? row_index[i] : i * 3;
    float* vrow = val + (level_stride ? ((int64_t)d * level_stride + vr) * 27 : (i * L + d) * 27);
    float* grow[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) grow[a] = dval + (level_stride ? ((int64_t)d * level_stride + gr + a) * 27 : ((i * 3 + a) * L + d) * 27);
    if (row_cells && level_stride) {                // the level-d cell of the site's rows (global unknown index, -1 = none)
        const int cj = sc.cell >= 0 ? lv.offset + sc.cell : -1;
        int32_t* rc = row_cells + (int64_t)d * level_stride;
        if (val) rc[vr] = cj;
        if (GRAD) { rc[gr] = cj; rc[gr + 1] = cj; rc[gr + 2] = cj; }
    }
    if (sc.cell < 0) {
        if (val)
            for (int s = 0; s < 27; ++s) vrow[s] = 0.f;
        if (GRAD)
            for (int a = 0; a < 3; ++a)
                for (int s = 0; s < 27; ++s) grow[a][s] = 0.f;
        return;
    }
    float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
    float t[K], phi[K], Jt[JAC ? K : 1][3], J[JAC ? K : 1][3];
    int nb[27];
    load_nbr_row(lv.nbr + (int64_t)sc.cell * 27, nb);
    trilerp_feat_row<K, JAC>(lv, sc, nb, inv_w, t, Jt);
    MlpView<K, H> m(w);
    mlp_residual<K, H, JAC>(m, t, Jt, phi, J);
    float bw[3][3], bd[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) bspline3(sc.u[a], bw[a], bd[a]);
    // All 27 (x 4 with gradients) results stay in registers and every output row leaves as one burst of
    // 16-byte stores: written word by word across the slot loop, the 108-byte rows kept ~10^6 partially
    // filled cache lines in flight and the kernel ran at 0.5 TB/s of useful stores.
    float ov[27], og[GRAD ? 3 : 1][27];
#pragma unroll
    for (int s = 0; s < 27; ++s) {
        const int j = nb[s];
        const int ox = s / 9, oy = (s / 3) % 3, oz = s % 3;
        float v = 0.f, g[3] = {0.f, 0.f, 0.f};
        if (j >= 0) {
            const float* ps = lv.psi + (int64_t)j * K;
            float dot = 0.f, jd[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float pk = ps[k];
                dot = fmaf(phi[k], pk, dot);
                if (JAC) { jd[0] = fmaf(J[k][0], pk, jd[0]); jd[1] = fmaf(J[k][1], pk, jd[1]); jd[2] = fmaf(J[k][2], pk, jd[2]); }
            }
            float bx = bw[0][ox], by = bw[1][oy], bz = bw[2][oz];
            float B = bx * by * bz;
            v = dot * B;
            if (GRAD) {
                g[0] = dot * (bd[0][ox] * by * bz * inv_w);
                g[1] = dot * (bx * bd[1][oy] * bz * inv_w);
                g[2] = dot * (bx * by * bd[2][oz] * inv_w);
                if (JAC) { g[0] = fmaf(jd[0], B, g[0]); g[1] = fmaf(jd[1], B, g[1]); g[2] = fmaf(jd[2], B, g[2]); }
            }
        }
        ov[s] = v * row_scale;
        if (GRAD) { og[0][s] = g[0] * row_scale; og[1][s] = g[1] * row_scale; og[2][s] = g[2] * row_scale; }
    }
    if (val) store_row27(vrow, ov);
    if (GRAD) {
#pragma unroll
        for (int a = 0; a < 3; ++a) store_row27(grow[a], og[a]);
    }
}

// ---- rank-K FACTORS of the kernel rows (kernel_dim 4; the matrix-free solve, csrc/fused.hip) --------------------------------------
// A dense-slot row is a rank-K object:  row[s] = B_s(u) <phi, psi_s>  (position rows),  d/dx_a: <phi, psi_s> dB_s/dx_a + <J_a, psi_s> B_s
// with phi = phi_d(x) (K floats), J_a = d phi / d x_a, u = the local coordinate of x in its level-d cell -- and u follows from the
// ONE fp32 product p = x * inv_w0 at every level (u_d = frac(p 2^-d)).  Instead of 108 bytes per row and level this writes
//   vec[d][row] (16 bytes):  position row: phi;  a normal site owns FOUR rows: a header row (phi) and one row per axis (J_a)
//   pos[row]    (16 bytes):  p (3 floats) + the row's kind (int bits: 0 position, 1 header, 2 + a gradient row of axis a)
// all pre-multiplied by sqrt(weight); the sweep rebuilds the 27 slots in registers from them and the psi stencil of the cell.
// The header row is a row of the operator like any other (all-zero: it contributes nothing, its target is 0).
// grid.y = level; one thread per site.
template <int H, bool GRAD, bool JAC>
__global__ void __launch_bounds__(128) k_kernel_factors(nksr_hier_t hier, const float* __restrict__ xyz, int64_t n, float row_scale_,
                              const float* __restrict__ site_scale, int64_t level_stride, const int32_t* __restrict__ row_index,
                              int32_t* __restrict__ row_cells, float4* __restrict__ vec, float4* __restrict__ pos) {
    constexpr int K = 4;
    const int d = blockIdx.y;
    const nksr_level_t& lv = hier.lv[d];
    __shared__ float w[MlpView<K, H>::SIZE];
    for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += blockDim.x) w[i] = lv.mlp[i];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    const float row_scale = site_scale ? site_scale[i] : row_scale_;
    const SiteCell sc = locate_site(lv, d, hier.inv_w0, x);
    constexpr int NR = GRAD ? 4 : 1;
    const int64_t r0 = row_index ? row_index[i] : i * NR;
    if (d == 0) {
        const float px = __fmul_rn(x[0], hier.inv_w0), py = __fmul_rn(x[1], hier.inv_w0), pz = __fmul_rn(x[2], hier.inv_w0);
#pragma unroll
        for (int q = 0; q < NR; ++q) pos[r0 + q] = make_float4(px, py, pz, __int_as_float(GRAD ? 1 + q : 0));
    }
    if (row_cells) {
        const int cj = sc.cell >= 0 ? lv.offset + sc.cell : -1;
#pragma unroll
        for (int q = 0; q < NR; ++q) row_cells[(int64_t)d * level_stride + r0 + q] = cj;
    }
    float4* out = vec + (int64_t)d * level_stride + r0;
    if (sc.cell < 0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) out[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
    float t[K], phi[K], Jt[JAC ? K : 1][3], J[JAC ? K : 1][3];
    trilerp_feat<K, JAC>(lv, d, sc, inv_w, t, Jt);
    MlpView<K, H> m(w);
    mlp_residual<K, H, JAC>(m, t, Jt, phi, J);
    out[0] = make_float4(phi[0] * row_scale, phi[1] * row_scale, phi[2] * row_scale, phi[3] * row_scale);
    if (GRAD) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
            out[1 + a] = JAC ? make_float4(J[0][a] * row_scale, J[1][a] * row_scale, J[2][a] * row_scale, J[3][a] * row_scale)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- dispatch on (K, H) ------------------------------------------------------------------------
extern "C" int nksr_voxel_psi(const float* feat, int32_t n, int kdim, int hidden, const float* mlp, float* psi_out,
                              void* stream) {
    if (n <= 0) return NKSR_OK;
    DISPATCH_KH(kdim, hidden, {
        hipLaunchKernelGGL((k_voxel_psi<K, H>), dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, feat, n, mlp, psi_out);
    })
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// NKSR_ROWS_LEVELS (probe, tools/rows_probe.py): only these levels ("0,2"); the rows of the others stay unwritten
uint32_t nksr_rows_level_map(int depth, int* nlev) {
    uint32_t level_map = 0;
    int n = 0;
    const char* e = getenv("NKSR_ROWS_LEVELS");
    if (e && *e) {
        for (const char* p = e; *p; ++p)
            if (*p >= '0' && *p <= '9' && (*p - '0') < depth) level_map |= (uint32_t)(*p - '0') << (4 * n++);
    } else {
        for (int d = 0; d < depth; ++d) level_map |= (uint32_t)d << (4 * n++);
    }
    *nlev = n;
    return level_map;
}
extern "C" int nksr_kernel_rows(const nksr_hier_t* h, const float* xyz, int64_t n, int approx, float row_scale, const float* site_scale,
                                int64_t level_stride, const int32_t* row_index, int32_t* row_cells, float* val, float* dval, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!val && !dval) return nksr_set_error(NKSR_ERR_ARG, "val and dval are both NULL");
    if ((row_index || row_cells) && (level_stride <= 0 || (val && dval)))
        return nksr_set_error(NKSR_ERR_ARG, "row_index / row_cells need the level-major layout and ONE kind of rows (val or dval)");
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    int nlev = 0;
    const uint32_t level_map = nksr_rows_level_map(h->depth, &nlev);
    dim3 grid(nksr_blocks(n, 128), nlev), block(128);
    DISPATCH_KH(h->kdim, h->hidden, {
        if (!dval) hipLaunchKernelGGL((k_kernel_rows<K, H, false, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, val, dval, level_map);
        else if (approx) hipLaunchKernelGGL((k_kernel_rows<K, H, true, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, val, dval, level_map);
        else hipLaunchKernelGGL((k_kernel_rows<K, H, true, true>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, val, dval, level_map);
    })
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_kernel_factors(const nksr_hier_t* h, const float* xyz, int64_t n, int grad, int approx, float row_scale,
                                   const float* site_scale, int64_t level_stride, const int32_t* row_index, int32_t* row_cells,
                                   float* vec_out, float* pos_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!h || !xyz || !vec_out || !pos_out || level_stride <= 0) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays / level_stride <= 0");
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (h->kdim != 4 || (h->hidden != 16 && h->hidden != 32))
        return nksr_set_error(NKSR_ERR_ARG, "kernel factors need kernel_dim 4 and hidden_dim 16 / 32 (got %d, %d)", h->kdim, h->hidden);
    if (((uintptr_t)vec_out | (uintptr_t)pos_out) & 15) return nksr_set_error(NKSR_ERR_ARG, "factor arrays must be 16-byte aligned");
    dim3 grid(nksr_blocks(n, 128), h->depth), block(128);
    float4* vec = (float4*)vec_out;
    float4* pos = (float4*)pos_out;
#define NKSR_LAUNCH_FACTORS(H_)                                                                                                              \
    do {                                                                                                                                     \
        if (!grad) hipLaunchKernelGGL((k_kernel_factors<H_, false, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, vec, pos); \
        else if (approx) hipLaunchKernelGGL((k_kernel_factors<H_, true, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, vec, pos); \
        else hipLaunchKernelGGL((k_kernel_factors<H_, true, true>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, site_scale, level_stride, row_index, row_cells, vec, pos); \
    } while (0)
    if (h->hidden == 16) NKSR_LAUNCH_FACTORS(16); else NKSR_LAUNCH_FACTORS(32);
#undef NKSR_LAUNCH_FACTORS
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- vector-Jacobian products of the kernel rows w.r.t. the basis features and the interpolator weights ---------------------------
// Training path only (models/nksr_net.py:105-112: the loss back-propagates through solve_non_fused / evaluate_f into the network's
// basis features and into the interpolators): with per-row factors  g_r[s] = a_r lambda[col] + b_r alpha[col]  this computes
//     dS/dtheta,  S = sum_r sum_s R'_r[s] g_r[s],   R' = row_scale * (dense-slot kernel rows of nksr_kernel_rows)
// -- the theta term of the implicit-function backward of the solve (a = t' - u, b = -v) and of evaluate_f (a = 0, b = dL/df),
// fields/kernel_field.py _SolveFunction / _EvaluateFunction.  One thread per (site, level), as the rows themselves are made: the
// forward of the row is recomputed (cell, trilinear stencil, interpolator with its forward-mode tangents, psi gathers), then
//   * dS/dpsi_j for the 27 neighbours            -> gpsi   (the voxel's own feature through the interpolator: k_psi_vjp)
//   * reverse mode through phi = t + MLP(t) and, for exact-gradient rows, through J = Jt + W3 D2 W2 D1 W1 Jt (the ReLU masks D
//     are the constants they are almost everywhere) -> weight cotangents (LDS accumulators, one global add per weight and
//     workgroup) and dS/dt, dS/dJt
//   * the transpose of the trilinear stencil     -> gfeat.
// Accumulation is by floating-point atomics (hardware global_atomic_add_f32 / ds_add_f32): gradients are reproducible to fp32
// rounding of a sum, not bit for bit -- unlike every kernel of the solve-time path.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
__device__ __forceinline__ void vjp_add(float* p, float v) { unsafeAtomicAdd(p, v); }
// weight cotangents: the 64 lanes of the (single-wavefront) workgroup add into ONE LDS word -- a butterfly sum and one plain add by
// lane 0 instead of 64 serialised LDS atomics (k_psi_vjp: 416 -> 90 us per 6 000 voxels, k_rows_vjp 850 -> 530).  Every lane must arrive (no divergence around
// the call): lanes without work carry zeros.
__device__ __forceinline__ void wave_acc(float* lds, float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) *lds += v;
}

template <int K, int H, bool JAC>
__device__ __forceinline__ void mlp_residual_vjp(const MlpView<K, H>& m, float* __restrict__ gw, const float t[K], const float Jt[JAC ? K : 1][3],
                                                 const float gphi[K], const float gJ[JAC ? K : 1][3], float gt[K], float gJt[JAC ? K : 1][3]) {
    // (called by ALL lanes of the wavefront, converged: the weight cotangents are summed over the lanes, wave_acc)
    float* gW1 = gw;
    float* gb1 = gW1 + H * K;
    float* gW2 = gb1 + H;
    float* gb2 = gW2 + H * H;
    float* gW3 = gb2 + H;
    float* gb3 = gW3 + K * H;
    float h1[H], d1[JAC ? H : 1][3], gh1[H], gd1[JAC ? H : 1][3];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float a = m.b1[h];
        float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = m.W1[h * K + k];
            a = fmaf(w, t[k], a);
            if (JAC) { da[0] = fmaf(w, Jt[k][0], da[0]); da[1] = fmaf(w, Jt[k][1], da[1]); da[2] = fmaf(w, Jt[k][2], da[2]); }
        }
        const bool on = a > 0.f;
        h1[h] = on ? a : 0.f;
        gh1[h] = 0.f;
        if (JAC) {
            d1[h][0] = on ? da[0] : 0.f; d1[h][1] = on ? da[1] : 0.f; d1[h][2] = on ? da[2] : 0.f;
            gd1[h][0] = gd1[h][1] = gd1[h][2] = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gt[k] = gphi[k];                                      // the residual skip
        if (JAC) { gJt[k][0] = gJ[k][0]; gJt[k][1] = gJ[k][1]; gJt[k][2] = gJ[k][2]; }
        wave_acc(gb3 + k, gphi[k]);
    }
#pragma unroll 1
    for (int g = 0; g < H; ++g) {
        float a = m.b2[g];
        float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float w = m.W2[g * H + h];
            a = fmaf(w, h1[h], a);
            if (JAC) { da[0] = fmaf(w, d1[h][0], da[0]); da[1] = fmaf(w, d1[h][1], da[1]); da[2] = fmaf(w, d1[h][2], da[2]); }
        }
        const float on = a > 0.f ? 1.f : 0.f;                 // unit off: h2 = d2 = 0 and nothing flows back through it
        const float h2 = on * a;
        float gh2 = 0.f, ge[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = m.W3[k * H + g];
            gh2 = fmaf(w, gphi[k], gh2);
            float gw3 = gphi[k] * h2;
            if (JAC) {
                ge[0] = fmaf(w, gJ[k][0], ge[0]); ge[1] = fmaf(w, gJ[k][1], ge[1]); ge[2] = fmaf(w, gJ[k][2], ge[2]);
                gw3 += on * (gJ[k][0] * da[0] + gJ[k][1] * da[1] + gJ[k][2] * da[2]);
            }
            wave_acc(gW3 + k * H + g, gw3);
        }
        gh2 *= on;
        if (JAC) { ge[0] *= on; ge[1] *= on; ge[2] *= on; }
        wave_acc(gb2 + g, gh2);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float w = m.W2[g * H + h];
            float gw2 = gh2 * h1[h];
            gh1[h] = fmaf(w, gh2, gh1[h]);
            if (JAC) {
                gw2 += ge[0] * d1[h][0] + ge[1] * d1[h][1] + ge[2] * d1[h][2];
                gd1[h][0] = fmaf(w, ge[0], gd1[h][0]); gd1[h][1] = fmaf(w, ge[1], gd1[h][1]); gd1[h][2] = fmaf(w, ge[2], gd1[h][2]);
            }
            wave_acc(gW2 + g * H + h, gw2);
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float on = h1[h] > 0.f ? 1.f : 0.f;             // (a1 > 0  <=>  h1 > 0)
        const float ga = on * gh1[h];
        float ge[3] = {0.f, 0.f, 0.f};
        if (JAC) { ge[0] = on * gd1[h][0]; ge[1] = on * gd1[h][1]; ge[2] = on * gd1[h][2]; }
        wave_acc(gb1 + h, ga);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = m.W1[h * K + k];
            float gw1 = ga * t[k];
            gt[k] = fmaf(w, ga, gt[k]);
            if (JAC) {
                gw1 += ge[0] * Jt[k][0] + ge[1] * Jt[k][1] + ge[2] * Jt[k][2];
                gJt[k][0] = fmaf(w, ge[0], gJt[k][0]); gJt[k][1] = fmaf(w, ge[1], gJt[k][1]); gJt[k][2] = fmaf(w, ge[2], gJt[k][2]);
            }
            wave_acc(gW1 + h * K + k, gw1);
        }
    }
}

struct VjpOut { float* gfeat[NKSR_MAX_DEPTH]; float* gpsi[NKSR_MAX_DEPTH]; float* gmlp[NKSR_MAX_DEPTH]; };

template <int K, int H, bool GRAD, bool JAC>
__global__ void __launch_bounds__(64) k_rows_vjp(nksr_hier_t hier, const float* __restrict__ xyz, int64_t n, float sw, const float* __restrict__ ca,
                                                 const float* __restrict__ cb, const float* __restrict__ alpha, const float* __restrict__ lam, VjpOut out) {
    const int d = blockIdx.y;
    const nksr_level_t& lv = hier.lv[d];
    __shared__ float w[MlpView<K, H>::SIZE];
    __shared__ float gw[MlpView<K, H>::SIZE];
    for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += 64) { w[i] = lv.mlp[i]; gw[i] = 0.f; }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    MlpView<K, H> m(w);
    const float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
    // a lane without a row (past the end, or its site lies in no active cell of this level) carries zeros through the weight part
    bool valid = false;
    SiteCell sc;
    int nb[27];
    float t[K], Jt[JAC ? K : 1][3], gphi[K], gJ[JAC ? K : 1][3];
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = 0.f; gphi[k] = 0.f; if (JAC) { Jt[k][0] = Jt[k][1] = Jt[k][2] = 0.f; gJ[k][0] = gJ[k][1] = gJ[k][2] = 0.f; } }
    if (i < n && lv.n > 0) {
        const float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
        sc = locate_site(lv, d, hier.inv_w0, x);
        valid = sc.cell >= 0;
    }
    if (valid) {
        float phi[K], J[JAC ? K : 1][3];
        trilerp_feat<K, JAC>(lv, d, sc, inv_w, t, Jt);
        mlp_residual<K, H, JAC>(m, t, Jt, phi, J);
        float bw[3][3], bd[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) bspline3(sc.u[a], bw[a], bd[a]);
        load_nbr_row(lv.nbr + (int64_t)sc.cell * 27, nb);
        float fa[GRAD ? 3 : 1], fb[GRAD ? 3 : 1];
#pragma unroll
        for (int a = 0; a < (GRAD ? 3 : 1); ++a) {
            fa[a] = ca ? ca[i * (GRAD ? 3 : 1) + a] : 0.f;
            fb[a] = cb ? cb[i * (GRAD ? 3 : 1) + a] : 0.f;
        }
#pragma unroll 1
        for (int s = 0; s < 27; ++s) {
            const int j = nb[s];
            if (j < 0) continue;
            const int ox = s / 9, oy = (s / 3) % 3, oz = s % 3;
            const float bx = sel3(bw[0], ox), by = sel3(bw[1], oy), bz = sel3(bw[2], oz);
            const float B = bx * by * bz;
            const float al = alpha[lv.offset + j], la = lam ? lam[lv.offset + j] : 0.f;
            float cphi, cJ[3] = {0.f, 0.f, 0.f};
            if (!GRAD) {
                cphi = (fa[0] * la + fb[0] * al) * sw * B;
            } else {
                const float g0 = (fa[0] * la + fb[0] * al) * sw, g1 = (fa[1] * la + fb[1] * al) * sw, g2 = (fa[2] * la + fb[2] * al) * sw;
                cphi = g0 * (sel3(bd[0], ox) * by * bz * inv_w) + g1 * (bx * sel3(bd[1], oy) * bz * inv_w) + g2 * (bx * by * sel3(bd[2], oz) * inv_w);
                if (JAC) { cJ[0] = g0 * B; cJ[1] = g1 * B; cJ[2] = g2 * B; }
            }
            const float* ps = lv.psi + (int64_t)j * K;
            float* gp = out.gpsi[d] + (int64_t)j * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float pk = ps[k];
                gphi[k] = fmaf(cphi, pk, gphi[k]);
                float gps = cphi * phi[k];
                if (JAC) {
                    gJ[k][0] = fmaf(cJ[0], pk, gJ[k][0]); gJ[k][1] = fmaf(cJ[1], pk, gJ[k][1]); gJ[k][2] = fmaf(cJ[2], pk, gJ[k][2]);
                    gps += cJ[0] * J[k][0] + cJ[1] * J[k][1] + cJ[2] * J[k][2];
                }
                if (gps != 0.f) vjp_add(gp + k, gps);
            }
        }
    }
    float gt[K], gJt[JAC ? K : 1][3];
    mlp_residual_vjp<K, H, JAC>(m, gw, t, Jt, gphi, gJ, gt, gJt);           // all 64 lanes, converged
    if (valid) {
        // transpose of the trilinear stencil (the same eight corners and weights as trilerp_feat)
        float v[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = sc.u[a] + 0.5f - (float)sc.hb[a];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
            const int s = (sc.hb[0] + cx) * 9 + (sc.hb[1] + cy) * 3 + (sc.hb[2] + cz);
            const int j = nb[s];
            if (j < 0) continue;
            const float wx = cx ? v[0] : 1.f - v[0], wy = cy ? v[1] : 1.f - v[1], wz = cz ? v[2] : 1.f - v[2];
            const float wt = wx * wy * wz;
            const float gx = (cx ? 1.f : -1.f) * wy * wz * inv_w, gy = wx * (cy ? 1.f : -1.f) * wz * inv_w, gz = wx * wy * (cz ? 1.f : -1.f) * inv_w;
            float* gf = out.gfeat[d] + (int64_t)j * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float g = wt * gt[k];
                if (JAC) g += gx * gJt[k][0] + gy * gJt[k][1] + gz * gJt[k][2];
                if (g != 0.f) vjp_add(gf + k, g);
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < MlpView<K, H>::SIZE; q += 64)
        if (gw[q] != 0.f) vjp_add(out.gmlp[d] + q, gw[q]);
}

// psi_j = f_j + MLP(f_j): gfeat_j += d psi_j / d f_j ^T gpsi_j, weight cotangents as above (one thread per voxel)
template <int K, int H>
__global__ void __launch_bounds__(64) k_psi_vjp(const float* __restrict__ feat, int n, const float* __restrict__ mlp, const float* __restrict__ gpsi,
                                                float* __restrict__ gfeat, float* __restrict__ gmlp) {
    __shared__ float w[MlpView<K, H>::SIZE];
    __shared__ float gw[MlpView<K, H>::SIZE];
    for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += 64) { w[i] = mlp[i]; gw[i] = 0.f; }
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x;
    float t[K], g[K], gt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = i < n ? feat[(int64_t)i * K + k] : 0.f; g[k] = i < n ? gpsi[(int64_t)i * K + k] : 0.f; }
    MlpView<K, H> m(w);
    mlp_residual_vjp<K, H, false>(m, gw, t, nullptr, g, nullptr, gt, nullptr);          // all 64 lanes, converged
    if (i < n) {
#pragma unroll
        for (int k = 0; k < K; ++k) gfeat[(int64_t)i * K + k] += gt[k];          // (this voxel's entry: no other thread of this launch touches it)
    }
    __syncthreads();
    for (int q = threadIdx.x; q < MlpView<K, H>::SIZE; q += 64)
        if (gw[q] != 0.f) vjp_add(gmlp + q, gw[q]);
}

extern "C" int nksr_kernel_rows_vjp(const nksr_hier_t* h, const float* xyz, int64_t n, int grad_rows, int approx, float row_scale, const float* coef_a,
                                    const float* coef_b, const float* alpha, const float* lam, const nksr_theta_grad_t* out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!h || !xyz || !alpha || !out || (!coef_a && !coef_b)) return nksr_set_error(NKSR_ERR_ARG, "rows vjp: NULL arrays");
    if (coef_a && !lam) return nksr_set_error(NKSR_ERR_ARG, "rows vjp: coef_a needs lam");
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    VjpOut o;
    for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
        o.gfeat[d] = out->gfeat[d]; o.gpsi[d] = out->gpsi[d]; o.gmlp[d] = out->gmlp[d];
        if (d < h->depth && h->lv[d].n > 0 && (!o.gfeat[d] || !o.gpsi[d] || !o.gmlp[d])) return nksr_set_error(NKSR_ERR_ARG, "rows vjp: NULL output of level %d", d);
    }
    dim3 grid(nksr_blocks(n, 64), h->depth), block(64);
    DISPATCH_KH(h->kdim, h->hidden, {
        if (!grad_rows) hipLaunchKernelGGL((k_rows_vjp<K, H, false, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, coef_a, coef_b, alpha, lam, o);
        else if (approx) hipLaunchKernelGGL((k_rows_vjp<K, H, true, false>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, coef_a, coef_b, alpha, lam, o);
        else hipLaunchKernelGGL((k_rows_vjp<K, H, true, true>), grid, block, 0, (hipStream_t)stream, *h, xyz, n, row_scale, coef_a, coef_b, alpha, lam, o);
    })
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_voxel_psi_vjp(const float* feat, int32_t n, int kdim, int hidden, const float* mlp, const float* gpsi, float* gfeat, float* gmlp,
                                  void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!feat || !mlp || !gpsi || !gfeat || !gmlp) return nksr_set_error(NKSR_ERR_ARG, "psi vjp: NULL arrays");
    DISPATCH_KH(kdim, hidden, {
        hipLaunchKernelGGL((k_psi_vjp<K, H>), dim3(nksr_blocks(n, 64)), dim3(64), 0, (hipStream_t)stream, feat, n, mlp, gpsi, gfeat, gmlp);
    })
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
