// Device helpers shared by the kernel-evaluation kernels (csrc/kfield.hip, csrc/rows.hip): the interpolator phi = t + MLP(t) with
// forward-mode tangents, a site's cell geometry, the trilinear stencil of the basis features.  DESIGN.md section 2.3.
#pragma once
#include "common.h"

template <int K, int H>
struct MlpView {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    __device__ explicit MlpView(const float* w) {
        W1 = w; b1 = W1 + H * K; W2 = b1 + H; b2 = W2 + H * H; W3 = b2 + H; b3 = W3 + K * H;
    }
    static constexpr int SIZE = H * K + H + H * H + H + K * H + K;
};

// phi = t + W3 relu(W2 relu(W1 t + b1) + b2) + b3 ; optional forward-mode tangents J[K][3]
// Layers 2 and 3 are fused: every hidden unit g of layer 2 (value + 3 tangents) is folded into the K outputs as soon as it
// exists, so only layer 1 (h1, d1) and the outputs are live -- 4 (H + K) instead of 8 H + 8 K registers with tangents (the
// K=16 / H=32 instantiations spilled to scratch and the K=4 / H=16 one sat at 196 VGPRs when all three layers were arrays).
// phi / J may alias t / Jt.
// the same weights read through the scalar cache into SGPRs (constant address space, uniform addresses): they are the same for every
// lane -- as LDS reads they cost a ds_read per four weights and the registers to hold them
typedef const __attribute__((address_space(4))) float nksr_cfloat;
template <int K, int H>
struct MlpViewC {
    nksr_cfloat *W1, *b1, *W2, *b2, *W3, *b3;
    __device__ explicit MlpViewC(const float* w) {
        W1 = (nksr_cfloat*)(uintptr_t)w; b1 = W1 + H * K; W2 = b1 + H; b2 = W2 + H * H; W3 = b2 + H; b3 = W3 + K * H;
    }
};

template <int K, int H, bool JAC, typename View>
__device__ __forceinline__ void mlp_residual(const View& m, const float t[K], const float Jt[K][3],
                                             float phi[K], float J[K][3]) {
    float h1[H];
    float d1[JAC ? H : 1][3];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float a = m.b1[h];
        float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float w = m.W1[h * K + k];
            a = fmaf(w, t[k], a);
            if (JAC) { da[0] = fmaf(w, Jt[k][0], da[0]); da[1] = fmaf(w, Jt[k][1], da[1]); da[2] = fmaf(w, Jt[k][2], da[2]); }
        }
        bool on = a > 0.f;
        h1[h] = on ? a : 0.f;
        if (JAC) { d1[h][0] = on ? da[0] : 0.f; d1[h][1] = on ? da[1] : 0.f; d1[h][2] = on ? da[2] : 0.f; }
    }
    float out[K], Jo[JAC ? K : 1][3];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        out[k] = t[k] + m.b3[k];
        if (JAC) { Jo[k][0] = Jt[k][0]; Jo[k][1] = Jt[k][1]; Jo[k][2] = Jt[k][2]; }
    }
    // kept as a LOOP when tangents are carried: fully unrolled, the compiler hoists the H*H + K*H weight reads of both layers and the
    // kernel needs ~400 registers (AGPR copies, scratch for H = 32); rolled, one row of W2 and one column of W3 are live at a time
#pragma unroll 1
    for (int g = 0; g < H; ++g) {
        float a = m.b2[g];
        float da[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float w = m.W2[g * H + h];
            a = fmaf(w, h1[h], a);
            if (JAC) { da[0] = fmaf(w, d1[h][0], da[0]); da[1] = fmaf(w, d1[h][1], da[1]); da[2] = fmaf(w, d1[h][2], da[2]); }
        }
        const bool on = a > 0.f;
        const float h2 = on ? a : 0.f;
        const float e0 = on ? da[0] : 0.f, e1 = on ? da[1] : 0.f, e2 = on ? da[2] : 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float w = m.W3[k * H + g];
            out[k] = fmaf(w, h2, out[k]);
            if (JAC) { Jo[k][0] = fmaf(w, e0, Jo[k][0]); Jo[k][1] = fmaf(w, e1, Jo[k][1]); Jo[k][2] = fmaf(w, e2, Jo[k][2]); }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        phi[k] = out[k];
        if (JAC) { J[k][0] = Jo[k][0]; J[k][1] = Jo[k][1]; J[k][2] = Jo[k][2]; }
    }
}

__device__ __forceinline__ float sel3(const float w[3], int i) { return i == 0 ? w[0] : (i == 1 ? w[1] : w[2]); }

struct SiteCell {
    int cell;      // voxel index of the containing cell or -1
    int I[3];      // integer coordinates of the containing cell
    int hb[3];     // half bits
    float u[3];    // local coordinate in [0,1)
};

// the 27 neighbour indices of an active cell as seven 16-byte loads (4-byte aligned: rows are 108 bytes) instead of 27 scalar ones:
// every load instruction of these per-site kernels touches 64 different lines (one per lane), their count is what they cost
struct i32x4_u { int x, y, z, w; } __attribute__((packed, aligned(4)));
struct i32x3_u { int x, y, z; } __attribute__((packed, aligned(4)));
__device__ __forceinline__ void load_nbr_row(const int32_t* __restrict__ row, int nb[27]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const i32x4_u v = reinterpret_cast<const i32x4_u*>(row)[q];
        nb[4 * q] = v.x; nb[4 * q + 1] = v.y; nb[4 * q + 2] = v.z; nb[4 * q + 3] = v.w;
    }
    const i32x3_u v = *reinterpret_cast<const i32x3_u*>(row + 24);
    nb[24] = v.x; nb[25] = v.y; nb[26] = v.z;
}

// neighbour slot s of the site's cell: through the 27-neighbour table when the cell is active,
// through the hash otherwise (FALLBACK: query points in inactive cells still see every existing
// voxel whose support covers them -- field.evaluate_f on arbitrary positions, models/loss.py:99)
template <bool FALLBACK>
__device__ __forceinline__ int nbr_of(const nksr_level_t& lv, int level, const SiteCell& sc, int s) {
    if (sc.cell >= 0) return lv.nbr[(int64_t)sc.cell * 27 + s];
    if (!FALLBACK) return -1;
    return hash_find(lv.hkeys, lv.hvals, lv.hcap,
                     morton_biased(sc.I[0] + s / 9 - 1, sc.I[1] + (s / 3) % 3 - 1, sc.I[2] + s % 3 - 1, NKSR_BIAS0 >> level));
}

// the containing cell's integer coordinates, half bits and local coordinates (no table look-up)
__device__ __forceinline__ SiteCell site_geometry(int level, float inv_w0, const float x[3]) {
    SiteCell sc;
    float scale = __int_as_float((127 - level) << 23);  // 2^-level
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float p;
        int Hd = half_index(x[a], inv_w0, p) >> level;
        sc.I[a] = Hd >> 1;
        sc.hb[a] = Hd & 1;
        sc.u[a] = p * scale - (float)sc.I[a];
    }
    sc.cell = -1;
    return sc;
}
__device__ __forceinline__ SiteCell locate_site(const nksr_level_t& lv, int level, float inv_w0, const float x[3]) {
    SiteCell sc = site_geometry(level, inv_w0, x);
    sc.cell = hash_find(lv.hkeys, lv.hvals, lv.hcap, morton_biased(sc.I[0], sc.I[1], sc.I[2], NKSR_BIAS0 >> level));
    return sc;
}
// hash_find whose FIRST probe (key and value of the home slot) was fetched by the caller: k0 / v0
__device__ __forceinline__ int hash_find_after(const int64_t* __restrict__ hkeys, const int32_t* __restrict__ hvals, int hcap, int64_t key,
                                               uint32_t slot, int64_t k0, int v0) {
    if (k0 == key) return v0;
    if (k0 == -1) return -1;
    for (int probe = 1; probe < hcap; ++probe) {
        slot = hash_next(slot, probe, hcap);
        const int64_t k = hkeys[slot];
        if (k == key) return hvals[slot];
        if (k == -1) return -1;
    }
    return -1;
}

// trilinear interpolation of the level's basis features (+ spatial tangents in world units)
template <int K, bool JAC, bool FALLBACK = false>
__device__ __forceinline__ void trilerp_feat(const nksr_level_t& lv, int level, const SiteCell& sc, float inv_w, float t[K],
                                             float Jt[K][3]) {
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = 0.f; if (JAC) { Jt[k][0] = Jt[k][1] = Jt[k][2] = 0.f; } }
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = sc.u[a] + 0.5f - (float)sc.hb[a];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
        int s = (sc.hb[0] + cx) * 9 + (sc.hb[1] + cy) * 3 + (sc.hb[2] + cz);  // (hb-1+c)+1
        int j = nbr_of<FALLBACK>(lv, level, sc, s);
        if (j < 0) continue;
        float wx = cx ? v[0] : 1.f - v[0], wy = cy ? v[1] : 1.f - v[1], wz = cz ? v[2] : 1.f - v[2];
        float w = wx * wy * wz;
        float gx = (cx ? 1.f : -1.f) * wy * wz * inv_w, gy = wx * (cy ? 1.f : -1.f) * wz * inv_w,
              gz = wx * wy * (cz ? 1.f : -1.f) * inv_w;
        const float* f = lv.feat + (int64_t)j * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float fk = f[k];
            t[k] = fmaf(fk, w, t[k]);
            if (JAC) { Jt[k][0] = fmaf(fk, gx, Jt[k][0]); Jt[k][1] = fmaf(fk, gy, Jt[k][1]); Jt[k][2] = fmaf(fk, gz, Jt[k][2]); }
        }
    }
}

// neighbour of the trilinear corner (cx, cy, cz) out of the ALREADY LOADED neighbour row of an active cell: slot (hb0 + cx, hb1 +
// cy, hb2 + cz) -- seven selects on the half bits instead of one more dependent load per corner (the row is needed for the psi
// gathers anyway: eight of the ~44 gather instructions of a (site, level) were the same 27 words read a second time)
template <int CX, int CY, int CZ>
__device__ __forceinline__ int corner_of_row(const int nb[27], const int hb[3]) {
    constexpr int B = CX * 9 + CY * 3 + CZ;
    const int a00 = hb[2] ? nb[B + 1] : nb[B], a01 = hb[2] ? nb[B + 4] : nb[B + 3];
    const int a10 = hb[2] ? nb[B + 10] : nb[B + 9], a11 = hb[2] ? nb[B + 13] : nb[B + 12];
    const int b0 = hb[1] ? a01 : a00, b1 = hb[1] ? a11 : a10;
    return hb[0] ? b1 : b0;
}
template <int K, bool JAC>
__device__ __forceinline__ void trilerp_feat_row(const nksr_level_t& lv, const SiteCell& sc, const int nb[27], float inv_w, float t[K],
                                                 float Jt[K][3]) {
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = 0.f; if (JAC) { Jt[k][0] = Jt[k][1] = Jt[k][2] = 0.f; } }
    float v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = sc.u[a] + 0.5f - (float)sc.hb[a];
    int jc[8];
    jc[0] = corner_of_row<0, 0, 0>(nb, sc.hb); jc[1] = corner_of_row<0, 0, 1>(nb, sc.hb);
    jc[2] = corner_of_row<0, 1, 0>(nb, sc.hb); jc[3] = corner_of_row<0, 1, 1>(nb, sc.hb);
    jc[4] = corner_of_row<1, 0, 0>(nb, sc.hb); jc[5] = corner_of_row<1, 0, 1>(nb, sc.hb);
    jc[6] = corner_of_row<1, 1, 0>(nb, sc.hb); jc[7] = corner_of_row<1, 1, 1>(nb, sc.hb);
#pragma unroll
    for (int c = 0; c < 8; ++c) {             // (same corners, same order, same arithmetic as trilerp_feat)
        const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
        const int j = jc[c];
        if (j < 0) continue;
        float wx = cx ? v[0] : 1.f - v[0], wy = cy ? v[1] : 1.f - v[1], wz = cz ? v[2] : 1.f - v[2];
        float w = wx * wy * wz;
        float gx = (cx ? 1.f : -1.f) * wy * wz * inv_w, gy = wx * (cy ? 1.f : -1.f) * wz * inv_w,
              gz = wx * wy * (cz ? 1.f : -1.f) * inv_w;
        const float* f = lv.feat + (int64_t)j * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float fk = f[k];
            t[k] = fmaf(fk, w, t[k]);
            if (JAC) { Jt[k][0] = fmaf(fk, gx, Jt[k][0]); Jt[k][1] = fmaf(fk, gy, Jt[k][1]); Jt[k][2] = fmaf(fk, gz, Jt[k][2]); }
        }
    }
}

// 27 consecutive floats / one psi or feature vector at a 4-byte aligned address
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// ---- dispatch on (kernel_dim, hidden_dim) ---------------------------------------------------------
#define DISPATCH_KH(K_, H_, ...)                                  \
    if (K_ == 4 && H_ == 16) { constexpr int K = 4, H = 16; __VA_ARGS__ } \
    else if (K_ == 16 && H_ == 32) { constexpr int K = 16, H = 32; __VA_ARGS__ } \
    else if (K_ == 4 && H_ == 32) { constexpr int K = 4, H = 32; __VA_ARGS__ } \
    else if (K_ == 16 && H_ == 16) { constexpr int K = 16, H = 16; __VA_ARGS__ } \
    else return nksr_set_error(NKSR_ERR_ARG, "unsupported (kernel_dim, hidden_dim) = (%d, %d)", K_, H_);

