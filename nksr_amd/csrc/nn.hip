// Sparse feature-hierarchy network kernels (network.encoder / network.unet; reference call sites
// models/nksr_net.py:73-78, hyper-parameters configs/default/train.yaml:17-18 "unet.f_maps: 32").
//   k_point_mlp      per-point 2-layer MLP on (local cell coordinate, orientation feature)
//   k_sparse_conv3   3x3x3 submanifold sparse convolution, gather-GEMM on the fp32 matrix cores
//   k_pool_children  mean of the (Morton-contiguous) children of every coarse voxel
//   k_gather_rows    feature transfer between hierarchies / parent -> child up-sampling
//   k_linear         per-voxel linear heads (structure / basis / normal / udf)
// The convolution is the only GEMM-shaped piece of the hot path: one wavefront owns a tile of 32
// output voxels x 32 output channels and issues, per tap, 16 v_mfma_f32_32x32x2_f32 (exact fp32,
// bitwise an fmaf chain) on a 32x32 gathered input tile staged in LDS; the 4 KiB tap weights are
// read straight from L2/L1 (every wavefront reads the same tile).
#include "common.h"

#define NN_C 32

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- point encoder MLP ----------------------------------------------------------------------------------
// in = [u - 0.5 (3), feat (3)] ; h = relu(W1 in + b1) ; out = W2 h + b2      (W1 [C,6], W2 [C,C])
__global__ void __launch_bounds__(256) k_point_mlp(const float* __restrict__ xyz, const float* __restrict__ feat, int64_t n,
                                                   float inv_w0, const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ W2, const float* __restrict__ b2,
                                                   float* __restrict__ out) {
    __shared__ float sW1[NN_C * 6], sb1[NN_C], sW2[NN_C * NN_C], sb2[NN_C];
    for (int i = threadIdx.x; i < NN_C * 6; i += blockDim.x) sW1[i] = W1[i];
    for (int i = threadIdx.x; i < NN_C * NN_C; i += blockDim.x) sW2[i] = W2[i];
    if (threadIdx.x < NN_C) { sb1[threadIdx.x] = b1[threadIdx.x]; sb2[threadIdx.x] = b2[threadIdx.x]; }
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float in[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float p;
        int I = half_index(xyz[i * 3 + a], inv_w0, p) >> 1;
        in[a] = (p - (float)I) - 0.5f;
        in[3 + a] = feat[i * 3 + a];
    }
    float h[NN_C];
#pragma unroll
    for (int c = 0; c < NN_C; ++c) {
        float a = sb1[c];
#pragma unroll
        for (int k = 0; k < 6; ++k) a = fmaf(sW1[c * 6 + k], in[k], a);
        h[c] = a > 0.f ? a : 0.f;
    }
    for (int c = 0; c < NN_C; ++c) {
        float a = sb2[c];
#pragma unroll
        for (int k = 0; k < NN_C; ++k) a = fmaf(sW2[c * NN_C + k], h[k], a);
        out[i * NN_C + c] = a;
    }
}

// ---- trilinear splat-mean of C-channel point features (C <= 32) onto one level ------------------------------
// one thread per (voxel, 8-channel group); gather form as k_splat_trilinear (hierarchy.hip)
__global__ void k_splat_mean_c(const float* __restrict__ xyz, const float* __restrict__ feat, int C,
                               const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                               const int32_t* __restrict__ nbr, const int32_t* __restrict__ ijk, int n, float inv_w,
                               float* __restrict__ out) {
    const int groups = (C + 7) / 8;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * groups) return;
    const int j = (int)(t / groups), g = (int)(t % groups);
    const int c0 = g * 8, nc = (C - c0) < 8 ? (C - c0) : 8;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    float wsum = 0.f;
    const float cx = (float)ijk[j * 3] + 0.5f, cy = (float)ijk[j * 3 + 1] + 0.5f, cz = (float)ijk[j * 3 + 2] + 0.5f;
    for (int s = 0; s < 27; ++s) {
        int c = nbr[(int64_t)j * 27 + s];
        if (c < 0) continue;
        for (int k = start[c]; k < end[c]; ++k) {
            float wx = 1.f - fabsf(__fmul_rn(xyz[(int64_t)k * 3], inv_w) - cx);
            float wy = 1.f - fabsf(__fmul_rn(xyz[(int64_t)k * 3 + 1], inv_w) - cy);
            float wz = 1.f - fabsf(__fmul_rn(xyz[(int64_t)k * 3 + 2], inv_w) - cz);
            if (wx <= 0.f || wy <= 0.f || wz <= 0.f) continue;
            float w = wx * wy * wz;
            wsum += w;
            const float* f = feat + (int64_t)k * C + c0;
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2)
                if (c2 < nc) acc[c2] = fmaf(w, f[c2], acc[c2]);
        }
    }
    const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
    for (int c2 = 0; c2 < nc; ++c2) out[(int64_t)j * C + c0 + c2] = acc[c2] * inv;
}

// ---- 3x3x3 submanifold sparse convolution, C_in = C_out = 32, fp32 MFMA ---------------------------------------
// out[i] = act( b + sum_s W[s]^T in[nbr[i][s]] ),  W [27, Cin, Cout] row-major, optional residual add.
// mfma_f32_32x32x2f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// accumulator register r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
__global__ void __launch_bounds__(256) k_sparse_conv3(const float* __restrict__ in, const int32_t* __restrict__ nbr, int n,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ residual, int relu, float* __restrict__ out) {
    __shared__ float tile[4][32][NN_C + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = (blockIdx.x * 4 + wave) * 32;
    if (base >= n) return;   // whole wavefront; no block-level barrier below
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float(*T)[NN_C + 1] = tile[wave];
    const int col = lane & 31, kh = lane >> 5;
    for (int s = 0; s < 27; ++s) {
        // stage the gathered 32 x 32 input tile: lane -> (row = q*8 + lane/8, 4 channels)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = q * 8 + (lane >> 3), ch = (lane & 7) * 4;
            const int vi = base + row;
            int j = -1;
            if (vi < n) j = nbr[(int64_t)vi * 27 + s];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j >= 0) v = *reinterpret_cast<const float4*>(in + (int64_t)j * NN_C + ch);
            T[row][ch] = v.x; T[row][ch + 1] = v.y; T[row][ch + 2] = v.z; T[row][ch + 3] = v.w;
        }
        const float* Ws = W + (int64_t)s * NN_C * NN_C;
#pragma unroll
        for (int kk = 0; kk < NN_C / 2; ++kk) {
            const int k = kk * 2 + kh;
            const float a = T[col][k];            // A[i = lane & 31][k]
            const float b = Ws[k * NN_C + col];   // B[k][j = lane & 31]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const float bj = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int vi = base + row;
        if (vi < n) {
            float v = acc[r] + bj;
            if (residual) v += residual[(int64_t)vi * NN_C + col];
            if (relu) v = v > 0.f ? v : 0.f;
            out[(int64_t)vi * NN_C + col] = v;
        }
    }
}

// ---- mean over the children of every coarse voxel (children are a contiguous Morton range) ------------------
__global__ void k_pool_children(const float* __restrict__ child_feat, const int32_t* __restrict__ start,
                                const int32_t* __restrict__ end, int n_parent, int C, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_parent * C) return;
    const int p = (int)(t / C), c = (int)(t % C);
    const int k0 = start[p], k1 = end[p];
    float a = 0.f;
    for (int k = k0; k < k1; ++k) a += child_feat[(int64_t)k * C + c];
    out[t] = k1 > k0 ? a / (float)(k1 - k0) : 0.f;
}

// out[i] = (idx[i] >= 0 ? src[idx[i]] : 0) (+ add[i])
__global__ void k_gather_rows(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, int C,
                              const float* __restrict__ add, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    const int64_t i = t / C;
    const int c = (int)(t % C);
    const int j = idx[i];
    float v = j >= 0 ? src[(int64_t)j * C + c] : 0.f;
    if (add) v += add[t];
    out[t] = v;
}

// out[i, o] = b[o] + sum_c W[o, c] in[i, c]   (Cin = 32, Cout <= 32)
__global__ void k_linear(const float* __restrict__ in, int64_t n, const float* __restrict__ W, const float* __restrict__ b,
                         int Cout, float* __restrict__ out) {
    __shared__ float sW[NN_C * NN_C], sb[NN_C];
    for (int i = threadIdx.x; i < Cout * NN_C; i += blockDim.x) sW[i] = W[i];
    if ((int)threadIdx.x < Cout) sb[threadIdx.x] = b ? b[threadIdx.x] : 0.f;
    __syncthreads();
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Cout) return;
    const int64_t i = t / Cout;
    const int o = (int)(t % Cout);
    float a = sb[o];
    const float* x = in + i * NN_C;
#pragma unroll
    for (int c = 0; c < NN_C; ++c) a = fmaf(sW[o * NN_C + c], x[c], a);
    out[t] = a;
}

#define LAUNCH1D(kern, n, stream, ...)                                                              \
    do {                                                                                            \
        if ((n) > 0) {                                                                              \
            hipLaunchKernelGGL(kern, dim3(nksr_blocks((n), 256)), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__); \
            NKSR_CHECK_LAUNCH();                                                                    \
        }                                                                                           \
    } while (0)

extern "C" int nksr_point_mlp(const float* xyz, const float* feat, int64_t n, float inv_w0, int C, const float* W1,
                              const float* b1, const float* W2, const float* b2, float* out, void* stream) {
    if (C != NN_C) return nksr_set_error(NKSR_ERR_ARG, "unet.f_maps must be %d", NN_C);
    LAUNCH1D(k_point_mlp, n, stream, xyz, feat, n, inv_w0, W1, b1, W2, b2, out);
    return NKSR_OK;
}
extern "C" int nksr_splat_mean(const float* xyz_sorted, const float* feat_sorted, int C, const int32_t* start,
                               const int32_t* end, const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w, float* out,
                               void* stream) {
    if (C < 1 || C > 64) return nksr_set_error(NKSR_ERR_ARG, "splat_mean supports 1..64 channels");
    LAUNCH1D(k_splat_mean_c, (int64_t)n * ((C + 7) / 8), stream, xyz_sorted, feat_sorted, C, start, end, nbr, ijk, n, inv_w, out);
    return NKSR_OK;
}
extern "C" int nksr_sparse_conv3(const float* in, const int32_t* nbr, int32_t n, int C, const float* W, const float* bias,
                                 const float* residual, int relu, float* out, void* stream) {
    if (C != NN_C) return nksr_set_error(NKSR_ERR_ARG, "unet.f_maps must be %d", NN_C);
    if (n <= 0) return NKSR_OK;
    hipLaunchKernelGGL(k_sparse_conv3, dim3(nksr_blocks(n, 128)), dim3(256), 0, (hipStream_t)stream, in, nbr, n, W, bias,
                       residual, relu, out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
extern "C" int nksr_pool_children(const float* child_feat, const int32_t* start, const int32_t* end, int32_t n_parent, int C,
                                  float* out, void* stream) {
    LAUNCH1D(k_pool_children, (int64_t)n_parent * C, stream, child_feat, start, end, n_parent, C, out);
    return NKSR_OK;
}
extern "C" int nksr_gather_rows(const float* src, const int32_t* idx, int64_t n, int C, const float* add, float* out,
                                void* stream) {
    LAUNCH1D(k_gather_rows, n * C, stream, src, idx, n, C, add, out);
    return NKSR_OK;
}
extern "C" int nksr_linear(const float* in, int64_t n, int Cin, const float* W, const float* b, int Cout, float* out,
                           void* stream) {
    if (Cin != NN_C || Cout < 1 || Cout > NN_C) return nksr_set_error(NKSR_ERR_ARG, "linear head expects Cin=%d, Cout<=%d", NN_C, NN_C);
    LAUNCH1D(k_linear, n * Cout, stream, in, n, W, b, Cout, out);
    return NKSR_OK;
}
