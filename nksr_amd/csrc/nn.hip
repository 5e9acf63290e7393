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
// The oracle rounds every fp32 product before it is used (numpy).  Device code contracts a * b + c into one fma by default --
// x * inv_w - centre then keeps the unrounded product, the trilinear weights move by an ulp of p and a splat whose normals nearly
// cancel amplifies that to 1e-4 in the unit target (measured in round 3) -- so contraction is off in this file; explicit fmaf stays.
#pragma clang fp contract(off)

#define NN_C 32

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- point encoder MLP ----------------------------------------------------------------------------------
// in = [u - 0.5 (3), feat (3)] ; h = relu(W1 in + b1) ; out = W2 h + b2      (W1 [C,6], W2 [C,C])
__global__ void __launch_bounds__(256) k_point_mlp(const float* __restrict__ xyz, const float* __restrict__ feat, int64_t n,
                                                   float inv_w0, const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ W2, const float* __restrict__ b2,
                                                   float* __restrict__ out) {
    __shared__ float sW1[NN_C * 6], sb1[NN_C], sW2[NN_C * NN_C], sb2[NN_C];
    __shared__ float img[4][64 * 33];
    for (int i = threadIdx.x; i < NN_C * 6; i += blockDim.x) sW1[i] = W1[i];
    for (int i = threadIdx.x; i < NN_C * NN_C; i += blockDim.x) sW2[i] = W2[i];
    if (threadIdx.x < NN_C) { sb1[threadIdx.x] = b1[threadIdx.x]; sb2[threadIdx.x] = b2[threadIdx.x]; }
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row0 = i - (threadIdx.x & 63);            // first row of the wavefront (a multiple of 64)
    if (row0 >= n) return;                                  // (a whole wavefront past the end; a partial one keeps its lanes for the write-out)
    if (i >= n) i = n - 1;                                  // (clamped: computes a row nobody stores)
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
    // the 32 outputs of a lane go through a wave-private LDS image (stride 33 words: conflict-free) and leave as the wavefront's 64
    // rows in one contiguous 8 KB run, 16 bytes per lane and instruction (a lane storing its own row word by word touched 64 lines per
    // store instruction: 1.75 ms per scene step for 1.3 GB)
    float* im = img[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < NN_C; ++c) {
        float a = sb2[c];
#pragma unroll
        for (int k = 0; k < NN_C; ++k) a = fmaf(sW2[c * NN_C + k], h[k], a);
        im[lane * 33 + c] = a;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t left = n - row0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int p = j * 64 + lane, r = p >> 3, q = (p & 7) * 4;
        if (r < left) {
            const float4 v = make_float4(im[r * 33 + q], im[r * 33 + q + 1], im[r * 33 + q + 2], im[r * 33 + q + 3]);
            *reinterpret_cast<float4*>(out + (row0 + r) * NN_C + q) = v;
        }
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

// C = 32 specialisation: one 32-lane half-wave per voxel.  The points that weigh at the voxel are listed first (common.h:
// splat_for_each_point, lane = neighbour cell), then lane = CHANNEL: four points per trip, each feature row one coalesced 128-byte
// read of the half-wave; the sum runs over the points in list order (cell after cell), the same in every lane -- no reduction
// over the lanes, one coalesced 128-byte store per voxel.
__global__ void __launch_bounds__(256) k_splat_mean32(const float* __restrict__ xyz, const float* __restrict__ feat,
                                                      const int32_t* __restrict__ start, const int32_t* __restrict__ end,
                                                      const int32_t* __restrict__ nbr, const int32_t* __restrict__ ijk, int n,
                                                      float inv_w, float* __restrict__ out) {
    __shared__ int lk[8][SPLAT_LIST];
    __shared__ float lw[8][SPLAT_LIST];
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 5, s = threadIdx.x & 31, h = threadIdx.x >> 5;
    const bool live = j < n;                                         // (a dead half-wave walks an empty list next to its partner)
    const int jc = live ? j : n - 1;
    const float cx = (float)ijk[jc * 3] + 0.5f, cy = (float)ijk[jc * 3 + 1] + 0.5f, cz = (float)ijk[jc * 3 + 2] + 0.5f;
    const int c = (live && s < 27) ? nbr[(int64_t)jc * 27 + s] : -1;
    float acc = 0.f, wsum = 0.f;
    splat_for_each_point(xyz, start, end, c, cx, cy, cz, inv_w, s, lk[h], lw[h], [&](const int (&kq)[4], const float (&wq)[4]) {
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = feat[(int64_t)kq[i] * 32 + s];
#pragma unroll
        for (int i = 0; i < 4; ++i) { wsum += wq[i]; acc = fmaf(wq[i], f[i], acc); }
    });
    if (live) out[(int64_t)j * 32 + s] = acc * (wsum > 0.f ? 1.f / wsum : 0.f);
}

// ---- UDF mask branch (NeuralField, models/nksr_net.py:124-130) -------------------------------------------------
// Plane features of one level: per voxel j the trilinear-weighted centroid offset (voxel units, relative
// to the voxel centre) and mean normal of the surrounding points.  out [n, 8] = (occupied, off xyz,
// unit normal xyz, 0).  One thread per voxel, same gather form (and point order) as k_splat_mean_c.
__global__ void k_splat_plane(const float* __restrict__ xyz, const float* __restrict__ nrm, const int32_t* __restrict__ start,
                              const int32_t* __restrict__ end, const int32_t* __restrict__ nbr,
                              const int32_t* __restrict__ ijk, int n, float inv_w, float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float cx = (float)ijk[j * 3] + 0.5f, cy = (float)ijk[j * 3 + 1] + 0.5f, cz = (float)ijk[j * 3 + 2] + 0.5f;
    float wsum = 0.f, ox = 0.f, oy = 0.f, oz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
    for (int s = 0; s < 27; ++s) {
        const int c = nbr[(int64_t)j * 27 + s];
        if (c < 0) continue;
        for (int k = start[c]; k < end[c]; ++k) {
            const float rx = __fmul_rn(xyz[(int64_t)k * 3], inv_w) - cx, ry = __fmul_rn(xyz[(int64_t)k * 3 + 1], inv_w) - cy,
                        rz = __fmul_rn(xyz[(int64_t)k * 3 + 2], inv_w) - cz;
            const float wx = 1.f - fabsf(rx), wy = 1.f - fabsf(ry), wz = 1.f - fabsf(rz);
            if (wx <= 0.f || wy <= 0.f || wz <= 0.f) continue;
            const float w = wx * wy * wz;
            wsum += w;
            ox = fmaf(w, rx, ox); oy = fmaf(w, ry, oy); oz = fmaf(w, rz, oz);
            nx = fmaf(w, nrm[(int64_t)k * 3], nx); ny = fmaf(w, nrm[(int64_t)k * 3 + 1], ny); nz = fmaf(w, nrm[(int64_t)k * 3 + 2], nz);
        }
    }
    float* o = out + (int64_t)j * 8;
    const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
    const float nn = sqrtf(nx * nx + ny * ny + nz * nz);
    const float invn = nn > 1e-8f ? 1.f / nn : 0.f;
    o[0] = wsum > 0.f ? 1.f : 0.f;
    o[1] = ox * inv; o[2] = oy * inv; o[3] = oz * inv;
    o[4] = nx * invn; o[5] = ny * invn; o[6] = nz * invn;
    o[7] = 0.f;
}

// udf(x) = w_d | sum_c t_c(x) <x/w_d - centre_c - off_c, n_c> | / sum_c t_c(x) over the occupied voxels among
// the 8 centres surrounding x (t = trilinear weight); NKSR_UDF_FAR where none is occupied.  With
// only_unset, queries that already hold a value from a finer level are left alone.
#define NKSR_UDF_FAR 1e30f
__global__ void k_udf_decode(nksr_level_t lv, const float* __restrict__ feat, const float* __restrict__ xyz, int64_t nq,
                             float inv_w, float w, int level, int only_unset, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    if (only_unset && out[i] < 0.5f * NKSR_UDF_FAR) return;
    const float px = __fmul_rn(xyz[i * 3], inv_w), py = __fmul_rn(xyz[i * 3 + 1], inv_w), pz = __fmul_rn(xyz[i * 3 + 2], inv_w);
    const float fx = floorf(px - 0.5f), fy = floorf(py - 0.5f), fz = floorf(pz - 0.5f);
    const int bx = (int)fx, by = (int)fy, bz = (int)fz;
    const float vx = px - 0.5f - fx, vy = py - 0.5f - fy, vz = pz - 0.5f - fz;
    float sw = 0.f, sd = 0.f;
    for (int c = 0; c < 8; ++c) {
        const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
        const int j = hash_find(lv.hkeys, lv.hvals, lv.hcap, morton_biased(bx + cx, by + cy, bz + cz, NKSR_BIAS0 >> level));
        if (j < 0) continue;
        const float* f = feat + (int64_t)j * 8;
        if (!(f[0] > 0.5f)) continue;
        const float t = (cx ? vx : 1.f - vx) * (cy ? vy : 1.f - vy) * (cz ? vz : 1.f - vz);
        const float rx = px - ((float)(bx + cx) + 0.5f) - f[1], ry = py - ((float)(by + cy) + 0.5f) - f[2],
                    rz = pz - ((float)(bz + cz) + 0.5f) - f[3];
        const float d = fmaf(rx, f[4], fmaf(ry, f[5], rz * f[6]));
        sw += t;
        sd = fmaf(t, d, sd);
    }
    if (sw > 0.f) out[i] = fabsf(sd / sw) * w;
    else if (!only_unset) out[i] = NKSR_UDF_FAR;
}

__global__ void k_fill_f32(float* __restrict__ p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- 3x3x3 submanifold sparse convolution, C_in = C_out = 32, fp32 MFMA ---------------------------------------
// out[i] = act( b + sum_s W[s]^T in[nbr[i][s]] ),  W [27, Cin, Cout] row-major, optional residual add.
// mfma_f32_32x32x2f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// accumulator register r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].
//
// Round 5: the A operand comes STRAIGHT from the gathered rows.  The two lanes of a voxel (l and l + 32) split its 32 input
// channels in HALVES -- lane (i, h) holds channels 16 h .. 16 h + 15 of in[nbr[i][s]] (four 16-byte loads of ITS OWN row) and
// feeds channel 16 h + kk to product kk, next to B = W[s][16 h + kk][col]: a tap sums its channels in the order 0, 16, 1, 17, ...
// (a fixed order; the sum over k of a matrix instruction has no preferred one).  Rounds 2-4 dealt the channels even / odd, which
// needed every row transposed through LDS first: 16 ds_write + 16 ds_read per tap and their waits between the loads and the
// products -- with 112 VGPRs (4 waves per SIMD, 2.5 resident on average by the counters) the matrix pipe sat at 54 % of its
// fp32 rate (`profiles/r05_scene_fused_pmc.json`: SQ_VALU_MFMA_BUSY_CYCLES).  Dropping that staging alone changed nothing (24.5 ms
// per scene step against 23.4): the vector-memory data path was the busy unit (TD_TD_BUSY 86 %) -- see the weight tile below.
// The taps are still software-pipelined: while the 16 products of tap s run, the rows and weights of tap s + 1 and the neighbour indices of tap
// s + 2 are in flight (all loads unconditional -- clamped index, masked value).
__global__ void __launch_bounds__(256) k_sparse_conv3(const float* __restrict__ in, const int32_t* __restrict__ nbr, int n,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      const float* __restrict__ residual, int relu, float* __restrict__ out) {
    // the weights of a tap (32 x 32 floats) are shared by the four wavefronts of the workgroup: ONE 16-byte load per thread and tap
    // into a double-buffered LDS tile instead of 16 four-byte loads per lane (the vector-memory data path was 86 % busy,
    // TD_TD_BUSY, with the matrix pipe at 54 %); one barrier per tap
    __shared__ float4 wt[2][NN_C * NN_C / 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = (blockIdx.x * 4 + wave) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int col = lane & 31, kh = lane >> 5;
    const int vi = base + col;
    const bool live = vi < n;                                       // (a wavefront past the end keeps step with the barriers)
    const int64_t nrow = (int64_t)(live ? vi : n - 1) * 27;
    const float4* W4 = reinterpret_cast<const float4*>(W);
    int jc = nbr[nrow], jn = nbr[nrow + 1];
    float4 vc[4], vn[4];
    {
        const float4* row = reinterpret_cast<const float4*>(in + (int64_t)(live && jc >= 0 ? jc : 0) * NN_C + 16 * kh);
#pragma unroll
        for (int q = 0; q < 4; ++q) vc[q] = row[q];
    }
    wt[0][threadIdx.x] = W4[threadIdx.x];
    __syncthreads();
    for (int s = 0; s < 27; ++s) {
        // requests for the taps to come (the last ones repeat tap 26: harmless)
        const int s1 = s + 1 < 27 ? s + 1 : 26, s2 = s + 2 < 27 ? s + 2 : 26;
        const float4 wnext = W4[(int64_t)s1 * (NN_C * NN_C / 4) + threadIdx.x];
        {
            const float4* row = reinterpret_cast<const float4*>(in + (int64_t)(live && jn >= 0 ? jn : 0) * NN_C + 16 * kh);
#pragma unroll
            for (int q = 0; q < 4; ++q) vn[q] = row[q];
        }
        const int j2 = nbr[nrow + s2];
        const float* wl = reinterpret_cast<const float*>(wt[s & 1]) + (16 * kh) * NN_C + col;      // W[s][16 kh + kk][col]
        float bc[NN_C / 2];
#pragma unroll
        for (int kk = 0; kk < NN_C / 2; ++kk) bc[kk] = wl[kk * NN_C];
        const bool ok = live && jc >= 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? vc[q].x : 0.f, bc[4 * q], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? vc[q].y : 0.f, bc[4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? vc[q].z : 0.f, bc[4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? vc[q].w : 0.f, bc[4 * q + 3], acc, 0, 0, 0);
        }
        jc = jn; jn = j2;
#pragma unroll
        for (int q = 0; q < 4; ++q) vc[q] = vn[q];
        wt[(s + 1) & 1][threadIdx.x] = wnext;          // (the tile of tap s - 1: every wavefront finished reading it before the last barrier)
        __syncthreads();
    }
    const float bj = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int vo = base + row;
        if (vo < n) {
            float v = acc[r] + bj;
            if (residual) v += residual[(int64_t)vo * NN_C + col];
            if (relu) v = v > 0.f ? v : 0.f;
            out[(int64_t)vo * NN_C + col] = v;
        }
    }
}

// ---- weight gradient of the convolution (training path, nn/backward.py) -------------------------------------------------------------
// gW[s][ci][co] = sum_i in[nbr[i][s]][ci] * gz[i][co]: per tap a [32 x n] x [n x 32] product, the reduction runs over the voxels.  One
// wavefront per (chunk of CW_CHUNK voxels, tap): v_mfma_f32_32x32x2_f32 takes TWO voxels per instruction -- lane l supplies
// A[ci = l & 31][k = l >> 5] = the gathered input row of voxel k of the pair and B[k][co = l & 31] = its output-gradient row, both
// coalesced 128-byte reads.  Every wavefront writes its own 32 x 32 partial (no atomics: deterministic); the host adds the chunks.
#define CW_CHUNK 2048
__global__ void __launch_bounds__(64) k_conv3_wgrad(const float* __restrict__ in, const int32_t* __restrict__ nbr, int n, const float* __restrict__ gz,
                                                    float* __restrict__ partial) {
    const int chunk = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const int ci = lane & 31, k = lane >> 5;
    const int i0 = chunk * CW_CHUNK, i1 = (i0 + CW_CHUNK < n) ? i0 + CW_CHUNK : n;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i = i0; i < i1; i += 8) {                   // four pairs per trip: their loads go out together
        float a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = i + 2 * q + k;
            const bool live = v < i1;
            const int j = live ? nbr[(int64_t)v * 27 + s] : -1;
            a[q] = j >= 0 ? in[(int64_t)j * NN_C + ci] : 0.f;
            b[q] = live ? gz[(int64_t)v * NN_C + ci] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
    }
    float* out = partial + ((int64_t)chunk * 27 + s) * NN_C * NN_C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * k;
        out[row * NN_C + ci] = acc[r];
    }
}
extern "C" int64_t nksr_conv3_wgrad_chunks(int32_t n) { return n > 0 ? (int64_t)nksr_blocks(n, CW_CHUNK) : 0; }
extern "C" int nksr_conv3_wgrad(const float* in, const int32_t* nbr, int32_t n, int C, const float* gz, float* partial, void* stream) {
    if (C != NN_C) return nksr_set_error(NKSR_ERR_ARG, "unet.f_maps must be %d", NN_C);
    if (n <= 0) return NKSR_OK;
    if (!in || !nbr || !gz || !partial) return nksr_set_error(NKSR_ERR_ARG, "conv3 wgrad: NULL arrays");
    hipLaunchKernelGGL(k_conv3_wgrad, dim3(nksr_blocks(n, CW_CHUNK), 27), dim3(64), 0, (hipStream_t)stream, in, nbr, n, gz, partial);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
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

// the same for rows of 32 floats on 16-byte aligned arrays: eight lanes per row, 16 bytes each (the generic kernel spends a 64-bit
// division per WORD: 4.6 ms per scene step at 1.9 TB/s)
__global__ void __launch_bounds__(256) k_gather_rows32(const float4* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                                                       const float4* __restrict__ add, float4* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * 8) return;
    const int j = idx[t >> 3];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j >= 0) v = src[(int64_t)j * 8 + (t & 7)];
    if (add) {
        const float4 a = add[t];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
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
    if ((uintptr_t)out & 15) return nksr_set_error(NKSR_ERR_ARG, "point_mlp: out must be 16-byte aligned (rows leave as 16-byte pieces)");
    LAUNCH1D(k_point_mlp, n, stream, xyz, feat, n, inv_w0, W1, b1, W2, b2, out);
    return NKSR_OK;
}
extern "C" int nksr_splat_mean(const float* xyz_sorted, const float* feat_sorted, int C, const int32_t* start,
                               const int32_t* end, const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w, float* out,
                               void* stream) {
    if (C < 1 || C > 64) return nksr_set_error(NKSR_ERR_ARG, "splat_mean supports 1..64 channels");
    if (C == 32 && n > 0) {
        hipLaunchKernelGGL(k_splat_mean32, dim3(nksr_blocks((int64_t)n * 32, 256)), dim3(256), 0, (hipStream_t)stream, xyz_sorted, feat_sorted, start, end,
                           nbr, ijk, n, inv_w, out);
        NKSR_CHECK_LAUNCH();
        return NKSR_OK;
    }
    LAUNCH1D(k_splat_mean_c, (int64_t)n * ((C + 7) / 8), stream, xyz_sorted, feat_sorted, C, start, end, nbr, ijk, n, inv_w, out);
    return NKSR_OK;
}
extern "C" int nksr_sparse_conv3(const float* in, const int32_t* nbr, int32_t n, int C, const float* W, const float* bias,
                                 const float* residual, int relu, float* out, void* stream) {
    if (C != NN_C) return nksr_set_error(NKSR_ERR_ARG, "unet.f_maps must be %d", NN_C);
    if (n <= 0) return NKSR_OK;
    if (((uintptr_t)in | (uintptr_t)W) & 15) return nksr_set_error(NKSR_ERR_ARG, "sparse_conv3: in and W must be 16-byte aligned (16-byte row / weight loads)");
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
    if (C == NN_C && !(((uintptr_t)src | (uintptr_t)add | (uintptr_t)out) & 15)) {
        if (n <= 0) return NKSR_OK;
        hipLaunchKernelGGL(k_gather_rows32, dim3(nksr_blocks(n * 8, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src), idx, n,
                           reinterpret_cast<const float4*>(add), reinterpret_cast<float4*>(out));
        NKSR_CHECK_LAUNCH();
        return NKSR_OK;
    }
    LAUNCH1D(k_gather_rows, n * C, stream, src, idx, n, C, add, out);
    return NKSR_OK;
}
extern "C" int nksr_linear(const float* in, int64_t n, int Cin, const float* W, const float* b, int Cout, float* out,
                           void* stream) {
    if (Cin != NN_C || Cout < 1 || Cout > NN_C) return nksr_set_error(NKSR_ERR_ARG, "linear head expects Cin=%d, Cout<=%d", NN_C, NN_C);
    LAUNCH1D(k_linear, n * Cout, stream, in, n, W, b, Cout, out);
    return NKSR_OK;
}

extern "C" int nksr_splat_plane(const float* xyz_sorted, const float* normal_sorted, const int32_t* start, const int32_t* end,
                                const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w, float* out, void* stream) {
    LAUNCH1D(k_splat_plane, n, stream, xyz_sorted, normal_sorted, start, end, nbr, ijk, n, inv_w, out);
    return NKSR_OK;
}
extern "C" int nksr_udf_decode(const nksr_level_t* level, int level_index, const float* feat, const float* xyz, int64_t n,
                               float inv_w, float voxel_size, int only_unset, float* out, void* stream) {
    if (!level) return nksr_set_error(NKSR_ERR_ARG, "null level");
    if (level->n <= 0) {
        if (!only_unset && n > 0) { LAUNCH1D(k_fill_f32, n, stream, out, n, NKSR_UDF_FAR); }
        return NKSR_OK;
    }
    LAUNCH1D(k_udf_decode, n, stream, *level, feat, xyz, n, inv_w, voxel_size, level_index, only_unset, out);
    return NKSR_OK;
}
