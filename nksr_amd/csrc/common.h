// Shared host/device helpers for the gfx950 kernels.  Integer conventions are the ones
// fixed in DESIGN.md section 2 (and restated independently in oracle/spec.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nksr_hip.h"

#define NKSR_WAVE 64
#define NKSR_BIAS0 (1 << 20)

extern thread_local char g_nksr_err[512];
int nksr_set_error(int code, const char* fmt, ...);

#define NKSR_CHECK_HIP(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return nksr_set_error(NKSR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,    \
                                  hipGetErrorString(_e), __FILE__, __LINE__);       \
    } while (0)

#define NKSR_CHECK_LAUNCH() NKSR_CHECK_HIP(hipGetLastError())

static inline int nksr_blocks(int64_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

// ---- Morton codes (x = lowest bit, 21 bits per axis) -------------------------------------
__host__ __device__ __forceinline__ uint64_t part1by2(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__host__ __device__ __forceinline__ uint64_t compact1by2(uint64_t v) {
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10C30C30C30C30C3ull;
    v = (v | (v >> 4)) & 0x100F00F00F00F00Full;
    v = (v | (v >> 8)) & 0x1F0000FF0000FFull;
    v = (v | (v >> 16)) & 0x1F00000000FFFFull;
    v = (v | (v >> 32)) & 0x1FFFFFull;
    return v;
}
// key of integer coordinates biased by `bias` (level d: NKSR_BIAS0 >> d; lattice: NKSR_BIAS0)
__host__ __device__ __forceinline__ int64_t morton_biased(int x, int y, int z, int bias) {
    return (int64_t)(part1by2((uint64_t)(x + bias)) | (part1by2((uint64_t)(y + bias)) << 1) |
                     (part1by2((uint64_t)(z + bias)) << 2));
}
__host__ __device__ __forceinline__ void morton_decode_biased(int64_t key, int bias, int& x, int& y, int& z) {
    x = (int)compact1by2((uint64_t)key) - bias;
    y = (int)compact1by2((uint64_t)key >> 1) - bias;
    z = (int)compact1by2((uint64_t)key >> 2) - bias;
}

// ---- open-addressing hash (linear probing, 64-bit keys, -1 = empty) ----------------------
__device__ __forceinline__ uint32_t hash_mix(int64_t k) {
    uint64_t h = (uint64_t)k;
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return (uint32_t)h;
}
// The eight children of a parent cell (keys that agree above their low three bits) share one 64-byte line of the table: the 27
// neighbours of a voxel then touch ~8 lines instead of 27, and Morton-sorted inserts / queries walk the table coherently (the
// neighbour tables of a 17 M-voxel level were bound by the 64-byte sectors a fully random slot fetches).  A collision moves to
// ANOTHER line (triangular steps over the lines, same place within the line): walking on within the line would run through the
// siblings, and a miss -- half the 27 neighbours of a surface voxel -- would pay for the whole cluster.
__device__ __forceinline__ uint32_t hash_slot(int64_t key, int hcap) {
    return ((hash_mix(key >> 3) << 3) | ((uint32_t)key & 7u)) & (uint32_t)(hcap - 1);
}
// probe = 1, 2, ...: the step that follows try number probe.  Triangular steps visit every line once per hcap / 8 tries; a key set
// whose low three bits are all alike (it can only use one place of every line) then moves on to the next place: hcap tries see
// every slot of the table.
__device__ __forceinline__ uint32_t hash_next(uint32_t slot, int probe, int hcap) {
    slot = (slot + ((uint32_t)probe << 3)) & (uint32_t)(hcap - 1);
    if (((uint32_t)probe & (uint32_t)((hcap >> 3) - 1)) == 0u) slot = (slot & ~7u) | ((slot + 1u) & 7u);
    return slot;
}
__device__ __forceinline__ int hash_find(const int64_t* __restrict__ hkeys, const int32_t* __restrict__ hvals,
                                         int hcap, int64_t key) {
    uint32_t slot = hash_slot(key, hcap);
    for (int probe = 1; probe <= hcap; ++probe) {
        int64_t k = hkeys[slot];
        if (k == key) return hvals[slot];
        if (k == -1) return -1;
        slot = hash_next(slot, probe, hcap);
    }
    return -1;
}

// ---- integer cell decisions from ONE fp32 product (oracle/spec.py) -------------------------
__device__ __forceinline__ int half_index(float x, float inv_w0, float& p) {
    p = __fmul_rn(x, inv_w0);
    return (int)floorf(__fmul_rn(p, 2.0f));
}

// quadratic B-spline weights of the centres at offset -1,0,+1 for local coordinate u
// (contraction off: left to the compiler, 0.75 - uc * uc became a fused multiply-add for one axis and a product + subtraction for
// another inside the same kernel; every kernel that forms these weights -- one lane per site or one lane per slot -- must round alike)
__device__ __forceinline__ void bspline3(float u, float w[3], float dw[3]) {
#pragma clang fp contract(off)
    float um = 1.0f - u, uc = u - 0.5f;
    w[0] = 0.5f * um * um;
    w[1] = 0.75f - uc * uc;
    w[2] = 0.5f * u * u;
    dw[0] = u - 1.0f;
    dw[1] = -2.0f * uc;
    dw[2] = u;
}

// ---- the points that weigh at a voxel, in a fixed order (trilinear splats: hierarchy.hip, nn.hip) --------------------------------
// One 32-lane half-wave per voxel, lane s < 27 = neighbour cell c (-1: none).  Every lane walks the points of ITS cell and keeps
// those with a positive trilinear weight at the voxel centre (cx, cy, cz; voxel units), at most SPLAT_CAP per round; the kept
// (point, weight) pairs of the 27 lanes are compacted -- cell after cell, points in their order -- into a list in LDS, and
// `consume(k[4], w[4])` gets them four at a time (missing ones: weight 0, point 0), so that the caller's loads of four points go
// out together and one row of a point is read by the whole half-wave at once.  (Round 3: every lane fetched the feature rows of
// its own cell's points inside the branchy walk -- 8 x 16 bytes per lane and point, one dependent round trip per point.)
// lk / lw: SPLAT_LIST ints / floats of LDS owned by this half-wave.  Rounds repeat until every cell is exhausted.
#define SPLAT_CAP 4
#define SPLAT_LIST 128
typedef float splat_f32x3 __attribute__((ext_vector_type(3), aligned(4)));
template <typename F>
__device__ __forceinline__ void splat_for_each_point(const float* __restrict__ xyz, const int32_t* __restrict__ start,
                                                     const int32_t* __restrict__ end, int c, float cx, float cy, float cz, float inv_w,
                                                     int s, int* lk, float* lw, F&& consume) {
#pragma clang fp contract(off)          // the weights are the oracle's: x * inv_w rounded before the subtraction
    int k = c >= 0 ? start[c] : 0;
    const int k1 = c >= 0 ? end[c] : 0;
    while (true) {
        // SPLAT_CAP candidates per round, their coordinates requested together (clamped index: point 0 exists whenever a cell does)
        splat_f32x3 pt[SPLAT_CAP];
#pragma unroll
        for (int i = 0; i < SPLAT_CAP; ++i) pt[i] = *reinterpret_cast<const splat_f32x3*>(xyz + (int64_t)(k + i < k1 ? k + i : 0) * 3);
        float wv[SPLAT_CAP];
        int rank[SPLAT_CAP], cnt = 0;
#pragma unroll
        for (int i = 0; i < SPLAT_CAP; ++i) {
            const float wx = 1.f - fabsf(pt[i].x * inv_w - cx), wy = 1.f - fabsf(pt[i].y * inv_w - cy), wz = 1.f - fabsf(pt[i].z * inv_w - cz);
            const bool keep = k + i < k1 && wx > 0.f && wy > 0.f && wz > 0.f;
            wv[i] = keep ? wx * wy * wz : 0.f;
            rank[i] = keep ? cnt : -1;
            cnt += keep ? 1 : 0;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int t = __shfl_up(incl, o, 32);
            if (s >= o) incl += t;
        }
        const int off = incl - cnt, P = __shfl(incl, 31, 32);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the reads of the previous round are done
#pragma unroll
        for (int i = 0; i < SPLAT_CAP; ++i)
            if (rank[i] >= 0) { lk[off + rank[i]] = k + i; lw[off + rank[i]] = wv[i]; }
        k = k + SPLAT_CAP < k1 ? k + SPLAT_CAP : k1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int Po = __shfl_xor(P, 32, 64), Pw = P > Po ? P : Po;     // the two halves of the wavefront loop together
        for (int p = 0; p < Pw; p += 4) {
            int kq[4];
            float wq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = p + i < P;
                const int kv = lk[(p + i) & (SPLAT_LIST - 1)];
                const float wv = lw[(p + i) & (SPLAT_LIST - 1)];
                kq[i] = ok ? kv : 0;
                wq[i] = ok ? wv : 0.f;
            }
            consume(kq, wq);
        }
        if (__ballot(k < k1) == 0ull) break;
    }
}

// ---- reductions over the 32 lanes of a half-wavefront (fixed trees: deterministic) ------------------------------------------------
__device__ __forceinline__ float half_sum(float p) {      // sum over the 32 lanes of this half-wave, fixed tree
    p += __shfl_xor(p, 16, 32);
    p += __shfl_xor(p, 8, 32);
    p += __shfl_xor(p, 4, 32);
    p += __shfl_xor(p, 2, 32);
    p += __shfl_xor(p, 1, 32);
    return p;
}

// sums of FOUR rows over the 32 lanes of a half-wave in 6 lane exchanges instead of 4 x 5: a transposing butterfly -- after the
// xor-16 step a lane keeps two of the four rows (its own + its partner's share), after xor-8 one, then 3 plain steps.
// Returns, in every lane, the total of row  2 * bit4(lane) + bit3(lane).
// lane exchanges inside a 16-lane row as DPP modifiers of a VALU move (no LDS crossbar): quad_perm [1,0,3,2] (xor 1),
// [2,3,0,1] (xor 2), row_half_mirror (l -> 7 - l: the other quad of an 8-lane group once quads are uniform), row_ror:8 (xor 8)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_sum4(float p0, float p1, float p2, float p3, int lane) {
    const bool hi = (lane >> 4) & 1, b = (lane >> 3) & 1;
    const float r0 = __shfl_xor(hi ? p0 : p2, 16, 32), r1 = __shfl_xor(hi ? p1 : p3, 16, 32);       // the only two crossbar trips
    const float q0 = (hi ? p2 : p0) + r0, q1 = (hi ? p3 : p1) + r1;          // rows {2,3} in the upper 16 lanes, {0,1} in the lower
    float r = (b ? q1 : q0) + dpp_move<0x128>(b ? q0 : q1);                  // row_ror:8
    r += dpp_move<0xB1>(r);                                                  // xor 1
    r += dpp_move<0x4E>(r);                                                  // xor 2
    r += dpp_move<0x141>(r);                                                 // row_half_mirror: the other quad
    return r;
}

// two rows: after the xor-16 step the lower 16 lanes hold row 0, the upper 16 row 1.  Returns the total of row bit4(lane).
__device__ __forceinline__ float half_sum2(float p0, float p1, int lane) {
    const bool hi = (lane >> 4) & 1;
    float r = (hi ? p1 : p0) + __shfl_xor(hi ? p0 : p1, 16, 32);
    r += dpp_move<0x128>(r);
    r += dpp_move<0xB1>(r);
    r += dpp_move<0x4E>(r);
    r += dpp_move<0x141>(r);
    return r;
}

