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
__device__ __forceinline__ int hash_find(const int64_t* __restrict__ hkeys, const int32_t* __restrict__ hvals,
                                         int hcap, int64_t key) {
    uint32_t slot = hash_mix(key) & (uint32_t)(hcap - 1);
    for (int probe = 0; probe < hcap; ++probe) {
        int64_t k = hkeys[slot];
        if (k == key) return hvals[slot];
        if (k == -1) return -1;
        slot = (slot + 1) & (uint32_t)(hcap - 1);
    }
    return -1;
}

// ---- integer cell decisions from ONE fp32 product (oracle/spec.py) -------------------------
__device__ __forceinline__ int half_index(float x, float inv_w0, float& p) {
    p = __fmul_rn(x, inv_w0);
    return (int)floorf(__fmul_rn(p, 2.0f));
}

// quadratic B-spline weights of the centres at offset -1,0,+1 for local coordinate u
__device__ __forceinline__ void bspline3(float u, float w[3], float dw[3]) {
    float um = 1.0f - u, uc = u - 0.5f;
    w[0] = 0.5f * um * um;
    w[1] = 0.75f - uc * uc;
    w[2] = 0.5f * u * u;
    dw[0] = u - 1.0f;
    dw[1] = -2.0f * uc;
    dw[2] = u;
}
