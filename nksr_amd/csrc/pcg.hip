// Jacobi-preconditioned conjugate gradients on the CSR normal equations.
// The CSR SpMV is the roofline kernel of this project:
//   algorithmic bytes per launch  B_spmv = 8*nnz + 12*M + 4   (fp32 vals, int32 cols/rowptr)
// Per iteration: (1) y = A p, (2) partial dot p.y (own small kernel, see k_pcg_dot), (3) x,r,z update
// fused with the partial dots r.r and r.z, (4) p update.  alpha/beta never leave the device; dot products are
// accumulated in fp64 with a fixed reduction order (deterministic).  A device-side `done` flag
// turns the remaining launches of a chunk into no-ops, so the host only syncs every
// `check_every` iterations.
#include "common.h"
#include "pcg_core.h"

#define PCG_BLOCK 256
#define PCG_MAX_BLOCKS 2048
#define SPCG_BS 1024            // unknowns per reduction block of the segmented PCG (256 threads x 4)

// Independent diagonal blocks ("segments": the chunks of a batched chunk solve, nksr_segments_t) run their OWN conjugate gradients
// inside shared launches: scalars per segment, dot products reduced per segment in a fixed, SEGMENT-RELATIVE order (blocks of
// SPCG_BS unknowns counted from the start of each of the segment's index ranges, partial sums added in block order), so the
// iterates of a segment do not depend on which other segments share the launch; a converged segment freezes (its blocks return
// early) while the others go on.  One segment [0, M) is the ordinary solve.
struct SegScalars {
    double rz[2];
    double bb;
    double rel;
    int iter;
    int done;               // 1: converged (or empty right-hand side), 2: hard failure (r.z <= 0 with the Jacobi preconditioner too, NaN)
    int jacobi;             // 1: the coarse-level block lost definiteness for this segment (r.z <= 0): it went on with Jacobi alone
    int pad;
};
struct PcgGlobal {          // what the host reads back every check_every iterations
    double max_rel;
    int max_iter;
    int done_all;
    int done_count;
    int nblocks;            // reduction blocks in use (k_seg_fill)
    int min_iter;           // fewest iterations any segment has taken so far (= all segments were still iterating up to here)
    int fallbacks;          // segments that dropped the coarse-level block and went on with Jacobi
    int pad;
};

struct PcgWork {
    float *r, *z, *p, *y;
    double* part1;          // [nb_max]
    double* part2;          // [3 * nb_max]: r.r, r.z with the preconditioner in use, r.z with plain Jacobi (the fallback's)
    int32_t *blk_lo, *blk_hi, *blk_seg;   // [nb_max] reduction blocks: unknown range + segment
    int32_t* seg_blk;       // [nseg + 1] first block of every segment
    SegScalars* sc;         // [nseg]
    PcgGlobal* g;
    int nseg, nranges, nb_max;
    const int32_t *lo, *hi; // [nseg * nranges] (device) or NULL: one segment [0, M)
    int M;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int spcg_nb_max(int32_t M, int nseg, int nranges) { return (M + SPCG_BS - 1) / SPCG_BS + nseg * nranges; }

static size_t pcg_vector_bytes_seg(int32_t M, int nseg, int nranges) {
    const size_t vec = align_up((size_t)M * sizeof(float), 256), nb = (size_t)spcg_nb_max(M, nseg, nranges);
    return 4 * vec + align_up(4 * nb * sizeof(double), 256) + align_up(3 * nb * sizeof(int32_t), 256) + align_up(((size_t)nseg + 1) * sizeof(int32_t), 256) +
           align_up((size_t)nseg * sizeof(SegScalars), 256) + 256;
}
static size_t pcg_vector_bytes(int32_t M) { return pcg_vector_bytes_seg(M, 1, 1); }
size_t nksr_pcg_vector_bytes(int32_t M) { return pcg_vector_bytes(M); }
extern "C" size_t nksr_pcg_vector_workspace_bytes_seg(int32_t M, int32_t nseg, int32_t nranges) {
    return pcg_vector_bytes_seg(M, nseg < 1 ? 1 : nseg, nranges < 1 ? 1 : nranges);
}
extern "C" size_t nksr_spmv_workspace_bytes(int64_t nnz);
extern "C" size_t nksr_pcg_workspace_bytes(int32_t M, int64_t nnz) {
    return pcg_vector_bytes(M) + nksr_spmv_workspace_bytes(nnz);
}

static PcgWork carve(void* ws, int M, const nksr_segments_t* seg) {
    PcgWork w;
    w.nseg = seg ? seg->nseg : 1;
    w.nranges = seg ? seg->nranges : 1;
    w.lo = seg ? seg->lo : nullptr;
    w.hi = seg ? seg->hi : nullptr;
    w.M = M;
    w.nb_max = spcg_nb_max(M, w.nseg, w.nranges);
    char* p = (char*)ws;
    const size_t vec = align_up((size_t)M * sizeof(float), 256), nb = (size_t)w.nb_max;
    w.r = (float*)p; p += vec;
    w.z = (float*)p; p += vec;
    w.p = (float*)p; p += vec;
    w.y = (float*)p; p += vec;
    w.part1 = (double*)p;
    w.part2 = w.part1 + nb; p += align_up(4 * nb * sizeof(double), 256);
    w.blk_lo = (int32_t*)p;
    w.blk_hi = w.blk_lo + nb;
    w.blk_seg = w.blk_hi + nb; p += align_up(3 * nb * sizeof(int32_t), 256);
    w.seg_blk = (int32_t*)p; p += align_up(((size_t)w.nseg + 1) * sizeof(int32_t), 256);
    w.sc = (SegScalars*)p; p += align_up((size_t)w.nseg * sizeof(SegScalars), 256);
    w.g = (PcgGlobal*)p;
    return w;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// block-wide fp64 sum; result valid in thread 0.  sm must hold blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    return t;
}

// every block re-reduces the partial array in the same fixed order -> identical value everywhere
__device__ __forceinline__ double reduce_partials(const double* __restrict__ part, int nb, int stride, double* sm) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) v += part[(int64_t)i * stride];
    double t = block_sum(v, sm);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    return bc;
}

// ---- SpMV ---------------------------------------------------------------------------------------
// nnz-balanced streaming CSR SpMV.  The (col,val) stream is cut into chunks of CHUNK entries
// regardless of row boundaries (rows range from ~30 to several thousand entries: a coarse voxel
// couples to every fine voxel under its support), so every workgroup moves the same number of
// bytes with perfectly coalesced wide loads, ~32 KiB in flight per workgroup, ~8 workgroups per CU.
// Products go to LDS; each wavefront then reduces whole row segments from LDS (butterfly), writing y
// for rows that START in the chunk and one carry per chunk for the row that started earlier.  A tiny
// fix-up kernel adds the carries in chunk order, so the result is deterministic.  x gathers hit L2
// (Morton-ordered unknowns).
//
// Two physical layouts (nksr_hip.h, col_format), both interleaved so that component j of the load of
// lane l is logical entry 64 j + l of its tile -- every gather instruction covers 64 CONSECUTIVE
// entries of the stream:
//   format 0: EPL = 4 entries per lane, 256-entry tiles, int32 columns (16 + 16 bytes per lane)
//   format 1: EPL = 3 entries per lane, 192-entry tiles, three 21-bit columns packed in one 64-bit
//             word (8 + 12 bytes per lane): 6.67 instead of 8 bytes per entry, M <= 2^21.
// Storage is zero-padded (column 0, value 0) to a multiple of the chunk size: no bounds checks.
// The (col, val) stream is read with the non-temporal hint (round 3): it is touched once, and without the hint it pushed x out of the
// 4 MB L2 of every XCD (x is 4.5 MB at the bench workload): 589 -> 559 us per application, 0.639 -> 0.673 of 8 TB/s on the same box
// (VARIANT 2 = plain loads, kept for the comparison; VARIANT 1 = no gather: 465 us = what the stream alone allows).
template <int EPL> struct SpmvFmt {
    static constexpr int TILE = 64 * EPL;
    static constexpr int QUAD = PCG_BLOCK * EPL;          // entries per workgroup-wide load
    static constexpr int QUADS = EPL == 4 ? 4 : 6;
    static constexpr int CHUNK = QUADS * QUAD;            // 4096 / 4608
};
#define SPMV_CHUNK_MIN 4096
static int spmv_chunk(int fmt) { return fmt == 1 ? SpmvFmt<3>::CHUNK : SpmvFmt<4>::CHUNK; }

__global__ void k_spmv_plan(const int32_t* __restrict__ rowptr, int M, int64_t nnz, int nchunks, int chunk,
                            int32_t* __restrict__ chunk_row) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nchunks) return;
    if (b == nchunks) { chunk_row[b] = M; return; }
    // row containing entry b*CHUNK: last r with rowptr[r] <= k
    const int64_t k = (int64_t)b * chunk;
    int lo = 0, hi = M;  // invariant: rowptr[lo] <= k < rowptr[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((int64_t)rowptr[mid] <= k) lo = mid; else hi = mid;
    }
    chunk_row[b] = lo;
}

struct f32x3_u { float x, y, z; } __attribute__((packed, aligned(4)));
typedef int spmv_v4i __attribute__((ext_vector_type(4)));
typedef float spmv_v4f __attribute__((ext_vector_type(4)));

template <int EPL, int VARIANT>
__global__ void __launch_bounds__(PCG_BLOCK) k_spmv(const int32_t* __restrict__ rowptr, const void* __restrict__ cols_,
                                                    const float* __restrict__ vals, int M, int nnz, int nchunks,
                                                    const int32_t* __restrict__ chunk_row, const float* __restrict__ x,
                                                    float* __restrict__ y, float* __restrict__ carry,
                                                    int32_t* __restrict__ carry_row, const int* __restrict__ done) {
    if (done && *done) return;
    typedef SpmvFmt<EPL> F;
    __shared__ __attribute__((aligned(16))) float prod[F::CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // chunks are dealt round-robin to the workgroups (a contiguous range per workgroup measured 8 %
    // slower: the concurrently active chunks then crowd the same HBM channels)
    // (round 3: giving every XCD one contiguous eighth of the chunks -- so that the x entries it gathers fit its own 4 MB L2 --
    // measured 6 % slower than round-robin, with or without the streaming hint below: same channel crowding)
    for (int b = blockIdx.x; b < nchunks; b += gridDim.x) {
        const int base = b * F::CHUNK;
        const int end = (base + F::CHUNK < nnz) ? base + F::CHUNK : nnz;
        // row pointers of the rows this wavefront will reduce (lane j: row r_first + wave + 4 j), fetched
        // now so that their latency hides behind the stream loads instead of stalling the row loop
        const int r_first = chunk_row[b], r_lim = chunk_row[b + 1];      // r_lim: row holding entry `end` (or M)
        int pre0 = 0x7fffffff, pre1 = 0x7fffffff;
        {
            const int r = r_first + wave + (PCG_BLOCK / 64) * lane;
            if (r <= r_lim && r < M) { pre0 = rowptr[r]; pre1 = rowptr[r + 1]; }
        }
        int c[F::QUADS][EPL];
        float v[F::QUADS][EPL];
#pragma unroll
        for (int q = 0; q < F::QUADS; ++q) {
            const int64_t g = (int64_t)(base + q * F::QUAD) / EPL + tid;       // lane slot (EPL entries)
            if (EPL == 4) {
                int4 ci;
                float4 vi;
                if (VARIANT != 2) {
                    const spmv_v4i cn = __builtin_nontemporal_load(reinterpret_cast<const spmv_v4i*>(cols_) + g);
                    const spmv_v4f vn = __builtin_nontemporal_load(reinterpret_cast<const spmv_v4f*>(vals) + g);
                    ci = make_int4(cn.x, cn.y, cn.z, cn.w); vi = make_float4(vn.x, vn.y, vn.z, vn.w);
                } else { ci = reinterpret_cast<const int4*>(cols_)[g]; vi = reinterpret_cast<const float4*>(vals)[g]; }
                c[q][0] = ci.x; c[q][1] = ci.y; c[q][2] = ci.z; c[q][EPL - 1] = ci.w;
                v[q][0] = vi.x; v[q][1] = vi.y; v[q][2] = vi.z; v[q][EPL - 1] = vi.w;
            } else {
                const unsigned long long pk = VARIANT != 2 ? __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(cols_) + g)
                                                           : reinterpret_cast<const unsigned long long*>(cols_)[g];
                f32x3_u vi;
                if (VARIANT != 2) {
                    const float* vp = vals + 3 * g;
                    vi.x = __builtin_nontemporal_load(vp); vi.y = __builtin_nontemporal_load(vp + 1); vi.z = __builtin_nontemporal_load(vp + 2);
                } else vi = reinterpret_cast<const f32x3_u*>(vals)[g];
                c[q][0] = (int)(pk & 0x1FFFFFull); c[q][1] = (int)((pk >> 21) & 0x1FFFFFull); c[q][2] = (int)((pk >> 42) & 0x1FFFFFull);
                v[q][0] = vi.x; v[q][1] = vi.y; v[q][2] = vi.z;
            }
        }
#pragma unroll
        for (int q = 0; q < F::QUADS; ++q) {
            float* pt = prod + q * F::QUAD + wave * F::TILE + lane;
#pragma unroll
            for (int j = 0; j < EPL; ++j)
                pt[64 * j] = v[q][j] * (VARIANT == 1 ? (float)c[q][j] : x[c[q][j]]);   // VARIANT 1: probe without the gather
        }
        __syncthreads();
        if (tid == 0 && pre0 >= base) carry_row[b] = -1;   // no row continues into this chunk
        int j = 0;
        for (int r = r_first + wave; r <= r_lim && r < M; r += PCG_BLOCK / 64, ++j) {
            int p0, p1;
            if (j < 64) { p0 = __builtin_amdgcn_readlane(pre0, j); p1 = __builtin_amdgcn_readlane(pre1, j); }
            else { p0 = rowptr[r]; p1 = rowptr[r + 1]; }
            if (p0 >= end) break;                          // row r_lim starts exactly at `end`: next chunk's
            const int k0 = p0 > base ? p0 : base, k1 = p1 < end ? p1 : end;
            float s = 0.f;
            for (int k = k0 + lane; k < k1; k += 64) s += prod[k - base];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) {
                if (p0 >= base) y[r] = s;
                else { carry[b] = s; carry_row[b] = r; }
            }
        }
        __syncthreads();
    }
}

// three int32 columns (< 2^21) of one lane slot -> one 64-bit word (format 1)
__global__ void k_pack_cols21(const int32_t* __restrict__ cols32, int64_t nslots, unsigned long long* __restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nslots) return;
    const unsigned long long a = (unsigned)cols32[3 * g], b = (unsigned)cols32[3 * g + 1], c = (unsigned)cols32[3 * g + 2];
    out[g] = a | (b << 21) | (c << 42);
}

// adds the per-chunk carries to y in chunk order (one thread per run of equal rows)
__global__ void k_spmv_fixup(int nchunks, const float* __restrict__ carry, const int32_t* __restrict__ carry_row,
                             float* __restrict__ y, const int* __restrict__ done) {
    if (done && *done) return;
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nchunks) return;
    const int r = carry_row[b];
    if (r < 0) return;
    if (b > 0 && carry_row[b - 1] == r) return;   // not the head of the run
    float s = 0.f;
    for (int j = b; j < nchunks && carry_row[j] == r; ++j) s += carry[j];
    y[r] += s;
}

// ---- reduction-block plan of the segmented PCG ------------------------------------------------------------------------------
__device__ __forceinline__ void seg_range(const PcgWork& w, int c, int k, int& lo, int& hi) {
    if (w.lo) { lo = w.lo[c * w.nranges + k]; hi = w.hi[c * w.nranges + k]; } else { lo = 0; hi = w.M; }
    if (hi < lo) hi = lo;
}
__device__ __forceinline__ int seg_range_blocks(const PcgWork& w, int c, int k) {
    int lo, hi;
    seg_range(w, c, k, lo, hi);
    return (hi - lo + SPCG_BS - 1) / SPCG_BS;
}
__global__ void k_seg_count(PcgWork w) {          // one thread per segment
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= w.nseg) return;
    int n = 0;
    for (int k = 0; k < w.nranges; ++k) n += seg_range_blocks(w, c, k);
    w.seg_blk[c + 1] = n;
}
__global__ void k_seg_scan(PcgWork w) {           // a few thousand segments at most: one thread
    int acc = 0;
    w.seg_blk[0] = 0;
    for (int c = 0; c < w.nseg; ++c) { acc += w.seg_blk[c + 1]; w.seg_blk[c + 1] = acc; }
    w.g->nblocks = acc;
    w.g->done_all = 0;
    w.g->done_count = 0;
    w.g->max_iter = 0;
    w.g->max_rel = 1.0;
}
__global__ void k_seg_fill(PcgWork w) {           // one wavefront per (segment, range)
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (idx >= w.nseg * w.nranges) return;
    const int c = idx / w.nranges, k = idx - c * w.nranges;
    int base = w.seg_blk[c];
    for (int q = 0; q < k; ++q) base += seg_range_blocks(w, c, q);
    int lo, hi;
    seg_range(w, c, k, lo, hi);
    const int nb = (hi - lo + SPCG_BS - 1) / SPCG_BS;
    for (int j = lane; j < nb; j += 64) {
        const int s = lo + j * SPCG_BS;
        w.blk_lo[base + j] = s;
        w.blk_hi[base + j] = s + SPCG_BS < hi ? s + SPCG_BS : hi;
        w.blk_seg[base + j] = c;
    }
}

// a segment finished (converged / empty / broke down): the launch-wide flag goes up when all have
__device__ __forceinline__ void seg_retire(PcgWork& w) {
    if (atomicAdd(&w.g->done_count, 1) + 1 == w.nseg) w.g->done_all = 1;
}

#define SPCG_BLOCK_PROLOGUE(check_done)                                \
    const int blk = blockIdx.x;                                        \
    if (blk >= w.g->nblocks) return;                                   \
    const int seg = w.blk_seg[blk];                                    \
    if ((check_done) && w.sc[seg].done) return;                        \
    const int lo = w.blk_lo[blk], hi = w.blk_hi[blk];                  \
    __shared__ double sm[PCG_BLOCK / 64];

__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_init(PcgWork w, const float* __restrict__ b, const float* __restrict__ diag,
                                                         float* __restrict__ x) {
    SPCG_BLOCK_PROLOGUE(false)
    double bb = 0.0, rz = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) {
        const float bi = b[i];
        const float zi = bi / diag[i];
        x[i] = 0.f;
        w.r[i] = bi;
        w.z[i] = zi;
        w.p[i] = zi;
        bb += (double)bi * bi;
        rz += (double)bi * zi;
    }
    const double t0 = block_sum(bb, sm);
    const double t1 = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        w.part2[3 * blk] = t0;
        w.part2[3 * blk + 1] = t1;
        w.part2[3 * blk + 2] = t1;
    }
}

// one workgroup per segment.  r.z <= 0 with the coarse-level block in z (the Chebyshev polynomial lost definiteness: eigenvalue
// bound too small) while the Jacobi r.z is positive: the segment starts with Jacobi alone (p = z = r / diag) instead of failing.
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_init_finish(PcgWork w, const float* __restrict__ diag) {
    __shared__ double sm[PCG_BLOCK / 64];
    const int c = blockIdx.x, b0 = w.seg_blk[c], nb = w.seg_blk[c + 1] - b0;
    const double bb = reduce_partials(w.part2 + 3 * (int64_t)b0, nb, 3, sm);
    const double rz = reduce_partials(w.part2 + 3 * (int64_t)b0 + 1, nb, 3, sm);
    const double rzj = reduce_partials(w.part2 + 3 * (int64_t)b0 + 2, nb, 3, sm);
    const bool fall = bb != 0.0 && !(rz > 0.0) && rzj > 0.0;
    if (fall)
        for (int k = 0; k < nb; ++k)
            for (int i = w.blk_lo[b0 + k] + threadIdx.x; i < w.blk_hi[b0 + k]; i += PCG_BLOCK) { const float zi = w.r[i] / diag[i]; w.z[i] = zi; w.p[i] = zi; }
    if (threadIdx.x == 0) {
        SegScalars& s = w.sc[c];
        s.bb = bb;
        s.rz[0] = fall ? rzj : rz;
        s.rz[1] = 0.0;
        s.rel = bb == 0.0 ? 0.0 : 1.0;
        s.iter = 0;
        s.done = 0;
        s.jacobi = fall ? 1 : 0;
        s.pad = 0;
        if (bb == 0.0 || !(s.rz[0] > 0.0)) { s.done = bb == 0.0 ? 1 : 2; seg_retire(w); }      // empty / zero right-hand side; 2 = breakdown
    }
}

// partial sums of p.Ap (fp64, fixed order).  Kept out of the operator on purpose: fused into the SpMV, lane 0 of every
// wavefront waited for a dependent x[r] load once per row, which cost the SpMV ~10 % (761 vs 680 us).
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_dot(PcgWork w) {
    SPCG_BLOCK_PROLOGUE(true)
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) acc += (double)w.p[i] * (double)w.y[i];
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) w.part1[blk] = t;
}

// x += alpha p ; r -= alpha y ; z = r / diag ; partial r.r and r.z   (alpha of the block's segment)
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_update(PcgWork w, const float* __restrict__ diag, float* __restrict__ x, int parity) {
    SPCG_BLOCK_PROLOGUE(true)
    const int b0 = w.seg_blk[seg];
    const double pAp = reduce_partials(w.part1 + b0, w.seg_blk[seg + 1] - b0, 1, sm);
    const float alpha = pAp > 0.0 ? (float)(w.sc[seg].rz[parity] / pAp) : 0.f;
    double rr = 0.0, rz = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) {
        const float pi = w.p[i], yi = w.y[i];
        x[i] = fmaf(alpha, pi, x[i]);
        const float ri = fmaf(-alpha, yi, w.r[i]);
        const float zi = ri / diag[i];
        w.r[i] = ri;
        w.z[i] = zi;
        rr += (double)ri * ri;
        rz += (double)ri * zi;
    }
    const double t0 = block_sum(rr, sm);
    const double t1 = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        w.part2[3 * blk] = t0;
        w.part2[3 * blk + 1] = t1;
        w.part2[3 * blk + 2] = t1;      // r.z with z = r / diag everywhere: what a fallback to Jacobi continues with
    }
}

// partial r.z again (second column of part2) after the coarse slice of z changed; init: p = z as well
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_rz(PcgWork w, int copy_p) {
    SPCG_BLOCK_PROLOGUE(!copy_p)
    if (!copy_p && w.sc[seg].jacobi) return;      // this segment's z is the Jacobi one of k_spcg_update: its partial is in place
    double rz = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) {
        const float zi = w.z[i];
        if (copy_p) w.p[i] = zi;
        rz += (double)w.r[i] * zi;
    }
    const double t = block_sum(rz, sm);
    if (threadIdx.x == 0) w.part2[3 * blk + 1] = t;
}

// p = z + beta p ; the first block of a segment publishes the scalars of its finished iteration
// Breakdown (r.z <= 0: the Chebyshev polynomial of the coarse-level block lost definiteness -- its eigenvalue bound was too small --
// or the residual ran away): the segment RESTARTS its conjugate gradients from the current x with Jacobi alone, p = z = r / diag,
// r.z from the third column of the partials (k_spcg_update); from then on the Chebyshev kernels skip it (SegScalars.jacobi).
// Every block of the segment takes the same decision from the same reduced numbers -- no flag is read here, so nothing races with
// the first block's write.  With Jacobi already in use the two r.z columns are equal, i.e. r.z <= 0 there is a hard failure.
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_pupdate(PcgWork w, const float* __restrict__ diag, int parity, float tol) {
    SPCG_BLOCK_PROLOGUE(true)
    const int b0 = w.seg_blk[seg], nb = w.seg_blk[seg + 1] - b0;
    const double rr = reduce_partials(w.part2 + 3 * (int64_t)b0, nb, 3, sm);
    const double rz_new = reduce_partials(w.part2 + 3 * (int64_t)b0 + 1, nb, 3, sm);
    const double rz_jac = reduce_partials(w.part2 + 3 * (int64_t)b0 + 2, nb, 3, sm);
    const double rz_old = w.sc[seg].rz[parity];
    const double bb = w.sc[seg].bb;
    const double rel = sqrt(rr / bb);
    const bool bad = !(rz_new > 0.0) || !(rel < 1e4);
    const bool fall = bad && rz_jac > 0.0 && rz_jac != rz_new && rel < 1e30;
    if (fall) {
        for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) { const float zi = w.r[i] / diag[i]; w.z[i] = zi; w.p[i] = zi; }
    } else {
        const float beta = (float)(rz_new / rz_old);
        for (int i = lo + threadIdx.x; i < hi; i += PCG_BLOCK) w.p[i] = fmaf(beta, w.p[i], w.z[i]);
    }
    __syncthreads();
    if (blk == b0 && threadIdx.x == 0) {
        SegScalars& s = w.sc[seg];
        s.rz[parity ^ 1] = fall ? rz_jac : rz_new;
        s.rel = rel;
        s.iter += 1;
        if (fall) s.jacobi = 1;
        if (rel <= (double)tol) { s.done = 1; seg_retire(w); }
        else if (!fall && (!(rz_new > 0.0) || !(rel < 1e30))) { s.done = 2; seg_retire(w); }      // r.z <= 0 with Jacobi (or NaN): nothing left to fall back to
    }
}

// what the host reads every check_every iterations (+ the per-segment results, when asked for)
__global__ void __launch_bounds__(PCG_BLOCK) k_spcg_summary(PcgWork w, double* __restrict__ seg_info) {
    __shared__ double smr[PCG_BLOCK];
    __shared__ int smi[PCG_BLOCK], smb[PCG_BLOCK], smn[PCG_BLOCK], smf[PCG_BLOCK];
    double mr = 0.0;
    int mi = 0, nbk = 0, mn = 0x7fffffff, nfb = 0;
    for (int c = threadIdx.x; c < w.nseg; c += PCG_BLOCK) {
        const SegScalars& s = w.sc[c];
        mr = s.rel > mr ? s.rel : mr;
        mi = s.iter > mi ? s.iter : mi;
        mn = s.iter < mn ? s.iter : mn;
        nbk += s.done == 2;
        nfb += s.jacobi != 0;
        if (seg_info) { seg_info[2 * c] = (double)s.iter; seg_info[2 * c + 1] = s.done == 2 ? -s.rel : s.rel; }
    }
    smr[threadIdx.x] = mr; smi[threadIdx.x] = mi; smb[threadIdx.x] = nbk; smn[threadIdx.x] = mn; smf[threadIdx.x] = nfb;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < PCG_BLOCK; ++t) { mr = smr[t] > mr ? smr[t] : mr; mi = smi[t] > mi ? smi[t] : mi; nbk += smb[t]; mn = smn[t] < mn ? smn[t] : mn; nfb += smf[t]; }
        w.g->max_rel = nbk ? -mr : mr;         // negative: some segment broke down (r.z <= 0)
        w.g->max_iter = mi;
        w.g->min_iter = mn;
        w.g->fallbacks = nfb;
    }
}

// ---- SpMV plan / workspace ------------------------------------------------------------------------
struct SpmvPlan {
    int nchunks;
    int32_t* chunk_row;   // [nchunks + 1]
    float* carry;         // [nchunks]
    int32_t* carry_row;   // [nchunks]
};

static int spmv_nchunks(int64_t nnz, int fmt) { return (int)((nnz + spmv_chunk(fmt) - 1) / spmv_chunk(fmt)); }

extern "C" size_t nksr_spmv_workspace_bytes(int64_t nnz) {
    size_t nc = (size_t)((nnz + SPMV_CHUNK_MIN - 1) / SPMV_CHUNK_MIN);   // enough for either format
    return align_up((nc + 1) * sizeof(int32_t), 256) + 2 * align_up(nc * sizeof(float), 256) + 256;
}

static SpmvPlan carve_spmv(void* ws, int64_t nnz, int fmt) {
    SpmvPlan p;
    p.nchunks = spmv_nchunks(nnz, fmt);
    char* c = (char*)ws;
    p.chunk_row = (int32_t*)c; c += align_up(((size_t)p.nchunks + 1) * sizeof(int32_t), 256);
    p.carry = (float*)c; c += align_up((size_t)p.nchunks * sizeof(float), 256);
    p.carry_row = (int32_t*)c;
    return p;
}

static int spmv_grid(int nchunks) {
    return nchunks < 1 ? 1 : (nchunks > PCG_MAX_BLOCKS ? PCG_MAX_BLOCKS : nchunks);
}

static int check_format(int col_format, int M) {
    if (col_format != 0 && col_format != 1) return nksr_set_error(NKSR_ERR_ARG, "col_format must be 0 or 1");
    if (col_format == 1 && M > (1 << 21)) return nksr_set_error(NKSR_ERR_CAPACITY, "col_format 1 packs 21-bit columns: M = %d > 2^21", M);
    return NKSR_OK;
}

extern "C" int nksr_spmv_plan(const int32_t* rowptr, int32_t M, int64_t nnz, int col_format, void* workspace, void* stream) {
    if (M <= 0 || nnz <= 0) return NKSR_OK;
    if (nnz >= ((int64_t)1 << 31) - SpmvFmt<3>::CHUNK) return nksr_set_error(NKSR_ERR_CAPACITY, "nnz exceeds int32");
    if (int rc = check_format(col_format, M)) return rc;
    SpmvPlan p = carve_spmv(workspace, nnz, col_format);
    hipLaunchKernelGGL(k_spmv_plan, dim3(nksr_blocks(p.nchunks + 1, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, M, nnz,
                       p.nchunks, spmv_chunk(col_format), p.chunk_row);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_pack_cols21(const int32_t* cols32, int64_t n_padded, uint64_t* packed_out, void* stream) {
    if (n_padded % SpmvFmt<3>::TILE) return nksr_set_error(NKSR_ERR_ARG, "n_padded must be a multiple of %d", SpmvFmt<3>::TILE);
    const int64_t nslots = n_padded / 3;
    if (nslots > 0) {
        hipLaunchKernelGGL(k_pack_cols21, dim3(nksr_blocks(nslots, 256)), dim3(256), 0, (hipStream_t)stream, cols32, nslots,
                           (unsigned long long*)packed_out);
        NKSR_CHECK_LAUNCH();
    }
    return NKSR_OK;
}

static int g_spmv_variant = 0;
extern "C" int nksr_spmv_set_variant(int v) { g_spmv_variant = v; return NKSR_OK; }

static int launch_spmv(const int32_t* rowptr, const void* cols, const float* vals, int M, int64_t nnz, int fmt, const SpmvPlan& p,
                       const float* x, float* y, const int* done, hipStream_t st) {
#define SPMV_LAUNCH(E, V) hipLaunchKernelGGL((k_spmv<E, V>), dim3(spmv_grid(p.nchunks)), dim3(PCG_BLOCK), 0, st, rowptr, cols, vals, M, (int)nnz, \
                       p.nchunks, p.chunk_row, x, y, p.carry, p.carry_row, done)
    if (fmt == 1) { if (g_spmv_variant == 1) SPMV_LAUNCH(3, 1); else if (g_spmv_variant == 2) SPMV_LAUNCH(3, 2); else SPMV_LAUNCH(3, 0); }
    else { if (g_spmv_variant == 1) SPMV_LAUNCH(4, 1); else if (g_spmv_variant == 2) SPMV_LAUNCH(4, 2); else SPMV_LAUNCH(4, 0); }
    hipLaunchKernelGGL(k_spmv_fixup, dim3(nksr_blocks(p.nchunks, 256)), dim3(256), 0, st, p.nchunks, p.carry, p.carry_row, y, done);
    return 0;
}

extern "C" int nksr_spmv_csr(const int32_t* rowptr, const void* cols, const float* vals, int32_t M, int64_t nnz, int col_format,
                             const float* x, float* y, void* workspace, void* stream) {
    if (M <= 0) return NKSR_OK;
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL (run nksr_spmv_plan first)");
    if (int rc = check_format(col_format, M)) return rc;
    SpmvPlan p = carve_spmv(workspace, nnz, col_format);
    launch_spmv(rowptr, cols, vals, M, nnz, col_format, p, x, y, nullptr, (hipStream_t)stream);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- optional live profiling of the SpMV launches (bench.py's roofline leg) ---------------------
#include <mutex>
#include <vector>
static int g_prof_enable = 0;
static double g_prof_ms = 0.0;
static long long g_prof_launches = 0;
static double g_prof_alg_bytes = 0.0, g_prof_phys_bytes = 0.0, g_prof_survey_bytes = 0.0, g_prof_last_survey = 0.0;
// several host threads may run solves at once (chunks on separate streams): every thread times with its own event pool and
// adds to the shared accumulators under a lock
static thread_local std::vector<hipEvent_t> g_prof_events;
static std::mutex g_prof_mutex;
static std::vector<float> g_prof_last_samples;
static std::vector<float> g_prof_samples;      // milliseconds of every timed application since the last nksr_pcg_profile call (at most 65536)

// bytes one SpMV launch moves: algorithmic CSR figure of SURVEY.md section 8d (8 nnz + 12 M + 4) and what the physical
// layout actually streams (values + packed / int32 columns over the padded storage + row pointers + x + y)
static void spmv_bytes(int M, int64_t nnz, int fmt, double* alg, double* phys) {
    *alg = 8.0 * (double)nnz + 12.0 * (double)M + 4.0;
    const int chunk = spmv_chunk(fmt);
    const double npad = (double)((nnz + chunk - 1) / chunk) * chunk;
    *phys = (fmt == 1 ? (4.0 + 8.0 / 3.0) : 8.0) * npad + 12.0 * (double)M + 4.0;
}

extern "C" int nksr_pcg_profile_bytes(double* algorithmic_out, double* physical_out) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (algorithmic_out) *algorithmic_out = g_prof_alg_bytes;
    if (physical_out) *physical_out = g_prof_phys_bytes;
    g_prof_last_survey = g_prof_survey_bytes;
    g_prof_alg_bytes = g_prof_phys_bytes = g_prof_survey_bytes = 0.0;
    return NKSR_OK;
}
// the same launches priced by SURVEY.md section 8d's formula (the CSR SpMV: identical to the algorithmic figure; the matrix-free
// operator: 16 bytes per stored entry, which it does not move) -- the value of the last nksr_pcg_profile_bytes call
extern "C" double nksr_pcg_profile_survey_bytes(void) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    return g_prof_last_survey;
}

extern "C" int nksr_pcg_profile(int enable, double* ms_out, int64_t* launches_out) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (ms_out) *ms_out = g_prof_ms;
    if (launches_out) *launches_out = g_prof_launches;
    g_prof_ms = 0.0;
    g_prof_launches = 0;
    g_prof_enable = enable;
    g_prof_last_samples.swap(g_prof_samples);
    g_prof_samples.clear();
    return NKSR_OK;
}
// the individual durations (milliseconds) behind the totals the LAST nksr_pcg_profile call returned: a slow box or a slow launch is
// visible in min / median, not in a mean
extern "C" int64_t nksr_pcg_profile_samples(float* ms_out, int64_t capacity) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    const int64_t n = (int64_t)g_prof_last_samples.size() < capacity ? (int64_t)g_prof_last_samples.size() : capacity;
    for (int64_t i = 0; i < n && ms_out; ++i) ms_out[i] = g_prof_last_samples[i];
    return (int64_t)g_prof_last_samples.size();
}

// ---- coarse-level block preconditioner -------------------------------------------------------------------------------------------
// The multi-level system is badly conditioned through its COARSE basis functions (each overlaps 124 neighbours of its own level and
// every finer voxel under its support): Jacobi needs 47 iterations per tree_depth-5 chunk where an exact solve of the diagonal
// block of the levels >= 2 (8 % of the unknowns, 4.5 % of the non-zeros) would need 12.  That block A_cc is assembled once per
// solve (plain CSR, nksr_assemble on the hierarchy with its fine levels masked) and z_c ~ A_cc^-1 r_c is approximated by a FIXED
// number of Jacobi-preconditioned Chebyshev steps -- a fixed polynomial in A_cc, hence a constant SPD preconditioner (plain CG
// stays valid) and deterministic.  One kernel per step, one wavefront per row:
//   t = (A_cc d)_j;  y_j += d_j;  res_j -= t;  d'_j = a d_j + b res_j / D_j          (d double-buffered: rows read their neighbours' d)
// Batched chunk solves: A_cc is block diagonal, every segment has its own eigenvalue bound and therefore its own polynomial
// (coefficient table coef[segment][1 + 2 i], row_seg = segment of every coarse row).
#define CHEB_STRIDE (1 + 2 * NKSR_PC_MAX_STEPS)

// coef[c] = { 1 / theta, a_0, b_0, a_1, b_1, ... } for the interval [lmax / ratio, lmax], lmax = scale * lambda[c]; a segment without
// a usable bound (no constraint rows on its coarse levels) gets { 1, 0, 0, ... }: one Jacobi step
// gersh (may be NULL): a Gershgorin bound of the same spectrum (nksr_coarse_gershgorin) -- a true upper bound, so the margin on the
// power estimate never has to carry the interval beyond it
__global__ void k_cheb_coeffs(int nseg, const float* __restrict__ lambda, const float* __restrict__ gersh, float scale, float ratio, int steps,
                              float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nseg) return;
    float* o = coef + (int64_t)c * CHEB_STRIDE;
    double lmax = (double)scale * (double)lambda[c];
    if (gersh && gersh[c] > 0.f && (double)gersh[c] < lmax) lmax = (double)gersh[c];
    if (!(lmax > 0.0) || !(lmax < 1e30)) {
        o[0] = 1.f;
        for (int i = 0; i < steps; ++i) { o[1 + 2 * i] = 0.f; o[2 + 2 * i] = 0.f; }
        return;
    }
    const double lmin = lmax / (double)ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    o[0] = (float)(1.0 / theta);
    for (int i = 0; i < steps; ++i) {
        const double rho_n = 1.0 / (2.0 * sigma - rho);
        o[1 + 2 * i] = (float)(rho_n * rho);
        o[2 + 2 * i] = (float)(2.0 * rho_n / delta);
        rho = rho_n;
    }
}

__global__ void __launch_bounds__(256) k_cheb_init(int n, const float* __restrict__ r, const float* __restrict__ diag,
                                                   const float* __restrict__ coef, const int32_t* __restrict__ row_seg,
                                                   float* __restrict__ res, float* __restrict__ d0, float* __restrict__ y,
                                                   const SegScalars* __restrict__ sc) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int seg = row_seg ? row_seg[j] : 0;
    if (sc && (sc[seg].done || sc[seg].jacobi)) return;
    const float rj = r[j];
    res[j] = rj;
    d0[j] = rj / diag[j] * (coef ? coef[(int64_t)seg * CHEB_STRIDE] : 1.f);
    y[j] = 0.f;
}

__global__ void __launch_bounds__(256) k_cheb_step(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                   const float* __restrict__ vals, const float* __restrict__ diag,
                                                   const float* __restrict__ coef, int step, const int32_t* __restrict__ row_seg,
                                                   float* __restrict__ res, const float* __restrict__ d_old, float* __restrict__ d_new,
                                                   float* __restrict__ y, const SegScalars* __restrict__ sc) {
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (j >= n) return;
    const int seg = row_seg ? row_seg[j] : 0;
    if (sc && (sc[seg].done || sc[seg].jacobi)) return;
    // a wavefront's time is a chain of dependent load latencies (the block streams from L2 / HBM): everything that does not depend on
    // the row's entries is requested up front, and four 64-entry groups (rows hold ~190 entries) are in flight at once -- clamped
    // addresses instead of predicated loads, so that the compiler does not serialise them
    const int k0 = rowptr[j], k1 = rowptr[j + 1];
    const float dj = d_old[j], rs = res[j], dg = diag[j], yj = y[j];
    const float a = coef[(int64_t)seg * CHEB_STRIDE + 1 + 2 * step], b = coef[(int64_t)seg * CHEB_STRIDE + 2 + 2 * step];
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = k0 + lane; base - lane < k1; base += 256) {
        float v[4];
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = base + 64 * q, kc = k < k1 ? k : k1 - 1;
            v[q] = vals[kc];
            c[q] = cols[kc];
            if (k >= k1) v[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = fmaf(v[q], d_old[c[q]], t[q]);
    }
    float tt = (t[0] + t[1]) + (t[2] + t[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o);
    if (lane == 0) {
        const float rj = rs - tt;
        y[j] = yj + dj;
        res[j] = rj;
        d_new[j] = fmaf(a, dj, b * rj / dg);
    }
}

// v' = D^-1 A_cc v (power iteration for the largest eigenvalue of the Jacobi-scaled block)
__global__ void __launch_bounds__(256) k_coarse_power(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                      const float* __restrict__ vals, const float* __restrict__ diag,
                                                      const float* __restrict__ v, float* __restrict__ out) {
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (j >= n) return;
    const int k0 = rowptr[j], k1 = rowptr[j + 1];
    const float dg = diag[j];
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = k0 + lane; base - lane < k1; base += 256) {          // as k_cheb_step: four entry groups in flight, clamped addresses
        float a[4];
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = base + 64 * q, kc = k < k1 ? k : k1 - 1;
            a[q] = vals[kc];
            c[q] = cols[kc];
            if (k >= k1) a[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[q] = fmaf(a[q], v[c[q]], t4[q]);
    }
    float t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) out[j] = t / dg;
}
// out[c] = sqrt(sum b^2 / sum a^2) over the coarse rows of segment c: one workgroup per segment, ranges in order, fixed strides
// (segment-relative order).  Without segments: the whole block.
__global__ void __launch_bounds__(PCG_BLOCK) k_norm_ratio(int n, int first, int nranges, const int32_t* __restrict__ lo,
                                                          const int32_t* __restrict__ hi, const float* __restrict__ a,
                                                          const float* __restrict__ b, float* __restrict__ out) {
    __shared__ double sm[PCG_BLOCK / 64];
    const int c = blockIdx.x;
    double sa = 0.0, sb = 0.0;
    for (int k = 0; k < nranges; ++k) {
        int l = lo ? lo[c * nranges + k] - first : 0, h = lo ? hi[c * nranges + k] - first : n;
        if (h > n) h = n;
        if (l < 0) { if (h <= 0) continue; l = 0; }       // ranges of the fine levels lie below `first`
        for (int i = l + threadIdx.x; i < h; i += PCG_BLOCK) { sa += (double)a[i] * a[i]; sb += (double)b[i] * b[i]; }
    }
    const double ta = block_sum(sa, sm), tb = block_sum(sb, sm);
    if (threadIdx.x == 0) out[c] = ta > 0.0 ? (float)sqrt(tb / ta) : 0.f;
}

extern "C" int nksr_coarse_lambda_max(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n, int iters,
                                      float* work, float* lambda_out, const nksr_segments_t* seg, int32_t first, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!rowptr || !cols || !vals || !diag || !work || !lambda_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    if (seg && (seg->nseg < 1 || seg->nranges < 1 || !seg->lo || !seg->hi)) return nksr_set_error(NKSR_ERR_ARG, "bad segments");
    if (iters < 2) iters = 2;
    hipStream_t st = (hipStream_t)stream;
    float* v[2] = {work, work + n};
    const dim3 g1(nksr_blocks(n, 256)), gw(nksr_blocks((int64_t)n * 64, 256));
    hipLaunchKernelGGL(k_cheb_init, g1, dim3(256), 0, st, n, diag, diag, (const float*)nullptr, (const int32_t*)nullptr, v[1], v[0], v[1],
                       (const SegScalars*)nullptr);   // v0 = 1 (diag / diag)
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_coarse_power, gw, dim3(256), 0, st, n, rowptr, cols, vals, diag, (const float*)v[i & 1], v[(i + 1) & 1]);
    hipLaunchKernelGGL(k_norm_ratio, dim3(seg ? seg->nseg : 1), dim3(PCG_BLOCK), 0, st, n, seg ? first : 0, seg ? seg->nranges : 1,
                       seg ? seg->lo : (const int32_t*)nullptr, seg ? seg->hi : (const int32_t*)nullptr, (const float*)v[(iters - 1) & 1],
                       (const float*)v[iters & 1], lambda_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- packed coarse block (nksr_coarse_precond_t.format 1) -----------------------------------------------------------------------
// A Chebyshev step streams the whole block: 8 bytes per entry (fp32 value + int32 column) x 4.2e8 entries x 8 steps per PCG
// iteration on the 64-chunk scene -- a third of the iteration.  The preconditioner only has to be a FIXED symmetric positive
// operator, so the block is stored Jacobi-scaled, S = D^-1/2 A_cc D^-1/2 (unit diagonal, |S_ij| <= 1): an entry is a 16-bit
// half-precision value and a 16-bit column LOCAL to the row's segment -- the coarse unknowns are renumbered segment-major
// (old_of_new), so a chunk's columns span < 2^16 -- i.e. 4 bytes.  The recurrence runs in the scaled variables
//   res' = D^-1/2 res,  d' = D^1/2 d:   t = S d';  y' += d';  res' -= t;  d' <- a d' + b res';      z = D^-1/2 y'
// (the same polynomial in D^-1 A_cc, up to the rounding of S to half precision: S_ij = half(v_ij * (dis_i * dis_j)) is bitwise symmetric).
#include <hip/hip_fp16.h>

// Entries with |S_ij| < drop are left out (drop = 0 keeps all): the far corners of the 5 x 5 x 5 stencil carry 1e-2 .. 1e-6 of
// the diagonal and most of the bytes; the dropped matrix is still symmetric (S_ij and S_ji are the same bits) and the polynomial of
// it is still a fixed symmetric positive operator.
__device__ __forceinline__ __half cc_scaled(float v, float di, float dj) { return __float2half_rn(v * (di * dj)); }
__device__ __forceinline__ bool cc_keep(__half hv, float drop) { return fabsf(__half2float(hv)) >= drop; }

__global__ void __launch_bounds__(256) k_cc_count(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                  const float* __restrict__ vals, const float* __restrict__ diag,
                                                  const int32_t* __restrict__ old_of_new, float drop, int32_t* __restrict__ lens) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const int j = old_of_new[i];
    const int k0 = rowptr[j], k1 = rowptr[j + 1] - 1;
    const float di = 1.f / sqrtf(diag[j]);
    int cnt = 0;
    for (int k = k0 + lane; k < k1; k += 64) cnt += cc_keep(cc_scaled(vals[k], di, 1.f / sqrtf(diag[cols[k]])), drop) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) lens[i] = cnt;
}

__global__ void __launch_bounds__(256) k_cc_pack(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                 const float* __restrict__ vals, const float* __restrict__ diag,
                                                 const int32_t* __restrict__ old_of_new, const int32_t* __restrict__ new_of_old,
                                                 const int32_t* __restrict__ row_seg, const int32_t* __restrict__ seg_base,
                                                 const int32_t* __restrict__ prow, float drop, uint32_t* __restrict__ pk, float* __restrict__ dis) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const int j = old_of_new[i];
    const int k0 = rowptr[j], k1 = rowptr[j + 1] - 1;          // the diagonal closes every row: it is dropped (unit diagonal)
    const float di = 1.f / sqrtf(diag[j]);
    const int base = seg_base[row_seg[i]];
    int o = prow[i];
    if (lane == 0) dis[i] = di;
    for (int kb = k0; kb < k1; kb += 64) {                      // kept entries keep their order
        const int k = kb + lane;
        bool keep = false;
        uint32_t word = 0u;
        if (k < k1) {
            const int c = cols[k];
            const __half hv = cc_scaled(vals[k], di, 1.f / sqrtf(diag[c]));
            keep = cc_keep(hv, drop);
            word = ((uint32_t)__half_as_ushort(hv) << 16) | (uint32_t)(new_of_old[c] - base);
        }
        const unsigned long long m = __ballot(keep);
        if (keep) pk[o + __popcll(m & ((1ull << lane) - 1ull))] = word;
        o += __popcll(m);
    }
}

__global__ void __launch_bounds__(256) k_cheb16_init(int n, const float* __restrict__ r, const int32_t* __restrict__ old_of_new,
                                                     const float* __restrict__ dis, const float* __restrict__ coef,
                                                     const int32_t* __restrict__ row_seg, float* __restrict__ res, float* __restrict__ d0,
                                                     float* __restrict__ y, const SegScalars* __restrict__ sc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int seg = row_seg[i];
    if (sc && (sc[seg].done || sc[seg].jacobi)) return;
    const float rs = r[old_of_new[i]] * dis[i];
    res[i] = rs;
    d0[i] = rs * (coef ? coef[(int64_t)seg * CHEB_STRIDE] : 1.f);
    y[i] = 0.f;
}

// One wavefront per CHEB16_ROWS = 16 consecutive rows, 16 lanes per row: lane (g, l) = (lane >> 4, lane & 15) works for the four rows
// i0 + 4 g + r, r < 4, and takes entries l, l + 16, ... of each.  LAST: the step also leaves z = D^-1/2 (y' + d') in the PCG's order.
// POWER: out = S v only (eigenvalue bound).
// Round 6.  After the drop (entries below 0.5 % of the unit diagonal) a row of the 64-chunk scene keeps 37 entries on average (median
// 28, a tenth none, 1 % more than 300: tools/cheb_rows_probe.py), not the ~240 it is assembled with.  Rounds 3-5 gave every row one
// 256-entry trip of a whole wavefront (four rows per wavefront: 16 entry loads + 16 gathers, ~85 % of their lanes clamped and masked)
// and ran at the latency of row pointers -> entries -> gathered vector over 380 000 wavefronts per step.  Sixteen lanes per row put
// the same 16 + 16 loads in flight for FOUR times the rows -- a row's partial sums depend on its own length only, so its result does
// not depend on its wave mates (a chunk alone == the chunk in a batch).
#define CHEB16_ROWS 16
template <bool POWER>
__global__ void __launch_bounds__(256) k_cheb16_step(int n, const int32_t* __restrict__ prow, const uint32_t* __restrict__ pk,
                                                     const float* __restrict__ coef, int step, const int32_t* __restrict__ row_seg,
                                                     const int32_t* __restrict__ seg_base, float* __restrict__ res,
                                                     const float* __restrict__ d_old, float* __restrict__ d_new, float* __restrict__ y,
                                                     const SegScalars* __restrict__ sc, int last, float* __restrict__ z,
                                                     const int32_t* __restrict__ old_of_new, const float* __restrict__ dis) {
    const int i0 = ((blockIdx.x * 256 + threadIdx.x) >> 6) * CHEB16_ROWS, lane = threadIdx.x & 63;
    if (i0 >= n) return;
    const int g = lane >> 4, l = lane & 15;
    // per-row scalars: lane q (< 16) owns row i0 + q; the row pointers come as one load of 17 lanes
    const int ir = i0 + (lane < CHEB16_ROWS ? lane : 0), irc = ir < n ? ir : n - 1;
    const int pl = prow[(i0 + lane <= n && lane <= CHEB16_ROWS) ? i0 + lane : n];
    const int sg = row_seg[irc];
    const bool live_l = lane < CHEB16_ROWS && ir < n && (POWER || !sc || !(sc[sg].done || sc[sg].jacobi));
    const int sb = seg_base[sg];
    const float dj = d_old[irc];
    float rs = 0.f, yj = 0.f, ca = 0.f, cb = 0.f;
    if (!POWER) {
        rs = res[irc]; yj = y[irc];
        ca = coef[(int64_t)sg * CHEB_STRIDE + 1 + 2 * step]; cb = coef[(int64_t)sg * CHEB_STRIDE + 2 + 2 * step];
    }
    // the four rows of this lane's group: first / last entry, vector base of their segment (a dead row is empty)
    int k0[4], k1[4], base_r[4];
    int maxlen = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = 4 * g + r;
        k0[r] = __shfl(pl, q);
        k1[r] = __shfl(pl, q + 1);
        base_r[r] = __shfl(sb, q);
        if (!__shfl((int)live_l, q)) k1[r] = k0[r];
        maxlen = (k1[r] - k0[r]) > maxlen ? (k1[r] - k0[r]) : maxlen;
    }
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) { const int m = __shfl_xor(maxlen, o); maxlen = m > maxlen ? m : maxlen; }     // (over the four groups: one trip count per wavefront)
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < maxlen; b += 64) {                     // four 16-entry trips of every row at once: 16 loads, then 16 gathers
        uint32_t w[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0[r] + b + 16 * q + l;
                const int kc = k < k1[r] ? k : (k1[r] > k0[r] ? k1[r] - 1 : 0);          // clamped address, value masked below
                w[q][r] = pk[kc];
                if (k >= k1[r]) w[q][r] = 0u;                                              // value +0.0, column 0 of the segment
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                t[r] = fmaf(__half2float(__ushort_as_half((unsigned short)(w[q][r] >> 16))), d_old[base_r[r] + (int)(w[q][r] & 0xFFFFu)], t[r]);
    }
    // the 16 lanes of a group sum every row (fixed tree); lane q of the wavefront ends up with the sum of row q
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float tt = t[r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) tt += __shfl_xor(tt, o);
        const float got = __shfl(tt, 16 * (lane >> 2) + 0);   // row q = 4 g' + r lives in group g' = q >> 2: lane 16 g' holds its sum
        if (lane < CHEB16_ROWS && (lane & 3) == r) tot = got;
    }
    if (live_l) {
        tot += dj;                                                        // the unit diagonal
        if (POWER) { d_new[ir] = tot; return; }
        const float rj = rs - tot;
        res[ir] = rj;
        d_new[ir] = fmaf(ca, dj, cb * rj);
        if (last) z[old_of_new[ir]] = (yj + dj) * dis[ir];
        else y[ir] = yj + dj;
    }
}

__global__ void __launch_bounds__(256) k_fill1(int n, float* __restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = 1.f;
}
// out[c] = ||b|| / ||a|| over the rows [seg_base[c], seg_base[c + 1]) (segment-major order): one workgroup per segment
__global__ void __launch_bounds__(PCG_BLOCK) k_norm_ratio_seg(const int32_t* __restrict__ seg_base, const float* __restrict__ a,
                                                              const float* __restrict__ b, float* __restrict__ out) {
    __shared__ double sm[PCG_BLOCK / 64];
    const int c = blockIdx.x;
    double sa = 0.0, sb = 0.0;
    for (int i = seg_base[c] + threadIdx.x; i < seg_base[c + 1]; i += PCG_BLOCK) { sa += (double)a[i] * a[i]; sb += (double)b[i] * b[i]; }
    const double ta = block_sum(sa, sm), tb = block_sum(sb, sm);
    if (threadIdx.x == 0) out[c] = ta > 0.0 ? (float)sqrt(tb / ta) : 0.f;
}

extern "C" int nksr_coarse_pack_count(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n,
                                      const int32_t* old_of_new, float drop_tol, int32_t* lens_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!rowptr || !cols || !vals || !diag || !old_of_new || !lens_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    if (!(drop_tol >= 0.f) || drop_tol >= 1.f) return nksr_set_error(NKSR_ERR_ARG, "drop_tol must be in [0, 1)");
    hipLaunchKernelGGL(k_cc_count, dim3(nksr_blocks((int64_t)n * 64, 256)), dim3(256), 0, (hipStream_t)stream, n, rowptr, cols, vals, diag, old_of_new,
                       drop_tol, lens_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_coarse_pack(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n,
                                const int32_t* old_of_new, const int32_t* new_of_old, const int32_t* row_seg_new, const int32_t* seg_base,
                                const int32_t* packed_rowptr, float drop_tol, uint32_t* packed_out, float* dis_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!rowptr || !cols || !vals || !diag || !old_of_new || !new_of_old || !row_seg_new || !seg_base || !packed_rowptr || !packed_out || !dis_out)
        return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    if (!(drop_tol >= 0.f) || drop_tol >= 1.f) return nksr_set_error(NKSR_ERR_ARG, "drop_tol must be in [0, 1)");
    hipLaunchKernelGGL(k_cc_pack, dim3(nksr_blocks((int64_t)n * 64, 256)), dim3(256), 0, (hipStream_t)stream, n, rowptr, cols, vals, diag,
                       old_of_new, new_of_old, row_seg_new, seg_base, packed_rowptr, drop_tol, packed_out, dis_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_coarse_lambda_max_packed(const nksr_coarse_precond_t* pc, int32_t nseg, int iters, float* work, float* lambda_out, void* stream) {
    if (!pc || pc->n <= 0) return NKSR_OK;
    if (pc->format != 1 || !pc->packed || !pc->packed_rowptr || !pc->row_seg || !pc->seg_base || !work || !lambda_out)
        return nksr_set_error(NKSR_ERR_ARG, "packed coarse block has NULL arrays");
    if (iters < 2) iters = 2;
    hipStream_t st = (hipStream_t)stream;
    const int n = pc->n;
    float* v[2] = {work, work + n};
    hipLaunchKernelGGL(k_fill1, dim3(nksr_blocks(n, 256)), dim3(256), 0, st, n, v[0]);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((k_cheb16_step<true>), dim3(nksr_blocks(((int64_t)n + CHEB16_ROWS - 1) / CHEB16_ROWS * 64, 256)), dim3(256), 0, st, n, pc->packed_rowptr, pc->packed,
                           (const float*)nullptr, 0, pc->row_seg, pc->seg_base, (float*)nullptr, (const float*)v[i & 1], v[(i + 1) & 1], (float*)nullptr,
                           (const SegScalars*)nullptr, 0, (float*)nullptr, (const int32_t*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(k_norm_ratio_seg, dim3(nseg), dim3(PCG_BLOCK), 0, st, pc->seg_base, (const float*)v[(iters - 1) & 1], (const float*)v[iters & 1], lambda_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// Gershgorin bound of the Jacobi-scaled block per segment: max over the rows of  sum_j |a_ij| / d_i  (format 0, spectrum of D^-1 A_cc)
// or  1 + sum_j |S_ij|  (format 1: unit diagonal).  2-3x above the true lambda_max on these blocks -- too loose to BE the interval
// (it costs ~50 % more iterations), right as a cap on margin x power estimate.
__global__ void __launch_bounds__(256) k_gersh_rows(int n, int format, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                    const float* __restrict__ vals, const float* __restrict__ diag,
                                                    const uint32_t* __restrict__ pk, float* __restrict__ rowsum) {
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    float t = 0.f;
    if (format == 1) {
        for (int k = k0 + lane; k < k1; k += 64) t += fabsf(__half2float(__ushort_as_half((unsigned short)(pk[k] >> 16))));
    } else {
        for (int k = k0 + lane; k < k1; k += 64) t += fabsf(vals[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) rowsum[i] = format == 1 ? 1.f + t : t / diag[i];
}
__global__ void __launch_bounds__(PCG_BLOCK) k_gersh_seg(int n, const int32_t* __restrict__ seg_base, const int32_t* __restrict__ row_seg,
                                                         int nseg, const float* __restrict__ rowsum, float* __restrict__ out) {
    __shared__ float sm[PCG_BLOCK];
    const int c = blockIdx.x;
    float m = 0.f;
    if (seg_base) {
        for (int i = seg_base[c] + threadIdx.x; i < seg_base[c + 1]; i += PCG_BLOCK) m = fmaxf(m, rowsum[i]);
    } else {
        for (int i = threadIdx.x; i < n; i += PCG_BLOCK)
            if (nseg == 1 || !row_seg || row_seg[i] == c) m = fmaxf(m, rowsum[i]);
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int o = PCG_BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sm[0];
}
extern "C" int nksr_coarse_gershgorin(const nksr_coarse_precond_t* pc, int32_t nseg, float* work, float* gersh_out, void* stream) {
    if (!pc || pc->n <= 0) return NKSR_OK;
    if (!work || !gersh_out || nseg < 1) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    if (pc->format == 1 ? (!pc->packed || !pc->packed_rowptr || !pc->seg_base) : (!pc->rowptr || !pc->vals || !pc->diag || (nseg > 1 && !pc->row_seg)))
        return nksr_set_error(NKSR_ERR_ARG, "coarse block has NULL arrays");
    hipStream_t st = (hipStream_t)stream;
    const int n = pc->n;
    hipLaunchKernelGGL(k_gersh_rows, dim3(nksr_blocks((int64_t)n * 64, 256)), dim3(256), 0, st, n, pc->format, pc->format == 1 ? pc->packed_rowptr : pc->rowptr,
                       pc->cols, pc->vals, pc->diag, pc->packed, work);
    hipLaunchKernelGGL(k_gersh_seg, dim3(nseg), dim3(PCG_BLOCK), 0, st, n, pc->format == 1 ? pc->seg_base : (const int32_t*)nullptr, pc->row_seg, nseg,
                       (const float*)work, gersh_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

static int cheb_check(const nksr_coarse_precond_t* pc, int nseg) {
    if (pc->n <= 0 || pc->steps < 1 || pc->steps > NKSR_PC_MAX_STEPS) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: 1..%d steps", NKSR_PC_MAX_STEPS);
    if (!(pc->lambda_scale > 0.f) || !(pc->ratio > 1.f)) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: lambda_scale > 0, ratio > 1");
    if (!pc->work || !pc->lambda || !pc->coef) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner has NULL arrays");
    if (pc->format == 1) {
        if (!pc->packed || !pc->packed_rowptr || !pc->dis || !pc->old_of_new || !pc->seg_base || !pc->row_seg)
            return nksr_set_error(NKSR_ERR_ARG, "packed coarse block has NULL arrays");
        return NKSR_OK;
    }
    if (!pc->rowptr || !pc->cols || !pc->vals || !pc->diag) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner has NULL arrays");
    if (nseg > 1 && !pc->row_seg) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: row_seg is required with more than one segment");
    return NKSR_OK;
}
// z_c = p_k(A_cc) r_c  (r, z: the coarse slices of the PCG vectors)
static void cheb_apply(const nksr_coarse_precond_t* pc, const float* r, float* z, const SegScalars* sc, hipStream_t st) {
    const int n = pc->n;
    float *res = pc->work, *d[2] = {pc->work + n, pc->work + 2 * (size_t)n};
    if (pc->format == 1) {      // packed block: scaled variables, segment-major order; y lives in the work array (z is written by the last step)
        float* y = pc->work + 3 * (size_t)n;
        hipLaunchKernelGGL(k_cheb16_init, dim3(nksr_blocks(n, 256)), dim3(256), 0, st, n, r, pc->old_of_new, pc->dis, (const float*)pc->coef, pc->row_seg,
                           res, d[0], y, sc);
        for (int i = 0; i < pc->steps; ++i)
            hipLaunchKernelGGL((k_cheb16_step<false>), dim3(nksr_blocks(((int64_t)n + CHEB16_ROWS - 1) / CHEB16_ROWS * 64, 256)), dim3(256), 0, st, n, pc->packed_rowptr, pc->packed,
                               (const float*)pc->coef, i, pc->row_seg, pc->seg_base, res, (const float*)d[i & 1], d[(i + 1) & 1], y, sc,
                               i == pc->steps - 1 ? 1 : 0, z, pc->old_of_new, pc->dis);
        return;
    }
    hipLaunchKernelGGL(k_cheb_init, dim3(nksr_blocks(n, 256)), dim3(256), 0, st, n, r, pc->diag, (const float*)pc->coef, pc->row_seg, res, d[0], z, sc);
    for (int i = 0; i < pc->steps; ++i)
        hipLaunchKernelGGL(k_cheb_step, dim3(nksr_blocks((int64_t)n * 64, 256)), dim3(256), 0, st, n, pc->rowptr, pc->cols, pc->vals, pc->diag,
                           (const float*)pc->coef, i, pc->row_seg, res, (const float*)d[i & 1], d[(i + 1) & 1], z, sc);
}

int nksr_pcg_run(PcgOperator& A, const float* diag, int32_t M, const float* b, float* x, float tol, int max_iter, int check_every,
                 void* vector_workspace, double* info_out, hipStream_t st, const nksr_coarse_precond_t* pc, const nksr_segments_t* seg) {
    if (M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; info_out[2] = 0; } return NKSR_OK; }
    if (!vector_workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    if (seg && (seg->nseg < 1 || seg->nranges < 1 || !seg->lo || !seg->hi)) return nksr_set_error(NKSR_ERR_ARG, "bad segments");
    if (check_every < 1) check_every = 1;
    PcgWork w = carve(vector_workspace, M, seg);
    const dim3 gb(w.nb_max), blkd(PCG_BLOCK);
    if (pc) {
        if (int rc = cheb_check(pc, w.nseg)) return rc;
        if (pc->first < 0 || pc->first + pc->n != M) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: the block must be the last %d unknowns", pc->n);
        hipLaunchKernelGGL(k_cheb_coeffs, dim3(nksr_blocks(w.nseg, 64)), dim3(64), 0, st, w.nseg, pc->lambda, pc->gersh, pc->lambda_scale, pc->ratio, pc->steps, pc->coef);
    }
    hipLaunchKernelGGL(k_seg_count, dim3(nksr_blocks(w.nseg, 64)), dim3(64), 0, st, w);
    hipLaunchKernelGGL(k_seg_scan, dim3(1), dim3(1), 0, st, w);
    hipLaunchKernelGGL(k_seg_fill, dim3(nksr_blocks((int64_t)w.nseg * w.nranges * 64, 256)), dim3(256), 0, st, w);
    hipLaunchKernelGGL(k_spcg_init, gb, blkd, 0, st, w, b, diag, x);
    if (pc) {
        cheb_apply(pc, w.r + pc->first, w.z + pc->first, nullptr, st);
        hipLaunchKernelGGL(k_spcg_rz, gb, blkd, 0, st, w, 1);
    }
    hipLaunchKernelGGL(k_spcg_init_finish, dim3(w.nseg), blkd, 0, st, w, diag);
    NKSR_CHECK_LAUNCH();
    PcgGlobal host;
    memset(&host, 0, sizeof(host));
    int launched = 0;
    const bool prof = g_prof_enable != 0;
    if (prof)
        while ((int)g_prof_events.size() < 2 * check_every) {
            hipEvent_t e;
            NKSR_CHECK_HIP(hipEventCreate(&e));
            g_prof_events.push_back(e);
        }
    // (Tried in round 3 and removed: capturing two iterations into a hipGraph and replaying it for small systems -- configs[1] and the
    // 10 000-point bunny sequence spend 5 / 8 ms in 42 / 93 iterations of 6-20 tiny launches.  No change (5.07 vs 4.56 ms, 8.76 vs 8.30):
    // the bound is the dependency latency between consecutive tiny kernels on the device, ~15 us each, not the host enqueue.)
    auto iterate = [&](int parity, int c) -> int {
        if (prof) (void)hipEventRecord(g_prof_events[2 * c], st);
        if (int rc = A.apply(w.p, w.y, &w.g->done_all, w.nseg > 1 ? &w.sc[0].done : nullptr, (int)(sizeof(SegScalars) / sizeof(int)), &w.g->done_count, w.nseg, st)) return rc;
        if (prof) (void)hipEventRecord(g_prof_events[2 * c + 1], st);
        hipLaunchKernelGGL(k_spcg_dot, gb, blkd, 0, st, w);
        hipLaunchKernelGGL(k_spcg_update, gb, blkd, 0, st, w, diag, x, parity);
        if (pc) {
            cheb_apply(pc, w.r + pc->first, w.z + pc->first, w.sc, st);
            hipLaunchKernelGGL(k_spcg_rz, gb, blkd, 0, st, w, 0);
        }
        hipLaunchKernelGGL(k_spcg_pupdate, gb, blkd, 0, st, w, diag, parity, tol);
        return NKSR_OK;
    };
    while (launched < max_iter) {
        int chunk = check_every < (max_iter - launched) ? check_every : (max_iter - launched);
        for (int c = 0; c < chunk; ++c)
            if (int rc = iterate((launched + c) & 1, c)) return rc;
        hipLaunchKernelGGL(k_spcg_summary, dim3(1), blkd, 0, st, w, seg ? seg->info : (double*)nullptr);
        NKSR_CHECK_LAUNCH();
        NKSR_CHECK_HIP(hipMemcpyAsync(&host, w.g, sizeof(host), hipMemcpyDeviceToHost, st));
        NKSR_CHECK_HIP(hipStreamSynchronize(st));
        if (prof) {
            // only applications in which EVERY segment was still iterating (the done flag turns later launches into no-ops, and the
            // operator skips the rows of segments that have converged: those move fewer bytes than A.bytes() says)
            double ba, bp, bs;
            A.bytes(&ba, &bp, &bs);
            std::lock_guard<std::mutex> lock(g_prof_mutex);
            for (int c = 0; c < chunk && launched + c < host.min_iter; ++c) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g_prof_events[2 * c], g_prof_events[2 * c + 1]) == hipSuccess) {
                    g_prof_ms += ms;
                    g_prof_launches += 1;
                    if (g_prof_samples.size() < 65536) g_prof_samples.push_back(ms);
                    g_prof_alg_bytes += ba;
                    g_prof_phys_bytes += bp;
                    g_prof_survey_bytes += bs;
                }
            }
        }
        launched += chunk;
        if (host.done_all) break;
    }
    if (info_out) {
        info_out[0] = (double)host.max_iter;
        info_out[1] = host.max_rel;          // negative: a segment stopped on r.z <= 0 with Jacobi too, or on NaN (hard failure)
        info_out[2] = (double)host.fallbacks; // segments whose coarse-level block lost definiteness: they went on with Jacobi alone
    }
    return NKSR_OK;
}

struct CsrOperator : PcgOperator {
    const int32_t* rowptr; const void* cols; const float* vals; int M; int64_t nnz; int fmt; SpmvPlan plan;
    int apply(const float* p, float* y, const int* done, const int*, int, const int*, int, hipStream_t st) override {
        launch_spmv(rowptr, cols, vals, M, nnz, fmt, plan, p, y, done, st);
        return NKSR_OK;
    }
    void bytes(double* a, double* ph, double* sv) override { spmv_bytes(M, nnz, fmt, a, ph); *sv = *a; }
};

extern "C" int nksr_pcg_solve(const int32_t* rowptr, const void* cols, const float* vals, const float* diag, int32_t M,
                              int64_t nnz, int col_format, const float* b, float* x, float tol, int max_iter, int check_every,
                              void* workspace, const nksr_coarse_precond_t* pc, double* info_out, void* stream) {
    if (M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; info_out[2] = 0; } return NKSR_OK; }
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    void* spmv_ws = (char*)workspace + pcg_vector_bytes(M);
    int rc = nksr_spmv_plan(rowptr, M, nnz, col_format, spmv_ws, stream);
    if (rc) return rc;
    CsrOperator A;
    A.rowptr = rowptr; A.cols = cols; A.vals = vals; A.M = M; A.nnz = nnz; A.fmt = col_format;
    A.plan = carve_spmv(spmv_ws, nnz, col_format);
    return nksr_pcg_run(A, diag, M, b, x, tol, max_iter, check_every, workspace, info_out, (hipStream_t)stream, pc);
}
