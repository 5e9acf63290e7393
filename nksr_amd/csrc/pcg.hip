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

struct PcgScalars {
    double rz[2];
    double bb;
    double rel;
    int iter;
    int done;
};

struct PcgWork {
    float *r, *z, *p, *y;
    double* part1;  // [PCG_MAX_BLOCKS]
    double* part2;  // [2*PCG_MAX_BLOCKS]
    PcgScalars* sc;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t pcg_vector_bytes(int32_t M) {
    size_t vec = align_up((size_t)M * sizeof(float), 256);
    return 4 * vec + 3 * PCG_MAX_BLOCKS * sizeof(double) + 256;
}
size_t nksr_pcg_vector_bytes(int32_t M) { return pcg_vector_bytes(M); }
extern "C" size_t nksr_spmv_workspace_bytes(int64_t nnz);
extern "C" size_t nksr_pcg_workspace_bytes(int32_t M, int64_t nnz) {
    return pcg_vector_bytes(M) + nksr_spmv_workspace_bytes(nnz);
}

static PcgWork carve(void* ws, int M) {
    PcgWork w;
    char* p = (char*)ws;
    size_t vec = align_up((size_t)M * sizeof(float), 256);
    w.r = (float*)p; p += vec;
    w.z = (float*)p; p += vec;
    w.p = (float*)p; p += vec;
    w.y = (float*)p; p += vec;
    w.part1 = (double*)p; p += PCG_MAX_BLOCKS * sizeof(double);
    w.part2 = (double*)p; p += 2 * PCG_MAX_BLOCKS * sizeof(double);
    w.sc = (PcgScalars*)p;
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
                const int4 ci = reinterpret_cast<const int4*>(cols_)[g];
                const float4 vi = reinterpret_cast<const float4*>(vals)[g];
                c[q][0] = ci.x; c[q][1] = ci.y; c[q][2] = ci.z; c[q][EPL - 1] = ci.w;
                v[q][0] = vi.x; v[q][1] = vi.y; v[q][2] = vi.z; v[q][EPL - 1] = vi.w;
            } else {
                const unsigned long long pk = reinterpret_cast<const unsigned long long*>(cols_)[g];
                const f32x3_u vi = reinterpret_cast<const f32x3_u*>(vals)[g];
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

__global__ void __launch_bounds__(PCG_BLOCK) k_pcg_init(int M, const float* __restrict__ b, const float* __restrict__ diag,
                                                        PcgWork w, float* __restrict__ x) {
    __shared__ double sm[PCG_BLOCK / 64];
    double bb = 0.0, rz = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        float bi = b[i];
        float zi = bi / diag[i];
        x[i] = 0.f;
        w.r[i] = bi;
        w.z[i] = zi;
        w.p[i] = zi;
        bb += (double)bi * bi;
        rz += (double)bi * zi;
    }
    double t0 = block_sum(bb, sm);
    double t1 = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        w.part2[2 * blockIdx.x] = t0;
        w.part2[2 * blockIdx.x + 1] = t1;
    }
}

__global__ void k_pcg_init_finish(PcgWork w, int nb) {
    __shared__ double sm[PCG_BLOCK / 64];
    double bb = reduce_partials(w.part2, nb, 2, sm);
    double rz = reduce_partials(w.part2 + 1, nb, 2, sm);
    if (threadIdx.x == 0) {
        w.sc->bb = bb;
        w.sc->rz[0] = rz;
        w.sc->rz[1] = 0.0;
        w.sc->rel = 1.0;
        w.sc->iter = 0;
        w.sc->done = (bb == 0.0) ? 1 : 0;
    }
}

// partial sums of p.Ap (fp64, fixed order).  Kept out of the SpMV on purpose: fused there, lane 0 of every
// wavefront waited for a dependent x[r] load once per row, which cost the SpMV ~10 % (761 vs 680 us).
__global__ void __launch_bounds__(PCG_BLOCK) k_pcg_dot(int M, PcgWork w) {
    if (w.sc->done) return;
    __shared__ double sm[PCG_BLOCK / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x)
        acc += (double)w.p[i] * (double)w.y[i];
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) w.part1[blockIdx.x] = t;
}

// x += alpha p ; r -= alpha y ; z = r / diag ; partial r.r and r.z
__global__ void __launch_bounds__(PCG_BLOCK) k_pcg_update(int M, const float* __restrict__ diag, PcgWork w,
                                                          float* __restrict__ x, int nb1, int parity) {
    if (w.sc->done) return;
    __shared__ double sm[PCG_BLOCK / 64];
    const double pAp = reduce_partials(w.part1, nb1, 1, sm);
    const float alpha = (float)(w.sc->rz[parity] / pAp);
    double rr = 0.0, rz = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        float pi = w.p[i], yi = w.y[i];
        x[i] = fmaf(alpha, pi, x[i]);
        float ri = fmaf(-alpha, yi, w.r[i]);
        float zi = ri / diag[i];
        w.r[i] = ri;
        w.z[i] = zi;
        rr += (double)ri * ri;
        rz += (double)ri * zi;
    }
    double t0 = block_sum(rr, sm);
    double t1 = block_sum(rz, sm);
    if (threadIdx.x == 0) {
        w.part2[2 * blockIdx.x] = t0;
        w.part2[2 * blockIdx.x + 1] = t1;
    }
}

// p = z + beta p ; block 0 publishes the scalars of the finished iteration
__global__ void __launch_bounds__(PCG_BLOCK) k_pcg_pupdate(int M, PcgWork w, int nb2, int parity, float tol) {
    if (w.sc->done) return;
    __shared__ double sm[PCG_BLOCK / 64];
    const double rr = reduce_partials(w.part2, nb2, 2, sm);
    const double rz_new = reduce_partials(w.part2 + 1, nb2, 2, sm);
    const double rz_old = w.sc->rz[parity];
    const double bb = w.sc->bb;
    const float beta = (float)(rz_new / rz_old);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x)
        w.p[i] = fmaf(beta, w.p[i], w.z[i]);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double rel = sqrt(rr / bb);
        w.sc->rz[parity ^ 1] = rz_new;
        w.sc->rel = rel;
        w.sc->iter += 1;
        if (rel <= (double)tol) w.sc->done = 1;
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
    if (fmt == 1) { if (g_spmv_variant == 1) SPMV_LAUNCH(3, 1); else SPMV_LAUNCH(3, 0); }
    else { if (g_spmv_variant == 1) SPMV_LAUNCH(4, 1); else SPMV_LAUNCH(4, 0); }
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
static double g_prof_alg_bytes = 0.0, g_prof_phys_bytes = 0.0;
// several host threads may run solves at once (chunks on separate streams): every thread times with its own event pool and
// adds to the shared accumulators under a lock
static thread_local std::vector<hipEvent_t> g_prof_events;
static std::mutex g_prof_mutex;

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
    g_prof_alg_bytes = g_prof_phys_bytes = 0.0;
    return NKSR_OK;
}

extern "C" int nksr_pcg_profile(int enable, double* ms_out, int64_t* launches_out) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (ms_out) *ms_out = g_prof_ms;
    if (launches_out) *launches_out = g_prof_launches;
    g_prof_ms = 0.0;
    g_prof_launches = 0;
    g_prof_enable = enable;
    return NKSR_OK;
}

// ---- coarse-level block preconditioner -------------------------------------------------------------------------------------------
// The multi-level system is badly conditioned through its COARSE basis functions (each overlaps 124 neighbours of its own level and
// every finer voxel under its support): Jacobi needs 47 iterations per tree_depth-5 chunk where an exact solve of the diagonal
// block of the levels >= 2 (8 % of the unknowns, 4.5 % of the non-zeros) would need 12.  That block A_cc is assembled once per
// solve (plain CSR, nksr_assemble on the hierarchy with its fine levels masked) and z_c ~ A_cc^-1 r_c is approximated by a FIXED
// number of Jacobi-preconditioned Chebyshev steps -- a fixed polynomial in A_cc, hence a constant SPD preconditioner (plain CG
// stays valid) and deterministic.  One kernel per step, one wavefront per row:
//   t = (A_cc d)_j;  y_j += d_j;  res_j -= t;  d'_j = a d_j + b res_j / D_j          (d double-buffered: rows read their neighbours' d)
__global__ void __launch_bounds__(256) k_cheb_init(int n, const float* __restrict__ r, const float* __restrict__ diag, float inv_theta,
                                                   float* __restrict__ res, float* __restrict__ d0, float* __restrict__ y,
                                                   const int* __restrict__ done) {
    if (done && *done) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float rj = r[j];
    res[j] = rj;
    d0[j] = rj / diag[j] * inv_theta;
    y[j] = 0.f;
}

__global__ void __launch_bounds__(256) k_cheb_step(int n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                   const float* __restrict__ vals, const float* __restrict__ diag, float a, float b,
                                                   float* __restrict__ res, const float* __restrict__ d_old, float* __restrict__ d_new,
                                                   float* __restrict__ y, const int* __restrict__ done) {
    if (done && *done) return;
    const int j = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (j >= n) return;
    // a wavefront's time is a chain of dependent load latencies (the block streams from L2 / HBM): everything that does not depend on
    // the row's entries is requested up front, and four 64-entry groups (rows hold ~190 entries) are in flight at once -- clamped
    // addresses instead of predicated loads, so that the compiler does not serialise them
    const int k0 = rowptr[j], k1 = rowptr[j + 1];
    const float dj = d_old[j], rs = res[j], dg = diag[j], yj = y[j];
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
// out = sqrt(sum b^2 / sum a^2), one workgroup, fixed order
__global__ void __launch_bounds__(PCG_BLOCK) k_norm_ratio(int n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    __shared__ double sm[PCG_BLOCK / 64];
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < n; i += PCG_BLOCK) { sa += (double)a[i] * a[i]; sb += (double)b[i] * b[i]; }
    const double ta = block_sum(sa, sm), tb = block_sum(sb, sm);
    if (threadIdx.x == 0) out[0] = ta > 0.0 ? (float)sqrt(tb / ta) : 0.f;
}

extern "C" int nksr_coarse_lambda_max(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n, int iters,
                                      float* work, float* lambda_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!rowptr || !cols || !vals || !diag || !work || !lambda_out) return nksr_set_error(NKSR_ERR_ARG, "NULL arrays");
    if (iters < 2) iters = 2;
    hipStream_t st = (hipStream_t)stream;
    float* v[2] = {work, work + n};
    const dim3 g1(nksr_blocks(n, 256)), gw(nksr_blocks((int64_t)n * 64, 256));
    hipLaunchKernelGGL(k_cheb_init, g1, dim3(256), 0, st, n, diag, diag, 1.f, v[1], v[0], v[1], (const int*)nullptr);   // v0 = 1 (diag / diag)
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_coarse_power, gw, dim3(256), 0, st, n, rowptr, cols, vals, diag, (const float*)v[i & 1], v[(i + 1) & 1]);
    hipLaunchKernelGGL(k_norm_ratio, dim3(1), dim3(PCG_BLOCK), 0, st, n, (const float*)v[(iters - 1) & 1], (const float*)v[iters & 1], lambda_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

struct ChebPlan { int steps; float inv_theta; float a[NKSR_PC_MAX_STEPS], b[NKSR_PC_MAX_STEPS]; };
static int cheb_plan(ChebPlan& P, const nksr_coarse_precond_t* pc) {
    if (pc->n <= 0 || pc->steps < 1 || pc->steps > NKSR_PC_MAX_STEPS) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: 1..%d steps", NKSR_PC_MAX_STEPS);
    if (!(pc->lambda_max > 0.f) || !(pc->ratio > 1.f)) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: lambda_max > 0, ratio > 1");
    if (!pc->rowptr || !pc->cols || !pc->vals || !pc->diag || !pc->work) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner has NULL arrays");
    const double lmax = pc->lambda_max, lmin = lmax / pc->ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
    const double sigma = theta / delta;
    double rho = 1.0 / sigma;
    P.steps = pc->steps;
    P.inv_theta = (float)(1.0 / theta);
    for (int i = 0; i < pc->steps; ++i) {
        const double rho_n = 1.0 / (2.0 * sigma - rho);
        P.a[i] = (float)(rho_n * rho);
        P.b[i] = (float)(2.0 * rho_n / delta);
        rho = rho_n;
    }
    return NKSR_OK;
}
// z_c = p_k(A_cc) r_c  (r, z: the coarse slices of the PCG vectors)
static void cheb_apply(const nksr_coarse_precond_t* pc, const ChebPlan& P, const float* r, float* z, const int* done, hipStream_t st) {
    const int n = pc->n;
    float *res = pc->work, *d[2] = {pc->work + n, pc->work + 2 * (size_t)n};
    hipLaunchKernelGGL(k_cheb_init, dim3(nksr_blocks(n, 256)), dim3(256), 0, st, n, r, pc->diag, P.inv_theta, res, d[0], z, done);
    for (int i = 0; i < P.steps; ++i)
        hipLaunchKernelGGL(k_cheb_step, dim3(nksr_blocks((int64_t)n * 64, 256)), dim3(256), 0, st, n, pc->rowptr, pc->cols, pc->vals, pc->diag,
                           P.a[i], P.b[i], res, (const float*)d[i & 1], d[(i + 1) & 1], z, done);
}

// partial r.z again (second column of part2) after the coarse slice of z changed; init: p = z as well
__global__ void __launch_bounds__(PCG_BLOCK) k_pcg_rz(int M, PcgWork w, int copy_p) {
    if (!copy_p && w.sc->done) return;
    __shared__ double sm[PCG_BLOCK / 64];
    double rz = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const float zi = w.z[i];
        if (copy_p) w.p[i] = zi;
        rz += (double)w.r[i] * zi;
    }
    const double t = block_sum(rz, sm);
    if (threadIdx.x == 0) w.part2[2 * blockIdx.x + 1] = t;
}

int nksr_pcg_run(PcgOperator& A, const float* diag, int32_t M, const float* b, float* x, float tol, int max_iter, int check_every,
                 void* vector_workspace, double* info_out, hipStream_t st, const nksr_coarse_precond_t* pc) {
    if (M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!vector_workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    if (check_every < 1) check_every = 1;
    PcgWork w = carve(vector_workspace, M);
    const int nbv = nksr_blocks(M, PCG_BLOCK) > PCG_MAX_BLOCKS ? PCG_MAX_BLOCKS : nksr_blocks(M, PCG_BLOCK);
    ChebPlan plan;
    if (pc) {
        if (int rc = cheb_plan(plan, pc)) return rc;
        if (pc->first < 0 || pc->first + pc->n != M) return nksr_set_error(NKSR_ERR_ARG, "coarse preconditioner: the block must be the last %d unknowns", pc->n);
    }
    hipLaunchKernelGGL(k_pcg_init, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, b, diag, w, x);
    if (pc) {
        cheb_apply(pc, plan, w.r + pc->first, w.z + pc->first, nullptr, st);
        hipLaunchKernelGGL(k_pcg_rz, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, w, 1);
    }
    hipLaunchKernelGGL(k_pcg_init_finish, dim3(1), dim3(PCG_BLOCK), 0, st, w, nbv);
    NKSR_CHECK_LAUNCH();
    PcgScalars host;
    memset(&host, 0, sizeof(host));
    int launched = 0;
    const bool prof = g_prof_enable != 0;
    if (prof)
        while ((int)g_prof_events.size() < 2 * check_every) {
            hipEvent_t e;
            NKSR_CHECK_HIP(hipEventCreate(&e));
            g_prof_events.push_back(e);
        }
    while (launched < max_iter) {
        int chunk = check_every < (max_iter - launched) ? check_every : (max_iter - launched);
        for (int c = 0; c < chunk; ++c) {
            const int parity = (launched + c) & 1;
            if (prof) (void)hipEventRecord(g_prof_events[2 * c], st);
            if (int rc = A.apply(w.p, w.y, &w.sc->done, st)) return rc;
            if (prof) (void)hipEventRecord(g_prof_events[2 * c + 1], st);
            hipLaunchKernelGGL(k_pcg_dot, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, w);
            hipLaunchKernelGGL(k_pcg_update, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, diag, w, x, nbv, parity);
            if (pc) {
                cheb_apply(pc, plan, w.r + pc->first, w.z + pc->first, &w.sc->done, st);
                hipLaunchKernelGGL(k_pcg_rz, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, w, 0);
            }
            hipLaunchKernelGGL(k_pcg_pupdate, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, w, nbv, parity, tol);
        }
        NKSR_CHECK_LAUNCH();
        NKSR_CHECK_HIP(hipMemcpyAsync(&host, w.sc, sizeof(host), hipMemcpyDeviceToHost, st));
        NKSR_CHECK_HIP(hipStreamSynchronize(st));
        if (prof) {
            // only applications that did real work (the done flag turns later ones into no-ops)
            double ba, bp;
            A.bytes(&ba, &bp);
            std::lock_guard<std::mutex> lock(g_prof_mutex);
            for (int c = 0; c < chunk && launched + c < host.iter; ++c) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g_prof_events[2 * c], g_prof_events[2 * c + 1]) == hipSuccess) {
                    g_prof_ms += ms;
                    g_prof_launches += 1;
                    g_prof_alg_bytes += ba;
                    g_prof_phys_bytes += bp;
                }
            }
        }
        launched += chunk;
        if (host.done) break;
    }
    if (info_out) {
        info_out[0] = (double)host.iter;
        info_out[1] = host.rel;
    }
    return NKSR_OK;
}

struct CsrOperator : PcgOperator {
    const int32_t* rowptr; const void* cols; const float* vals; int M; int64_t nnz; int fmt; SpmvPlan plan;
    int apply(const float* p, float* y, const int* done, hipStream_t st) override {
        launch_spmv(rowptr, cols, vals, M, nnz, fmt, plan, p, y, done, st);
        return NKSR_OK;
    }
    void bytes(double* a, double* ph) override { spmv_bytes(M, nnz, fmt, a, ph); }
};

extern "C" int nksr_pcg_solve(const int32_t* rowptr, const void* cols, const float* vals, const float* diag, int32_t M,
                              int64_t nnz, int col_format, const float* b, float* x, float tol, int max_iter, int check_every,
                              void* workspace, double* info_out, void* stream) {
    if (M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    void* spmv_ws = (char*)workspace + pcg_vector_bytes(M);
    int rc = nksr_spmv_plan(rowptr, M, nnz, col_format, spmv_ws, stream);
    if (rc) return rc;
    CsrOperator A;
    A.rowptr = rowptr; A.cols = cols; A.vals = vals; A.M = M; A.nnz = nnz; A.fmt = col_format;
    A.plan = carve_spmv(spmv_ws, nnz, col_format);
    return nksr_pcg_run(A, diag, M, b, x, tol, max_iter, check_every, workspace, info_out, (hipStream_t)stream);
}
