// Jacobi-preconditioned conjugate gradients on the CSR normal equations.
// The CSR SpMV is the roofline kernel of this project:
//   algorithmic bytes per launch  B_spmv = 8*nnz + 12*M + 4   (fp32 vals, int32 cols/rowptr)
// Per iteration: (1) y = A p fused with the partial dot p.y, (2) x,r,z update fused with the
// partial dots r.r and r.z, (3) p update.  alpha/beta never leave the device; dot products are
// accumulated in fp64 with a fixed reduction order (deterministic).  A device-side `done` flag
// turns the remaining launches of a chunk into no-ops, so the host only syncs every
// `check_every` iterations.
#include "common.h"

#define PCG_BLOCK 256
#define PCG_MAX_BLOCKS 1024

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

extern "C" size_t nksr_pcg_workspace_bytes(int32_t M) {
    size_t vec = align_up((size_t)M * sizeof(float), 256);
    return 4 * vec + 3 * PCG_MAX_BLOCKS * sizeof(double) + 256;
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

// ---- SpMV: one wavefront per row, lanes stride the row's (col,val) stream -------------------
template <bool DOT>
__global__ void __launch_bounds__(PCG_BLOCK) k_spmv(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                    const float* __restrict__ vals, int M, const float* __restrict__ x,
                                                    float* __restrict__ y, double* __restrict__ part,
                                                    const int* __restrict__ done) {
    if (DOT && *done) return;
    __shared__ double sm[PCG_BLOCK / 64];
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * PCG_BLOCK + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PCG_BLOCK) >> 6;
    double dot = 0.0;
    for (int row = wave_global; row < M; row += nwaves) {
        const int k0 = rowptr[row], k1 = rowptr[row + 1];
        float acc = 0.f;
        for (int k = k0 + lane; k < k1; k += 64) acc = fmaf(vals[k], x[cols[k]], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            y[row] = acc;
            if (DOT) dot += (double)acc * (double)x[row];
        }
    }
    if (DOT) {
        double t = block_sum(dot, sm);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
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

static int spmv_grid(int M) {
    int rows_per_block = PCG_BLOCK / 64;
    int nb = (M + rows_per_block - 1) / rows_per_block;
    return nb < 1 ? 1 : (nb > PCG_MAX_BLOCKS ? PCG_MAX_BLOCKS : nb);
}

extern "C" int nksr_spmv_csr(const int32_t* rowptr, const int32_t* cols, const float* vals, int32_t M, const float* x,
                             float* y, void* stream) {
    if (M <= 0) return NKSR_OK;
    hipLaunchKernelGGL((k_spmv<false>), dim3(spmv_grid(M)), dim3(PCG_BLOCK), 0, (hipStream_t)stream, rowptr, cols, vals,
                       M, x, y, (double*)nullptr, (const int*)nullptr);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

// ---- optional live profiling of the SpMV launches (bench.py's roofline leg) ---------------------
#include <vector>
static int g_prof_enable = 0;
static double g_prof_ms = 0.0;
static long long g_prof_launches = 0;
static std::vector<hipEvent_t> g_prof_events;

extern "C" int nksr_pcg_profile(int enable, double* ms_out, int64_t* launches_out) {
    if (ms_out) *ms_out = g_prof_ms;
    if (launches_out) *launches_out = g_prof_launches;
    g_prof_ms = 0.0;
    g_prof_launches = 0;
    g_prof_enable = enable;
    return NKSR_OK;
}

extern "C" int nksr_pcg_solve(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t M,
                              const float* b, float* x, float tol, int max_iter, int check_every, void* workspace,
                              double* info_out, void* stream) {
    if (M <= 0) { if (info_out) { info_out[0] = 0; info_out[1] = 0; } return NKSR_OK; }
    if (!workspace) return nksr_set_error(NKSR_ERR_ARG, "workspace is NULL");
    if (check_every < 1) check_every = 1;
    hipStream_t st = (hipStream_t)stream;
    PcgWork w = carve(workspace, M);
    const int nbv = nksr_blocks(M, PCG_BLOCK) > PCG_MAX_BLOCKS ? PCG_MAX_BLOCKS : nksr_blocks(M, PCG_BLOCK);
    const int nbs = spmv_grid(M);
    hipLaunchKernelGGL(k_pcg_init, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, b, diag, w, x);
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
            if (prof) hipEventRecord(g_prof_events[2 * c], st);
            hipLaunchKernelGGL((k_spmv<true>), dim3(nbs), dim3(PCG_BLOCK), 0, st, rowptr, cols, vals, M, w.p, w.y, w.part1,
                               &w.sc->done);
            if (prof) hipEventRecord(g_prof_events[2 * c + 1], st);
            hipLaunchKernelGGL(k_pcg_update, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, diag, w, x, nbs, parity);
            hipLaunchKernelGGL(k_pcg_pupdate, dim3(nbv), dim3(PCG_BLOCK), 0, st, M, w, nbv, parity, tol);
        }
        NKSR_CHECK_LAUNCH();
        NKSR_CHECK_HIP(hipMemcpyAsync(&host, w.sc, sizeof(host), hipMemcpyDeviceToHost, st));
        NKSR_CHECK_HIP(hipStreamSynchronize(st));
        if (prof) {
            // only launches that did real work (the done flag turns later ones into no-ops)
            for (int c = 0; c < chunk && launched + c < host.iter; ++c) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g_prof_events[2 * c], g_prof_events[2 * c + 1]) == hipSuccess) {
                    g_prof_ms += ms;
                    g_prof_launches += 1;
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
