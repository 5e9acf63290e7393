// f(x) = sum_d sum_s alpha_j K_d(x, c_j) (and its gradient) at arbitrary positions: field.evaluate_f, the lattice samples of
// extract_dual_mesh (reference call sites models/loss.py:189-198, examples/recons_simple.py:27).  DESIGN.md section 3.2.
//
// One lane per query, the levels in sequence.  What the kernel is bound by is the LATENCY of its gathers, so they are issued in
// batches and branch-free: the home-slot hash probes of all levels first; per level the neighbour row of the containing cell (seven
// 16-byte loads), then the eight corner features together, then the 27 psi vectors nine at a time -- an absent neighbour reads voxel 0
// with weight 0 (adding 0 * x leaves a sum as skipping the term does, so the sums are those of the slot-by-slot loop).  Round 5's
// kernel took every neighbour through  index -> wait -> branch -> load -> wait -> fma  (35 dependent round trips per level, 34 000
// instructions with the hash fallback inlined at every one of them).  A query in a cell that is NOT active still sees every voxel
// whose support covers it: that path walks the same corners and slots through the hash in rolled loops (rare: the lattice samples of
// the mesher lie in active cells).
#include "kfield_dev.h"

template <int K, int H>
__global__ void __launch_bounds__(128) k_evaluate_f(nksr_hier_t hier, const float* __restrict__ alpha, const float* __restrict__ xyz,
                             int64_t n, float* __restrict__ fout, int active_only) {
    constexpr bool GRAD = false, JAC = false;
    float* gout = nullptr;
    const int L = hier.depth;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    float f = 0.f, gr[3] = {0.f, 0.f, 0.f};
    // the containing cell at EVERY level first: the home-slot probes of all levels go out together
    int cellv[NKSR_MAX_DEPTH];
    {
        int64_t key[NKSR_MAX_DEPTH], k0[NKSR_MAX_DEPTH];
        uint32_t slot[NKSR_MAX_DEPTH];
        int v0[NKSR_MAX_DEPTH];
#pragma unroll
        for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
            key[d] = 0; k0[d] = -1; slot[d] = 0; v0[d] = -1;
            if (d < L && hier.lv[d].n > 0) {               // uniform
                const SiteCell g = site_geometry(d, hier.inv_w0, x);
                key[d] = morton_biased(g.I[0], g.I[1], g.I[2], NKSR_BIAS0 >> d);
                slot[d] = hash_slot(key[d], hier.lv[d].hcap);
                k0[d] = hier.lv[d].hkeys[slot[d]];
                v0[d] = hier.lv[d].hvals[slot[d]];
            }
        }
#pragma unroll
        for (int d = 0; d < NKSR_MAX_DEPTH; ++d)
            cellv[d] = (d < L && hier.lv[d].n > 0) ? hash_find_after(hier.lv[d].hkeys, hier.lv[d].hvals, hier.lv[d].hcap, key[d], slot[d], k0[d], v0[d]) : -1;
    }
    constexpr int CB = K == 4 ? 8 : 2;                     // corner features / psi vectors requested together (registers: K floats each)
    constexpr int PB = K == 4 ? 9 : 3;
#pragma unroll 1
    for (int d = 0; d < L; ++d) {
        const nksr_level_t& lv = hier.lv[d];
        if (lv.n == 0) continue;
        SiteCell sc = site_geometry(d, hier.inv_w0, x);
        int cell = cellv[0];
#pragma unroll
        for (int q = 1; q < NKSR_MAX_DEPTH; ++q) cell = d == q ? cellv[q] : cell;
        sc.cell = cell;
        if (active_only && cell < 0) continue;             // the support of the kernel ROWS (training path: forward = what backward differentiates)
        const float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
        const float* __restrict__ featp = lv.feat;
        const float* __restrict__ psip = lv.psi;
        const float* __restrict__ alphap = alpha ? alpha + lv.offset : nullptr;
        float t[K], phi[K], Jt[JAC ? K : 1][3], J[JAC ? K : 1][3];
#pragma unroll
        for (int k = 0; k < K; ++k) { t[k] = 0.f; if (JAC) { Jt[k][0] = Jt[k][1] = Jt[k][2] = 0.f; } }
        float v[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = sc.u[a] + 0.5f - (float)sc.hb[a];
        int nb[27];
        if (cell >= 0) {
            load_nbr_row(lv.nbr + (int64_t)cell * 27, nb);
            int jc[8];
            jc[0] = corner_of_row<0, 0, 0>(nb, sc.hb); jc[1] = corner_of_row<0, 0, 1>(nb, sc.hb);
            jc[2] = corner_of_row<0, 1, 0>(nb, sc.hb); jc[3] = corner_of_row<0, 1, 1>(nb, sc.hb);
            jc[4] = corner_of_row<1, 0, 0>(nb, sc.hb); jc[5] = corner_of_row<1, 0, 1>(nb, sc.hb);
            jc[6] = corner_of_row<1, 1, 0>(nb, sc.hb); jc[7] = corner_of_row<1, 1, 1>(nb, sc.hb);
#pragma unroll
            for (int c0 = 0; c0 < 8; c0 += CB) {
                float fv[CB][K];
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    const float* fp = featp + (int64_t)(jc[c0 + q] >= 0 ? jc[c0 + q] : 0) * K;
#pragma unroll
                    for (int k = 0; k < K; ++k) fv[q][k] = fp[k];
                }
#pragma unroll
                for (int q = 0; q < CB; ++q) {             // (same corners, same order, same arithmetic as trilerp_feat)
                    const int c = c0 + q, cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
                    const bool have = jc[c] >= 0;
                    const float wx = cx ? v[0] : 1.f - v[0], wy = cy ? v[1] : 1.f - v[1], wz = cz ? v[2] : 1.f - v[2];
                    const float w = have ? wx * wy * wz : 0.f;
                    const float gx = have ? (cx ? 1.f : -1.f) * wy * wz * inv_w : 0.f, gy = have ? wx * (cy ? 1.f : -1.f) * wz * inv_w : 0.f,
                                gz = have ? wx * wy * (cz ? 1.f : -1.f) * inv_w : 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        t[k] = fmaf(fv[q][k], w, t[k]);
                        if (JAC) { Jt[k][0] = fmaf(fv[q][k], gx, Jt[k][0]); Jt[k][1] = fmaf(fv[q][k], gy, Jt[k][1]); Jt[k][2] = fmaf(fv[q][k], gz, Jt[k][2]); }
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int c = 0; c < 8; ++c) {
                const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
                const int j = hash_find(lv.hkeys, lv.hvals, lv.hcap,
                                        morton_biased(sc.I[0] + sc.hb[0] + cx - 1, sc.I[1] + sc.hb[1] + cy - 1, sc.I[2] + sc.hb[2] + cz - 1, NKSR_BIAS0 >> d));
                if (j < 0) continue;
                const float wx = cx ? v[0] : 1.f - v[0], wy = cy ? v[1] : 1.f - v[1], wz = cz ? v[2] : 1.f - v[2];
                const float w = wx * wy * wz;
                const float gx = (cx ? 1.f : -1.f) * wy * wz * inv_w, gy = wx * (cy ? 1.f : -1.f) * wz * inv_w, gz = wx * wy * (cz ? 1.f : -1.f) * inv_w;
                const float* fp = featp + (int64_t)j * K;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float fk = fp[k];
                    t[k] = fmaf(fk, w, t[k]);
                    if (JAC) { Jt[k][0] = fmaf(fk, gx, Jt[k][0]); Jt[k][1] = fmaf(fk, gy, Jt[k][1]); Jt[k][2] = fmaf(fk, gz, Jt[k][2]); }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MlpViewC<K, H> m(lv.mlp);                          // (interpolator weights through the scalar cache)
        mlp_residual<K, H, JAC>(m, t, Jt, phi, J);
        __builtin_amdgcn_sched_barrier(0);                     // (no psi load before the interpolator is through)
        float bw[3][3], bd[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) bspline3(sc.u[a], bw[a], bd[a]);
        float fl = 0.f, gl[3] = {0.f, 0.f, 0.f};
        // one neighbour's term, added in slot order (a == 0 for an absent neighbour of the batched path: the sums stay what they were)
        auto term = [&](const float* ps, float a, float bx, float by, float bz, float dx, float dy, float dz) {
            float dot = 0.f, jd[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                dot = fmaf(phi[k], ps[k], dot);
                if (JAC) { jd[0] = fmaf(J[k][0], ps[k], jd[0]); jd[1] = fmaf(J[k][1], ps[k], jd[1]); jd[2] = fmaf(J[k][2], ps[k], jd[2]); }
            }
            const float B = bx * by * bz;
            fl = fmaf(a, dot * B, fl);
            if (GRAD) {
                float g0 = dot * (dx * by * bz * inv_w), g1 = dot * (bx * dy * bz * inv_w), g2 = dot * (bx * by * dz * inv_w);
                if (JAC) { g0 = fmaf(jd[0], B, g0); g1 = fmaf(jd[1], B, g1); g2 = fmaf(jd[2], B, g2); }
                gl[0] = fmaf(a, g0, gl[0]); gl[1] = fmaf(a, g1, gl[1]); gl[2] = fmaf(a, g2, gl[2]);
            }
        };
        if (cell >= 0) {
#pragma unroll
            for (int s0 = 0; s0 < 27; s0 += PB) {
                float pv[PB][K], av[PB];
                // (the spline weights too are opaque until this batch: left alone, the compiler forms the 27 (x 4 with gradients)
                // weight products ahead of everything and holds them: 100 registers)
                asm volatile("" : "+v"(bw[0][0]), "+v"(bw[0][1]), "+v"(bw[0][2]), "+v"(bw[1][0]), "+v"(bw[1][1]), "+v"(bw[1][2]),
                                  "+v"(bw[2][0]), "+v"(bw[2][1]), "+v"(bw[2][2]) : "v"(fl));
                if (GRAD)
                    asm volatile("" : "+v"(bd[0][0]), "+v"(bd[0][1]), "+v"(bd[0][2]), "+v"(bd[1][0]), "+v"(bd[1][1]), "+v"(bd[1][2]),
                                      "+v"(bd[2][0]), "+v"(bd[2][1]), "+v"(bd[2][2]) : "v"(fl));
#pragma unroll
                for (int q = 0; q < PB; ++q) {
                    int j = nb[s0 + q] >= 0 ? nb[s0 + q] : 0;
                    // (the index waits for the sums so far -- phi at the first batch: the compiler otherwise requests all 27 vectors
                    // before the interpolator, 108 registers that leave two waves per SIMD)
                    asm volatile("" : "+v"(j) : "v"(fl), "v"(phi[0]));
                    const float* pp = psip + (int64_t)j * K;
#pragma unroll
                    for (int k = 0; k < K; ++k) pv[q][k] = pp[k];
                    av[q] = alphap ? alphap[j] : 1.f;      // alpha == NULL: psi arrives pre-multiplied by it
                }
#pragma unroll
                for (int q = 0; q < PB; ++q) {
                    const int s = s0 + q, ox = s / 9, oy = (s / 3) % 3, oz = s % 3;
                    term(pv[q], nb[s] >= 0 ? av[q] : 0.f, bw[0][ox], bw[1][oy], bw[2][oz], bd[0][ox], bd[1][oy], bd[2][oz]);
                }
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < 27; ++s) {
                const int ox = s / 9, oy = (s / 3) % 3, oz = s % 3;
                const int j = hash_find(lv.hkeys, lv.hvals, lv.hcap, morton_biased(sc.I[0] + ox - 1, sc.I[1] + oy - 1, sc.I[2] + oz - 1, NKSR_BIAS0 >> d));
                if (j < 0) continue;
                float pv[K];
#pragma unroll
                for (int k = 0; k < K; ++k) pv[k] = psip[(int64_t)j * K + k];
                term(pv, alphap ? alphap[j] : 1.f, sel3(bw[0], ox), sel3(bw[1], oy), sel3(bw[2], oz), sel3(bd[0], ox), sel3(bd[1], oy), sel3(bd[2], oz));
            }
        }
        f += fl;
        if (GRAD) { gr[0] += gl[0]; gr[1] += gl[1]; gr[2] += gl[2]; }
    }
    fout[i] = f;
    if (GRAD) { gout[i * 3] = gr[0]; gout[i * 3 + 1] = gr[1]; gout[i * 3 + 2] = gr[2]; }
}

// ---- value AND gradient: round 5's kernel (one neighbour at a time).  The batched form above needs 200+ registers once the three
// gradient sums ride along (the compiler forms the 4 x 27 spline-weight products ahead of the loads whatever the source order);
// evaluate_f(grad=True) is the training / normal-query path, not the mesher's.
template <int K, int H, bool GRAD, bool JAC>
__global__ void __launch_bounds__(128) k_evaluate_f_grad(nksr_hier_t hier, const float* __restrict__ alpha, const float* __restrict__ xyz,
                             int64_t n, float* __restrict__ fout, float* __restrict__ gout, int active_only) {
    extern __shared__ __attribute__((aligned(16))) float wall[];
    const int L = hier.depth;
    for (int d = 0; d < L; ++d)
        for (int i = threadIdx.x; i < MlpView<K, H>::SIZE; i += blockDim.x)
            wall[d * MlpView<K, H>::SIZE + i] = hier.lv[d].mlp[i];
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[3] = {xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]};
    float f = 0.f, gr[3] = {0.f, 0.f, 0.f};
    // the containing cell at EVERY level first: the home-slot probes of all levels go out together (one round trip instead of one per
    // level at the head of each level's chain  cell -> neighbour row -> features / psi)
    int cellv[NKSR_MAX_DEPTH];
    {
        int64_t key[NKSR_MAX_DEPTH], k0[NKSR_MAX_DEPTH];
        uint32_t slot[NKSR_MAX_DEPTH];
        int v0[NKSR_MAX_DEPTH];
#pragma unroll
        for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
            key[d] = 0; k0[d] = -1; slot[d] = 0; v0[d] = -1;
            if (d < L && hier.lv[d].n > 0) {               // uniform
                const SiteCell g = site_geometry(d, hier.inv_w0, x);
                key[d] = morton_biased(g.I[0], g.I[1], g.I[2], NKSR_BIAS0 >> d);
                slot[d] = hash_slot(key[d], hier.lv[d].hcap);
                k0[d] = hier.lv[d].hkeys[slot[d]];
                v0[d] = hier.lv[d].hvals[slot[d]];
            }
        }
#pragma unroll
        for (int d = 0; d < NKSR_MAX_DEPTH; ++d)
            cellv[d] = (d < L && hier.lv[d].n > 0) ? hash_find_after(hier.lv[d].hkeys, hier.lv[d].hvals, hier.lv[d].hcap, key[d], slot[d], k0[d], v0[d]) : -1;
    }
#pragma unroll
    for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
        if (d >= L) break;
        const nksr_level_t& lv = hier.lv[d];
        if (lv.n == 0) continue;
        SiteCell sc = site_geometry(d, hier.inv_w0, x);
        sc.cell = cellv[d];
        if (active_only && sc.cell < 0) continue;      // the support of the kernel ROWS (training path: forward = what backward differentiates)
        float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
        float t[K], phi[K], Jt[JAC ? K : 1][3], J[JAC ? K : 1][3];
        // (the corners are NOT taken from the neighbour row here as k_kernel_rows does: with the row live across the interpolator this
        // kernel needs 98 instead of 72 registers -- four waves per SIMD instead of six -- and ran 21 % slower, 1 240 against 1 027 us)
        trilerp_feat<K, JAC, true>(lv, d, sc, inv_w, t, Jt);
        MlpView<K, H> m(wall + d * MlpView<K, H>::SIZE);
        mlp_residual<K, H, JAC>(m, t, Jt, phi, J);
        float bw[3][3], bd[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) bspline3(sc.u[a], bw[a], bd[a]);
        float fl = 0.f, gl[3] = {0.f, 0.f, 0.f};
        int nbv[27];
        if (sc.cell >= 0) load_nbr_row(lv.nbr + (int64_t)sc.cell * 27, nbv);
        else {
#pragma unroll
            for (int s = 0; s < 27; ++s) nbv[s] = nbr_of<true>(lv, d, sc, s);
        }
#pragma unroll
        for (int s = 0; s < 27; ++s) {
            const int j = nbv[s];
            if (j < 0) continue;
            const int ox = s / 9, oy = (s / 3) % 3, oz = s % 3;
            const float* ps = lv.psi + (int64_t)j * K;
            float dot = 0.f, jd[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float pk = ps[k];
                dot = fmaf(phi[k], pk, dot);
                if (JAC) { jd[0] = fmaf(J[k][0], pk, jd[0]); jd[1] = fmaf(J[k][1], pk, jd[1]); jd[2] = fmaf(J[k][2], pk, jd[2]); }
            }
            const float a = alpha ? alpha[lv.offset + j] : 1.f;      // alpha == NULL: psi arrives pre-multiplied by it
            float bx = sel3(bw[0], ox), by = sel3(bw[1], oy), bz = sel3(bw[2], oz);
            float B = bx * by * bz;
            fl = fmaf(a, dot * B, fl);
            if (GRAD) {
                float g0 = dot * (sel3(bd[0], ox) * by * bz * inv_w), g1 = dot * (bx * sel3(bd[1], oy) * bz * inv_w),
                      g2 = dot * (bx * by * sel3(bd[2], oz) * inv_w);
                if (JAC) { g0 = fmaf(jd[0], B, g0); g1 = fmaf(jd[1], B, g1); g2 = fmaf(jd[2], B, g2); }
                gl[0] = fmaf(a, g0, gl[0]); gl[1] = fmaf(a, g1, gl[1]); gl[2] = fmaf(a, g2, gl[2]);
            }
        }
        f += fl;
        if (GRAD) { gr[0] += gl[0]; gr[1] += gl[1]; gr[2] += gl[2]; }
    }
    fout[i] = f;
    if (GRAD) { gout[i * 3] = gr[0]; gout[i * 3 + 1] = gr[1]; gout[i * 3 + 2] = gr[2]; }
}


extern "C" int nksr_evaluate_f(const nksr_hier_t* h, const float* alpha, const float* xyz, int64_t n, int approx, int active_only,
                               float* f_out, float* grad_out, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    dim3 grid(nksr_blocks(n, 128)), block(128);
    DISPATCH_KH(h->kdim, h->hidden, {
        const size_t lds = (size_t)h->depth * MlpView<K, H>::SIZE * sizeof(float);
        if (!grad_out) hipLaunchKernelGGL((k_evaluate_f<K, H>), grid, block, 0, (hipStream_t)stream, *h, alpha, xyz, n, f_out, active_only);
        else if (approx) hipLaunchKernelGGL((k_evaluate_f_grad<K, H, true, false>), grid, block, lds, (hipStream_t)stream, *h, alpha, xyz, n, f_out, grad_out, active_only);
        else hipLaunchKernelGGL((k_evaluate_f_grad<K, H, true, true>), grid, block, lds, (hipStream_t)stream, *h, alpha, xyz, n, f_out, grad_out, active_only);
    })
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}
