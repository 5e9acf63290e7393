// Kernel rows of the matrix-free operator, both site sets in one pass (KernelField.solve with fused_mode, models/nksr_net.py:105-112,
// examples/recons_waymo.py:33).  DESIGN.md section 3.2.
#include "kfield_dev.h"
#include <stdlib.h>

// ---- the rows of the matrix-free operator in ONE pass over its merged row list (kernel_dim 4): one lane per ROW ---------------------
// rows_all of the operator interleaves the two site sets in Morton order: [3 gradient rows of a normal site][the position rows of
// the points in the same cell] ...  Written by one launch per set (k_kernel_rows with a row_index), every 128-byte line of the
// array is written TWICE, partially, by two kernels seconds of traffic apart: the partial lines go to HBM as masked /
// read-modify-write bursts -- 1.1 TB/s for the position rows and 1.8 TB/s for the gradient rows of the 64-chunk scene, whatever
// the shape of the stores (tools/probes/store_probe.hip), against 5.7 TB/s for the same bytes written as contiguous images.
// Here lane = row of the list, wavefront = 64 consecutive rows:
//   * row_src[r] = (site << 2) | kind names the row's site (kind 0: value row of a position site, 1 + a: d/dx_a row of a normal
//     site; < 0: a pad row: zeros).  The three rows of a normal site are three neighbouring lanes: their gathers coalesce.
//   * a lane carries the VALUE channel of the interpolator and ONE tangent channel (its own axis) as a pair: every multiply-add
//     of trilinear stencil, interpolator and psi products is one v_pk_fma_f32 (value, tangent) -- each half an IEEE fma in the
//     order of k_kernel_rows: the rows are bit-identical to its rows.
//   * the 27 results go to a wave-private LDS image (stride 27 words: conflict-free; the image first holds the lanes' neighbour rows)
//     and leave as ONE contiguous 6 912-byte run per wavefront, 16 bytes per lane and instruction.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f pk_fma(float w, v2f b, v2f c) { const v2f ww = {w, w}; return __builtin_elementwise_fma(ww, b, c); }
// c + a.x * b / c + a.y * b, both halves: the broadcast is the instruction's op_sel (left to the compiler, the high half of the second
// pair of a 16-byte load is first moved into a fresh register pair: a third of the vector instructions of the slot loop were moves)
__device__ __forceinline__ v2f pk_fma_lo(v2f a, v2f b, v2f c) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ v2f pk_fma_hi(v2f a, v2f b, v2f c) {
    v2f d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// sum_k a[k] * b[k] + c over the four entries of a 16-byte vector, in order
__device__ __forceinline__ v2f pk_dot4(v4f a, const v2f b[4], v2f c) {
    const v2f a01 = {a.x, a.y}, a23 = {a.z, a.w};
    c = pk_fma_lo(a01, b[0], c); c = pk_fma_hi(a01, b[1], c); c = pk_fma_lo(a23, b[2], c); c = pk_fma_hi(a23, b[3], c);
    return c;
}

// CMP: COMPACT rows (nksr_fused_op_t.compact, csrc/fused.hip): a row keeps the slots of its cell's existing neighbours only; the cell
// comes from row_cells (made before the cells' places in the array could be laid out: nksr_row_cells_merged) and cmp = the operator's
// nbr32 table tells where the cell's rows lie.  The words of a wavefront's 64 rows are still ONE contiguous run of the array (cells
// lie in row order), now of variable length and starting at any word: it is staged in LDS at the same offset mod 4 it has in
// memory, so that 16-byte pieces of the image are 16-byte pieces of the array.
// PRE: the rows' cells are given (row_cells, nksr_row_cells_merged) instead of looked up: the look-up heads the kernel's chain of
// dependent loads (row source -> position -> hash probe -> neighbour row -> features); given, the cell is one coalesced load that
// does not wait for the position (CMP implies PRE).
template <int H, bool JAC, bool CMP, bool PRE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(H == 16 ? 4 : 2))) k_kernel_rows_merged(nksr_hier_t hier, const float* __restrict__ xyz_a, const float* __restrict__ ss_a, float rs_a,
                              const float* __restrict__ xyz_b, const float* __restrict__ ss_b, float rs_b, const int32_t* __restrict__ row_src,
                              int64_t rows_total, int32_t* __restrict__ row_cells, const int32_t* __restrict__ cmp, float* __restrict__ rows_out, uint32_t level_map) {
    constexpr int K = 4;
    const int d = (level_map >> (4 * blockIdx.y)) & 15;
    const nksr_level_t& lv = hier.lv[d];
    constexpr int IMG = CMP ? 64 * 30 + 8 : 64 * 27;      // (compact: up to three pad words behind every row that ends a cell, + the shift)
    __shared__ __attribute__((aligned(16))) float img[4][IMG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t R0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
    if (R0 >= rows_total) return;
    float* im = img[wv] + 27 * lane;                           // the lane's 27 words: its neighbour row waits here, then (dense) its row
    const int64_t R = R0 + lane;
    const int src = R < rows_total ? row_src[R] : -1;
    const int cj_given = (PRE && R < rows_total) ? row_cells[(int64_t)d * rows_total + R] : -1;
    const int kind = src & 3, site = src >> 2;
    const int ax = kind - 1;                                   // the lane's gradient axis (-1: a value row)
    int cell = -1;
    SiteCell sc;
    float scale = 0.f;
    // compact: the lane's words of the run -- image offset io, k of them (+ pad zeros behind a cell's last row), mask of its cell
    int io = 0, kc = 0, padz = 0, shift = 0, nrun = 0;
    unsigned cmask = 0;
    int64_t G = 0;
    if (src >= 0) {
        const float* xp = (kind == 0 ? xyz_a : xyz_b) + (int64_t)site * 3;
        const float x[3] = {xp[0], xp[1], xp[2]};
        scale = kind == 0 ? (ss_a ? ss_a[site] : rs_a) : (ss_b ? ss_b[site] : rs_b);
        if (PRE) {
            sc = site_geometry(d, hier.inv_w0, x);
            cell = cj_given >= 0 ? cj_given - lv.offset : -1;
            sc.cell = cell;
        } else {
            sc = locate_site(lv, d, hier.inv_w0, x);
            cell = sc.cell;
        }
    }
    if (!PRE && row_cells && R < rows_total) row_cells[(int64_t)d * rows_total + R] = cell >= 0 ? lv.offset + cell : -1;
    if (CMP) {
        int64_t g = 0;
        int nw = 0;
        if (cell >= 0) {
            const int4 tb = *reinterpret_cast<const int4*>(cmp + ((int64_t)(lv.offset + cell) * 32 + 28));      // first row, last row, first word / 4, mask
            cmask = (unsigned)tb.w;
            kc = __popc(cmask);
            g = (int64_t)tb.z * 4 + (int64_t)((int)R - tb.x) * kc;
            padz = (int)R == tb.y ? (int)((-(int64_t)(tb.y - tb.x + 1) * kc) & 3) : 0;
            nw = kc + padz;
        }
        const unsigned long long bal = __ballot(nw > 0);
        if (bal == 0ull) return;                               // (no words at this level: rows without a cell, pad rows)
        const int fl = __builtin_ctzll(bal), ll = 63 - __builtin_clzll(bal);
        const int glo = (int)(g & 0xffffffffll), ghi = (int)(g >> 32);
        G = ((int64_t)__shfl(ghi, fl) << 32) | (unsigned)__shfl(glo, fl);
        const int64_t gl = ((int64_t)__shfl(ghi, ll) << 32) | (unsigned)__shfl(glo, ll);
        shift = (int)(G & 3);
        nrun = (int)(gl - G) + __shfl(nw, ll);
        io = nw > 0 ? (int)(g - G) + shift : 0;
    }
    if (cell < 0) {
        if (!CMP) {
#pragma unroll
            for (int q = 0; q < 27; ++q) im[q] = 0.f;
        }
    } else {
        const float inv_w = hier.inv_w0 * __int_as_float((127 - d) << 23);
        // a value row's lane carries NO tangent (factor 0 at its source) and scales by 1 where a gradient row scales by 1 / w:
        // one instruction stream for both kinds of rows, no selects
        const float iw_t = ax < 0 ? 0.f : inv_w, iw_r = ax < 0 ? 1.f : inv_w;
        const f32x4_u* __restrict__ feat4 = reinterpret_cast<const f32x4_u*>(lv.feat);
        const f32x4_u* __restrict__ psi4 = reinterpret_cast<const f32x4_u*>(lv.psi);
        v2f tj[K];                                            // (t_k, d t_k / d x_ax)
        {
            // the neighbour row waits in the lane's part of the LDS image (27 registers less across the interpolator); the slot loop
            // takes it back nine words at a time, right before it overwrites them with its results
            int nb[27];
            load_nbr_row(lv.nbr + (int64_t)cell * 27, nb);
#pragma unroll
            for (int q = 0; q < 27; ++q) im[q] = __int_as_float(nb[q]);
            int jc[8];
            jc[0] = corner_of_row<0, 0, 0>(nb, sc.hb); jc[1] = corner_of_row<0, 0, 1>(nb, sc.hb);
            jc[2] = corner_of_row<0, 1, 0>(nb, sc.hb); jc[3] = corner_of_row<0, 1, 1>(nb, sc.hb);
            jc[4] = corner_of_row<1, 0, 0>(nb, sc.hb); jc[5] = corner_of_row<1, 0, 1>(nb, sc.hb);
            jc[6] = corner_of_row<1, 1, 0>(nb, sc.hb); jc[7] = corner_of_row<1, 1, 1>(nb, sc.hb);
            // the eight corner features go out together, branch-free: a missing corner reads voxel 0 with weight 0 (adding f * 0 leaves
            // the sums as skipping the corner does)
            f32x4_u fc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) fc[c] = feat4[jc[c] >= 0 ? jc[c] : 0];
            float v[3], W[3][2], P[3][2];                     // trilinear weights of the two corners per axis; P: the tangent's factors
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v[a] = sc.u[a] + 0.5f - (float)sc.hb[a];
                W[a][0] = 1.f - v[a]; W[a][1] = v[a];
                P[a][0] = ax == a ? -1.f : W[a][0];
                P[a][1] = ax == a ? 1.f : W[a][1];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) tj[k] = v2f{0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) {                     // (same corners, same order, same arithmetic as trilerp_feat_row)
                const int cx = c >> 2, cy = (c >> 1) & 1, cz = c & 1;
                const bool have = jc[c] >= 0;
                const v2f wg = {have ? W[0][cx] * W[1][cy] * W[2][cz] : 0.f, have ? P[0][cx] * P[1][cy] * P[2][cz] * iw_t : 0.f};
                const v2f f01 = {fc[c].x, fc[c].y}, f23 = {fc[c].z, fc[c].w};
                tj[0] = pk_fma_lo(f01, wg, tj[0]); tj[1] = pk_fma_hi(f01, wg, tj[1]); tj[2] = pk_fma_lo(f23, wg, tj[2]); tj[3] = pk_fma_hi(f23, wg, tj[3]);
            }
        }
        // phi = t + MLP(t) with ONE tangent channel (mlp_residual's value and d/dx_ax arithmetic, as pairs).  The weights are the same
        // for every lane: read through the scalar cache into SGPRs (constant address space, uniform addresses) -- as LDS reads
        // they cost a ds_read per four weights and a register move per odd one
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat* wc = (cfloat*)(uintptr_t)lv.mlp;
        cfloat* W1 = wc; cfloat* b1 = W1 + H * K; cfloat* W2 = b1 + H; cfloat* b2 = W2 + H * H; cfloat* W3 = b2 + H; cfloat* b3 = W3 + K * H;
        v2f out[K];
        {
            v2f h1[H];
            const v2f zero2 = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < H; ++h) {
                v2f acc = {b1[h], 0.f};
#pragma unroll
                for (int k = 0; k < K; ++k) acc = pk_fma(W1[h * K + k], tj[k], acc);
                h1[h] = acc.x > 0.f ? acc : zero2;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) out[k] = v2f{tj[k].x + b3[k], tj[k].y};
#pragma unroll 1
            for (int g = 0; g < H; ++g) {
                v2f acc = {b2[g], 0.f};
#pragma unroll
                for (int h = 0; h < H; ++h) acc = pk_fma(W2[g * H + h], h1[h], acc);
                const v2f h2 = acc.x > 0.f ? acc : zero2;
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = pk_fma(W3[k * H + g], h2, out[k]);
            }
        }
        float bw[3][3], bd[3][3], C[3][3];                     // C: the row's factors (d/du on a gradient row's own axis)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bspline3(sc.u[a], bw[a], bd[a]);
#pragma unroll
            for (int o = 0; o < 3; ++o) C[a][o] = ax == a ? bd[a][o] : bw[a][o];
        }
        // the 27 psi vectors in three batches of nine, each batch's loads issued together (absent neighbour: voxel 0, result 0)
        if (!CMP) {
#pragma unroll
            for (int q0 = 0; q0 < 27; q0 += 9) {
                f32x4_u ps[9];
                int nb[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) nb[i] = __float_as_int(im[q0 + i]);
#pragma unroll
                for (int i = 0; i < 9; ++i) ps[i] = psi4[nb[i] >= 0 ? nb[i] : 0];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const int q = q0 + i;
                    const int ox = q / 9, oy = (q / 3) % 3, oz = q % 3;
                    const v4f p4 = {ps[i].x, ps[i].y, ps[i].z, ps[i].w};
                    const v2f dj = pk_dot4(p4, out, v2f{0.f, 0.f});   // (<phi, psi>, <d phi / d x_ax, psi>)
                    float r = dj.x * (C[0][ox] * C[1][oy] * C[2][oz] * iw_r);
                    if (JAC) r = fmaf(dj.y, bw[0][ox] * bw[1][oy] * bw[2][oz], r);
                    im[q] = (nb[i] >= 0 ? r : 0.f) * scale;
                }
            }
        }
        if (CMP) {
            // every lane takes its neighbour row back first: the image is about to be overwritten, at offsets that are other lanes'
            int nbq[27];
#pragma unroll
            for (int q = 0; q < 27; ++q) nbq[q] = __float_as_int(im[q]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float* dst = img[wv] + io;                         // the lane's k words, slots of existing neighbours in slot order
#pragma unroll
            for (int q0 = 0; q0 < 27; q0 += 9) {
                f32x4_u ps[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) ps[i] = psi4[nbq[q0 + i] >= 0 ? nbq[q0 + i] : 0];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const int q = q0 + i;
                    const int ox = q / 9, oy = (q / 3) % 3, oz = q % 3;
                    const v4f p4 = {ps[i].x, ps[i].y, ps[i].z, ps[i].w};
                    const v2f dj = pk_dot4(p4, out, v2f{0.f, 0.f});
                    float r = dj.x * (C[0][ox] * C[1][oy] * C[2][oz] * iw_r);
                    if (JAC) r = fmaf(dj.y, bw[0][ox] * bw[1][oy] * bw[2][oz], r);
                    if (nbq[q] >= 0) dst[__popc(cmask & ((1u << q) - 1u))] = r * scale;
                }
            }
            for (int q = 0; q < padz; ++q) dst[kc + q] = 0.f;
        }
    }
    // the wavefront's 64 rows as one contiguous run
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float* wimg = img[wv];
    if (CMP) {
        // image words [shift, shift + nrun) <-> array words [G, G + nrun); piece p = image words 4 p .. 4 p + 3 <-> array words G - shift + 4 p ..
        float* gbase = rows_out + (G - shift);
        const int end = shift + nrun;
        for (int p = lane; 4 * p < end; p += 64) {
            const int w0 = 4 * p;
            if (w0 >= shift && w0 + 3 < end) {
                *reinterpret_cast<float4*>(gbase + w0) = *reinterpret_cast<const float4*>(wimg + w0);
            } else {
                for (int q = w0; q < w0 + 4; ++q)
                    if (q >= shift && q < end) gbase[q] = wimg[q];
            }
        }
        return;
    }
    const int64_t left = rows_total - R0;
    float* gbase = rows_out + ((int64_t)d * rows_total + R0) * 27;
    if (left >= 64) {                                          // 432 16-byte pieces: six full instructions + 48 lanes of a seventh
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int w0 = (j * 64 + lane) * 4;
            if (j < 6 || lane < 48) {
                const float4 t4 = *reinterpret_cast<const float4*>(wimg + w0);
                const f32x4_u o4 = {t4.x, t4.y, t4.z, t4.w};
                *reinterpret_cast<f32x4_u*>(gbase + w0) = o4;
            }
        }
    } else {
        for (int q = lane; q < (int)left * 27; q += 64) gbase[q] = wimg[q];
    }
}

// row -> cell at every level (the compact layout needs the cells' row spans before the rows can be written): one lane per row, the
// home-slot probes of all levels issued together
__global__ void __launch_bounds__(256) k_row_cells_merged(nksr_hier_t hier, const float* __restrict__ xyz_a, const float* __restrict__ xyz_b,
                                                          const int32_t* __restrict__ row_src, int64_t rows_total, int32_t* __restrict__ row_cells) {
    const int64_t R = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (R >= rows_total) return;
    const int L = hier.depth;
    const int src = row_src[R];
    if (src < 0) {
        for (int d = 0; d < L; ++d) row_cells[(int64_t)d * rows_total + R] = -1;
        return;
    }
    const float* xp = ((src & 3) == 0 ? xyz_a : xyz_b) + (int64_t)(src >> 2) * 3;
    const float x[3] = {xp[0], xp[1], xp[2]};
    int64_t key[NKSR_MAX_DEPTH], k0[NKSR_MAX_DEPTH];
    uint32_t slot[NKSR_MAX_DEPTH];
    int v0[NKSR_MAX_DEPTH];
#pragma unroll
    for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
        key[d] = 0; k0[d] = -1; slot[d] = 0; v0[d] = -1;
        if (d < L && hier.lv[d].n > 0) {
            const SiteCell g = site_geometry(d, hier.inv_w0, x);
            key[d] = morton_biased(g.I[0], g.I[1], g.I[2], NKSR_BIAS0 >> d);
            slot[d] = hash_slot(key[d], hier.lv[d].hcap);
            k0[d] = hier.lv[d].hkeys[slot[d]];
            v0[d] = hier.lv[d].hvals[slot[d]];
        }
    }
#pragma unroll
    for (int d = 0; d < NKSR_MAX_DEPTH; ++d) {
        if (d >= L) break;
        const int c = hier.lv[d].n > 0 ? hash_find_after(hier.lv[d].hkeys, hier.lv[d].hvals, hier.lv[d].hcap, key[d], slot[d], k0[d], v0[d]) : -1;
        row_cells[(int64_t)d * rows_total + R] = c >= 0 ? hier.lv[d].offset + c : -1;
    }
}

uint32_t nksr_rows_level_map(int depth, int* nlev);          // (csrc/kfield.hip)

extern "C" int nksr_kernel_rows_merged(const nksr_hier_t* h, const float* xyz_pos, const float* scale_pos, float row_scale_pos,
                                       const float* xyz_nrm, const float* scale_nrm, float row_scale_nrm, int approx, const int32_t* row_src,
                                       int64_t rows_total, int32_t* row_cells, int cells_given, const int32_t* compact_nbr32, float* rows_out, void* stream) {
    if (rows_total <= 0) return NKSR_OK;
    if (!h || !row_src || !rows_out || (!xyz_pos && !xyz_nrm)) return nksr_set_error(NKSR_ERR_ARG, "merged rows: NULL arrays");
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    if (h->kdim != 4 || (h->hidden != 16 && h->hidden != 32))
        return nksr_set_error(NKSR_ERR_ARG, "merged kernel rows need kernel_dim 4 and hidden_dim 16 / 32 (got %d, %d)", h->kdim, h->hidden);
    int nlev = 0;
    const uint32_t level_map = nksr_rows_level_map(h->depth, &nlev);
    dim3 grid(nksr_blocks(rows_total, 256), nlev), block(256);
    const bool jac = xyz_nrm && !approx;
    if (compact_nbr32 && (!row_cells || ((uintptr_t)rows_out & 15))) return nksr_set_error(NKSR_ERR_ARG, "compact rows need row_cells (nksr_row_cells_merged) and a 16-byte aligned array");
    if (cells_given && !row_cells) return nksr_set_error(NKSR_ERR_ARG, "cells_given without row_cells");
#define NKSR_LAUNCH_MERGED(H_, J_)                                                                                                             \
    do {                                                                                                                                       \
        if (compact_nbr32) hipLaunchKernelGGL((k_kernel_rows_merged<H_, J_, true, true>), grid, block, 0, (hipStream_t)stream, *h, xyz_pos, scale_pos, row_scale_pos, \
                                              xyz_nrm, scale_nrm, row_scale_nrm, row_src, rows_total, row_cells, compact_nbr32, rows_out, level_map); \
        else if (cells_given) hipLaunchKernelGGL((k_kernel_rows_merged<H_, J_, false, true>), grid, block, 0, (hipStream_t)stream, *h, xyz_pos, scale_pos, row_scale_pos, \
                                                 xyz_nrm, scale_nrm, row_scale_nrm, row_src, rows_total, row_cells, compact_nbr32, rows_out, level_map); \
        else hipLaunchKernelGGL((k_kernel_rows_merged<H_, J_, false, false>), grid, block, 0, (hipStream_t)stream, *h, xyz_pos, scale_pos, row_scale_pos, \
                                xyz_nrm, scale_nrm, row_scale_nrm, row_src, rows_total, row_cells, compact_nbr32, rows_out, level_map);          \
    } while (0)
    if (h->hidden == 16) { if (jac) NKSR_LAUNCH_MERGED(16, true); else NKSR_LAUNCH_MERGED(16, false); }
    else { if (jac) NKSR_LAUNCH_MERGED(32, true); else NKSR_LAUNCH_MERGED(32, false); }
#undef NKSR_LAUNCH_MERGED
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

extern "C" int nksr_row_cells_merged(const nksr_hier_t* h, const float* xyz_pos, const float* xyz_nrm, const int32_t* row_src, int64_t rows_total,
                                     int32_t* row_cells_out, void* stream) {
    if (rows_total <= 0) return NKSR_OK;
    if (!h || !row_src || !row_cells_out || (!xyz_pos && !xyz_nrm)) return nksr_set_error(NKSR_ERR_ARG, "row cells: NULL arrays");
    if (h->depth < 1 || h->depth > NKSR_MAX_DEPTH) return nksr_set_error(NKSR_ERR_ARG, "bad depth %d", h->depth);
    hipLaunchKernelGGL(k_row_cells_merged, dim3(nksr_blocks(rows_total, 256)), dim3(256), 0, (hipStream_t)stream, *h, xyz_pos, xyz_nrm, row_src, rows_total, row_cells_out);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

__global__ void k_row_sources(const int32_t* __restrict__ first_row, int64_t n, int ncomp, int kind0, int32_t* __restrict__ row_src) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = first_row[i];
    for (int c = 0; c < ncomp; ++c) row_src[r + c] = (int32_t)(i << 2) | (kind0 + c);
}
extern "C" int nksr_row_sources(const int32_t* first_row, int64_t n, int ncomp, int kind0, int32_t* row_src, void* stream) {
    if (n <= 0) return NKSR_OK;
    if (!first_row || !row_src) return nksr_set_error(NKSR_ERR_ARG, "row sources: NULL arrays");
    if (n >= (1ll << 29) || ncomp < 1 || kind0 < 0 || kind0 + ncomp > 4) return nksr_set_error(NKSR_ERR_ARG, "row sources: n >= 2^29 or bad kinds");
    hipLaunchKernelGGL(k_row_sources, dim3(nksr_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, first_row, n, ncomp, kind0, row_src);
    NKSR_CHECK_LAUNCH();
    return NKSR_OK;
}

