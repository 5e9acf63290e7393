/* nksr_hip.h -- C-ABI of the MI355X-native NKSR solve-time hot path.
 *
 * Drop-in boundary.  The reference's native boundary is the
 * compiled `_C` module inside the un-vendored `nksr` wheel (README.md:46), reached
 * through the Python surface evidenced at examples/recons_simple.py:25-27,
 * models/nksr_net.py:57-62,91-112 and models/loss.py:189-198.  Each entry point
 * below names the reference-side call it serves.  Conventions:
 *   - plain device pointers + sizes; the caller (Python/torch) owns every buffer
 *   - `stream` is a hipStream_t passed as void*
 *   - return 0 on success, negative on error; nksr_last_error() gives the message
 *   - no hidden host synchronisation unless the doc comment says "syncs"
 */
#ifndef NKSR_HIP_H
#define NKSR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NKSR_MAX_DEPTH 6
#define NKSR_OK 0
#define NKSR_ERR_ARG (-1)
#define NKSR_ERR_HIP (-2)
#define NKSR_ERR_CAPACITY (-3)

const char* nksr_last_error(void);
int nksr_version(void);

/* ---- one level of the sparse voxel hierarchy (SparseFeatureHierarchy.grids[d],
 *      models/nksr_net.py:57-62, models/loss.py:33-46) ------------------------------- */
typedef struct {
    int32_t n;                 /* active voxels, canonical order = ascending Morton key */
    int32_t offset;            /* first unknown index of this level in alpha            */
    const int64_t* keys;       /* [n] sorted Morton keys                                */
    const int32_t* ijk;        /* [n,3]                                                 */
    const int32_t* nbr;        /* [n,27] neighbour voxel index or -1                    */
    const int64_t* hkeys;      /* open-addressing hash: keys (-1 = empty)               */
    const int32_t* hvals;      /*                      values                           */
    int32_t hcap;              /* capacity (power of two)                               */
    const float* feat;         /* [n,K] basis features                                  */
    const float* psi;          /* [n,K] phi_d(c_j)                                      */
    const float* mlp;          /* interpolator weights W1[H,K] b1[H] W2[H,H] b2[H] W3[K,H] b3[K] */
} nksr_level_t;

typedef struct {
    int32_t depth, kdim, hidden;
    float inv_w0;              /* fp32 reciprocal of the finest voxel size              */
    nksr_level_t lv[NKSR_MAX_DEPTH];
} nksr_hier_t;

/* ---- device primitives (rocPRIM-backed; tmp==NULL -> size query) --------------------- */
int nksr_sort_keys_u64(void* tmp, size_t* tmp_bytes, const uint64_t* in, uint64_t* out, int64_t n,
                       int begin_bit, int end_bit, void* stream);
int nksr_sort_pairs_u64_u32(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout,
                            const uint32_t* vin, uint32_t* vout, int64_t n, int begin_bit, int end_bit, void* stream);
int nksr_unique_u64(void* tmp, size_t* tmp_bytes, const uint64_t* in, uint64_t* out, int64_t* d_count,
                    int64_t n, void* stream);
int nksr_exclusive_sum_i32(void* tmp, size_t* tmp_bytes, const int32_t* in, int32_t* out, int64_t n, void* stream);
int nksr_exclusive_sum_i64(void* tmp, size_t* tmp_bytes, const int64_t* in, int64_t* out, int64_t n, void* stream);

/* ---- hierarchy build/query (SparseFeatureHierarchy.build_point_splatting,
 *      models/nksr_net.py:62; grids[d].active_grid_coords models/loss.py:36) ------------ */
/* mode 0: 8 nearest voxel centres per point (encoder hierarchy); mode 1: containing
 * cell + 26 neighbours (decoder structure rule).  keys_out has n*8 / n*27 entries. */
int nksr_splat_keys(const float* xyz, int64_t n, float inv_w0, int level, int mode, int64_t* keys_out, void* stream);
/* Same footprints generated from the unique level-`level` CELLS that hold points instead of from every
 * point: mode 0 -> 8 keys per cell at level+1 (the level-(l+1) half index of a point is its level-l cell
 * index), mode 1 -> 27 keys per cell at the same level. */
int nksr_cell_footprint_keys(const int64_t* cell_keys, int64_t nc, int level, int mode, int64_t* keys_out, void* stream);
/* Axis-aligned bounding box of a cloud (the host side of detail_level / chunk_size needs it: NKSR-USAGE.md:129-137):
 * out6 = (min x, y, z, max x, y, z), exact; out6[0] = NaN when any coordinate is NaN / infinite (the readback of the box is the
 * finiteness check of the input); work: nksr_bbox_work_floats() floats of scratch. */
int nksr_bbox(const float* xyz, int64_t n, float* work, float* out6, void* stream);
int64_t nksr_bbox_work_floats(void);
/* The same key streams (xyz != NULL: nksr_splat_keys; cell_keys != NULL: nksr_cell_footprint_keys; exactly one of them) with the
 * duplicates inside each workgroup's run of Morton-ordered elements removed (LDS hash set): the multiset differs, the SET of keys
 * is the same, so sort + unique behind it give the identical level.  keys_out: room for n * (8 | 27) keys; *count_out (device)
 * = number of keys written, in no particular order.  mode 2 (cell_keys, level 0): the corner keys of nksr_cell_corner_keys;
 * mode 3 (cell_keys, level 0): cell_keys IS the stream (n non-negative keys, e.g. the edge keys of nksr_mc_emit). */
int nksr_footprint_keys_dedup(const float* xyz, const int64_t* cell_keys, int64_t n, float inv_w0, int level, int mode,
                              int64_t* keys_out, int64_t* count_out, void* stream);
/* Morton key of the level-0 cell containing each point. */
int nksr_point_keys(const float* xyz, int64_t n, float inv_w0, int64_t* keys_out, void* stream);
int nksr_decode_keys(const int64_t* keys, int64_t n, int level, int32_t* ijk_out, void* stream);
int nksr_encode_keys(const int32_t* ijk, int64_t n, int level, int64_t* keys_out, void* stream);
/* Open-addressing table Morton key -> value (the key's rank): hcap a power of two >= max(8, 2 n); hkeys must be pre-filled with 0xFF
 * bytes.  The eight children of a cell (keys equal above their low three bits) share one 64-byte line of hkeys; a collision moves to
 * another line (csrc/common.h: hash_slot / hash_next). */
int nksr_hash_build(const int64_t* keys, int32_t n, int64_t* hkeys, int32_t* hvals, int32_t hcap, void* stream);
int nksr_hash_query(const int64_t* q, int64_t nq, const int64_t* hkeys, const int32_t* hvals, int32_t hcap,
                    int32_t* idx_out, void* stream);
int nksr_build_nbr(const int32_t* ijk, int32_t n, int level, const int64_t* hkeys, const int32_t* hvals,
                   int32_t hcap, int32_t* nbr_out, void* stream);
/* The same table derived from the next-coarser level (every voxel's parent active): parent_idx [n] = index of voxel i's parent in
 * the coarse level (-1: none -- then the whole level falls back to the hash, decided on the device), coarse_nbr [n_coarse, 27],
 * work = 2 n_coarse + 1 int32, the first n_coarse and the last ZEROED by the caller.  Same result as nksr_build_nbr. */
int nksr_build_nbr_from_parent(const int32_t* ijk, const int64_t* keys, int32_t n, int level, const int64_t* hkeys, const int32_t* hvals,
                               int32_t hcap, const int32_t* parent_idx, const int32_t* coarse_nbr, int32_t n_coarse, int32_t* work,
                               int32_t* nbr_out, void* stream);
/* start/end of the sites (sorted level-0 Morton keys) that fall inside each level-d voxel */
int nksr_site_ranges(const int64_t* site_keys, int64_t ns, const int64_t* vox_keys, int32_t n, int level,
                     int32_t* start_out, int32_t* end_out, void* stream);

/* Trilinear splat (weighted sum + weight sum) of per-point features onto the voxels of one
 * level; points must be Morton-sorted with start/end = nksr_site_ranges of that level.
 * Point encoder skip path (network.encoder, models/nksr_net.py:73). */
int nksr_splat_trilinear(const float* xyz_sorted, const float* feat_sorted, int C, const int32_t* start,
                         const int32_t* end, const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w,
                         float* out, float* wsum_out, void* stream);

/* ---- sparse feature-hierarchy network (network.encoder / network.unet, models/nksr_net.py:73-78;
 *      csrc/nn.hip).  C = unet.f_maps = 32 (configs/default/train.yaml:17-18). --------------------- */
/* per-point MLP on [local cell coordinate - 1/2 (3), orientation feature (3)]:  W1 [C,6], W2 [C,C] */
int nksr_point_mlp(const float* xyz, const float* feat, int64_t n, float inv_w0, int C, const float* W1,
                   const float* b1, const float* W2, const float* b2, float* out, void* stream);
/* trilinear splat-MEAN of C-channel point features onto one level (points Morton-sorted) */
int nksr_splat_mean(const float* xyz_sorted, const float* feat_sorted, int C, const int32_t* start,
                    const int32_t* end, const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w, float* out,
                    void* stream);
/* 3x3x3 submanifold sparse convolution on the fp32 matrix cores: out = act(b + sum_s W[s]^T in[nbr[:,s]]
 * (+ residual)),  W [27, C, C]  in / W: 16-byte aligned (NKSR_ERR_ARG otherwise). */
int nksr_sparse_conv3(const float* in, const int32_t* nbr, int32_t n, int C, const float* W, const float* bias,
                      const float* residual, int relu, float* out, void* stream);
/* Weight gradient of nksr_sparse_conv3 (training path, network.unet under autograd, models/nksr_net.py:74-78):
 * partial[chunk][s][ci][co] = sum over the chunk's voxels i of in[nbr[i][s]][ci] * gz[i][co]  (gz = output gradient through the
 * activation); nksr_conv3_wgrad_chunks(n) chunks, the caller adds them up (fixed order: deterministic). */
int64_t nksr_conv3_wgrad_chunks(int32_t n);
int nksr_conv3_wgrad(const float* in, const int32_t* nbr, int32_t n, int C, const float* gz, float* partial, void* stream);
/* mean over the children (contiguous Morton range start/end in the finer level) of every voxel */
int nksr_pool_children(const float* child_feat, const int32_t* start, const int32_t* end, int32_t n_parent, int C,
                       float* out, void* stream);
/* out[i] = (idx[i] >= 0 ? src[idx[i]] : 0) (+ add[i]) -- hierarchy transfer / parent->child up-sampling */
int nksr_gather_rows(const float* src, const int32_t* idx, int64_t n, int C, const float* add, float* out, void* stream);
/* per-voxel linear head: out [n, Cout] = in [n, 32] W^T + b */
int nksr_linear(const float* in, int64_t n, int Cin, const float* W, const float* b, int Cout, float* out, void* stream);

/* ---- UDF mask branch: NeuralField(svh=udf_svh, decoder=network.udf_decoder, features=feat.udf_features)
 *      .set_level_set(2 * voxel_size)  (models/nksr_net.py:124-130; configs/carla/train.yaml:8-9) ---------- */
/* plane features of one level, out [n, 8] = (occupied, centroid offset xyz in voxel units, unit mean normal
 * xyz, 0): trilinear-weighted over the Morton-sorted points around every voxel */
int nksr_splat_plane(const float* xyz_sorted, const float* normal_sorted, const int32_t* start, const int32_t* end,
                     const int32_t* nbr, const int32_t* ijk, int32_t n, float inv_w, float* out, void* stream);
/* unsigned plane distance decoded from the 8 voxel centres around every query (1e30 where none is
 * occupied); only_unset != 0 keeps entries already decoded at a finer level */
int nksr_udf_decode(const nksr_level_t* level, int level_index, const float* feat, const float* xyz, int64_t n,
                    float inv_w, float voxel_size, int only_unset, float* out, void* stream);

/* ---- neural kernel (KernelField, models/nksr_net.py:91-96) --------------------------- */
int nksr_voxel_psi(const float* feat, int32_t n, int kdim, int hidden, const float* mlp, float* psi_out, void* stream);
/* Dense-slot kernel rows at arbitrary sites.  val [n, L, 27] (may be NULL when only the gradient rows are
 * wanted); dval [n, 3, L, 27] (may be NULL).  approx!=0 drops the d(phi)/dx term (approx_kernel_grad, recons_waymo.py:33).  Every output is
 * multiplied by row_scale (the assembly takes rows pre-multiplied by sqrt(set weight), see nksr_assemble) -- or, when site_scale
 * (device, [n]) is given, by site_scale[i] instead: the sites of a batched chunk solve carry their own chunk's weight. */
int nksr_kernel_rows(const nksr_hier_t* h, const float* xyz, int64_t n, int approx, float row_scale, const float* site_scale,
                     int64_t level_stride, const int32_t* row_index, int32_t* row_cells, float* val, float* dval, void* stream);
/* level_stride > 0: LEVEL-MAJOR output, row (site i, component a) of level d at val / dval + ((d * level_stride + r_i + a) * 27) with
 * r_i = row_index ? row_index[i] : i * ncomp (ncomp = 1 for val, 3 for dval; ONE of val / dval when row_index is given) -- the layout
 * of the matrix-free solve (nksr_fused_op_t.rows_all): row_index lets several site sets share one Morton-ordered row list;
 * row_cells [L, level_stride] (may be NULL) receives the global unknown index of the level-d cell of every row written (-1 = none). */
/* The rows of BOTH site sets of the matrix-free operator in ONE pass over its merged row list (kernel_dim 4; KernelField.solve,
 * models/nksr_net.py:105-112 with fused_mode, examples/recons_waymo.py:33): rows_out [L][rows_total][27], row r of the list is
 * row_src[r] = (site << 2) | kind -- kind 0: the value row of position site `site` (xyz_pos), kind 1 + a: the d/dx_a row of normal
 * site `site` (xyz_nrm); row_src[r] < 0: a pad row (zeros, row_cells -1).  One launch writes every 128-byte line of rows_out whole
 * (a launch per set writes the interleaved rows as partial lines: 1.1 - 1.8 TB/s against 5.7); values bit-identical to
 * nksr_kernel_rows.  scale_* (device, per site) or row_scale_* as for nksr_kernel_rows; either set may be absent (xyz NULL).
 * sites < 2^29 per set. */
int nksr_kernel_rows_merged(const nksr_hier_t* h, const float* xyz_pos, const float* scale_pos, float row_scale_pos,
                            const float* xyz_nrm, const float* scale_nrm, float row_scale_nrm, int approx, const int32_t* row_src,
                            int64_t rows_total, int32_t* row_cells, int cells_given, const int32_t* compact_nbr32, float* rows_out, void* stream);
/* cells_given == 0: row_cells (may be NULL) is WRITTEN (a hash probe per row and level); != 0: row_cells is READ (nksr_row_cells_merged
 * made it: one pass with the probes of all levels in flight together, and the row kernel's chain of dependent loads loses two links).
 * compact_nbr32 == NULL: dense rows as above.  compact_nbr32 = nksr_fused_op_t.nbr32: COMPACT rows
 * (nksr_fused_op_t.compact) -- row_cells [L][rows_total] is READ (nksr_row_cells_merged made it before the tables could be built) and
 * every wavefront writes the words of its 64 rows, level by level, as one contiguous run of the compact array.
 * nksr_row_cells_merged: row_cells_out[d][r] = global unknown index of the level-d cell of row r's site, -1 = none (pad rows: -1). */
int nksr_row_cells_merged(const nksr_hier_t* h, const float* xyz_pos, const float* xyz_nrm, const int32_t* row_src, int64_t rows_total,
                          int32_t* row_cells_out, void* stream);
/* row_src of nksr_kernel_rows_merged from the sets' first rows: row_src[first_row[i] + c] = (i << 2) | (kind0 + c), c < ncomp
 * (ncomp 1, kind0 0: position sites; ncomp 3, kind0 1: normal sites).  The caller pre-fills row_src with -1. */
int nksr_row_sources(const int32_t* first_row, int64_t n, int ncomp, int kind0, int32_t* row_src, void* stream);
/* Rank-4 FACTORS of the same rows (kernel_dim 4 only; the matrix-free solve's row format since round 5, nksr_fused_op_t.fac_vec /
 * fac_pos): per row and level one 16-byte record instead of 27 slots.  grad == 0: one row per site, vec = phi_d(x) * scale.
 * grad != 0: FOUR rows per site -- a header row (vec = phi * scale) and one row per axis a (vec = d phi / d x_a * scale; 0 with
 * approx != 0).  Row r of site i = (row_index ? row_index[i] : i * rows_per_site) + q.  vec_out [L][level_stride][4],
 * pos_out [level_stride][4] = (x * inv_w0, kind bits), row_cells as for nksr_kernel_rows.  Rows of a site outside every active
 * cell of a level are 0 there. */
int nksr_kernel_factors(const nksr_hier_t* h, const float* xyz, int64_t n, int grad, int approx, float row_scale, const float* site_scale,
                        int64_t level_stride, const int32_t* row_index, int32_t* row_cells, float* vec_out, float* pos_out, void* stream);
/* f(x) (and gradient if grad_out != NULL): field.evaluate_f, models/loss.py:189-198.  alpha == NULL: the hierarchy's psi arrays
 * already hold alpha_j psi_j (one gather per neighbour instead of two).  A query outside every active cell of a level still sees the
 * voxels whose support covers it (hash lookups); active_only != 0 drops those levels instead -- the support of nksr_kernel_rows,
 * which is what the training path's backward differentiates. */
int nksr_evaluate_f(const nksr_hier_t* h, const float* alpha, const float* xyz, int64_t n, int approx, int active_only,
                    float* f_out, float* grad_out, void* stream);

/* ---- backward of the kernel rows w.r.t. theta = (basis features, interpolator weights): the training path, models/nksr_net.py:
 *      105-112 (loss -> solve_non_fused / evaluate_f -> network).  With per-row factors g_r[s] = a_r lam[col] + b_r alpha[col] the
 *      kernels ADD  dS/dtheta,  S = sum_r sum_s row_scale R_r[s] g_r[s]  (R = the rows of nksr_kernel_rows: value rows, or
 *      gradient rows with / without the d(phi)/dx term) to the caller's zero-initialised arrays.  coef_a / coef_b: [n] (value rows) or
 *      [n, 3] (gradient rows), either may be NULL (= 0; coef_a needs lam).  Floating-point atomics: reproducible to rounding. */
typedef struct {
    float* gfeat[NKSR_MAX_DEPTH]; /* [n_d, kdim] += through the trilinear stencil of the sites                                    */
    float* gpsi[NKSR_MAX_DEPTH];  /* [n_d, kdim] += dS/dpsi_j of the neighbour voxels (finish with nksr_voxel_psi_vjp)           */
    float* gmlp[NKSR_MAX_DEPTH];  /* packed interpolator weights of the level (W1 b1 W2 b2 W3 b3) +=                             */
} nksr_theta_grad_t;
int nksr_kernel_rows_vjp(const nksr_hier_t* h, const float* xyz, int64_t n, int grad_rows, int approx, float row_scale, const float* coef_a,
                         const float* coef_b, const float* alpha, const float* lam, const nksr_theta_grad_t* out, void* stream);
/* psi_j = f_j + MLP(f_j) (nksr_voxel_psi) backwards: gfeat_j += (d psi_j / d f_j)^T gpsi_j, gmlp += the weight cotangents */
int nksr_voxel_psi_vjp(const float* feat, int32_t n, int kdim, int hidden, const float* mlp, const float* gpsi, float* gfeat, float* gmlp,
                       void* stream);

/* ---- normal-equation assembly (KernelField.solve_non_fused, models/nksr_net.py:105-112) */
typedef struct {
    int64_t n;                 /* sites                                                */
    int32_t ncomp;             /* 1 (position rows, G) or 3 (gradient rows, Q)         */
    float weight;              /* weight of the set (>= 0).  For a bitwise symmetric matrix pass val / target
                                * pre-multiplied by sqrt(w) (nksr_kernel_rows row_scale) and weight = 1 */
    const float* val;          /* [n, ncomp, L, 27] dense-slot rows                    */
    const float* target;       /* [n, ncomp] right-hand side values or NULL (zero)     */
    const int32_t* start[NKSR_MAX_DEPTH]; /* per level: [n_d] site range per voxel     */
    const int32_t* end[NKSR_MAX_DEPTH];
    int64_t level_stride;      /* 0: val is site-major as above.  > 0: val is the LEVEL-MAJOR array of the matrix-free operator
                                * ([L, level_stride, 27], nksr_fused_op_t.rows_all) and site i owns the rows row_index[i] .. + ncomp */
    const int32_t* row_index;  /* [n] first row of every site (level-major layout; NULL = i * ncomp) */
    const int32_t* compact_cells; /* COMPACT rows of the matrix-free operator (nksr_fused_op_t.compact): its row_cells [L][level_stride] and */
    const int32_t* compact_nbr32; /* nbr32 [M][32]; NULL = dense rows.  val is then the compact array, one row per "site" (ncomp 1, no row_index) */
    int64_t level_base;        /* level-major dense rows only: val STARTS at this level ([L - level_base, level_stride, 27]; the factor form of the
                                * operator keeps dense rows of its coarse levels only) -- the assembly must then be asked for levels >= level_base */
} nksr_siteset_t;

/* Structure pass.  Per row: rowcount = structural upper entries (column voxel exists, B-spline supports
 * overlap, col > row), crosscount = those of them whose column lies on a coarser level (only these are
 * mirrored), samelow = same-level neighbours with col < row (emitted by the row itself: bitwise equal to the
 * transposed entry); indeg[col] += 1 for every cross-level upper entry (integer atomics; indeg must be
 * zeroed).  Final CSR row = [indeg cross-level mirrors][samelow][rowcount own upper][diagonal]. */
int nksr_assemble_count(const nksr_hier_t* h, void* workspace, int32_t* rowcount, int32_t* crosscount, int32_t* samelow,
                        int32_t* indeg, void* stream);
/* Bytes of scratch (per-cell blocks + per-row column map) shared by nksr_assemble_count / nksr_assemble. */
size_t nksr_assemble_workspace_bytes(const nksr_hier_t* h);
/* Two-phase assembly of  sum_s w_s R_s^T R_s + reg I  (csrc/assemble.hip): per-cell dense blocks (fp32 MFMA),
 * then one wavefront per row.  rowptr = exclusive scan of (indeg + samelow + rowcount + 1), mir_off =
 * exclusive scan of crosscount.  Same-level lower, own upper entries and the diagonal are written straight
 * into cols_out / vals_out (tile-interleaved physical layout, see nksr_spmv_csr); the mirrored copies of the
 * cross-level entries go to mir_keys (src_row << col_bits | dst_row) / mir_vals.  Also writes diag_out and
 * b = sum_s w_s R_s^T t_s.
 * split_scratch (nksr_assemble_split_bytes(h, sum_s n_s ncomp_s) bytes; NULL = off): levels whose cells hold > 1024 site rows
 * on average are accumulated by several wavefronts per cell and reduced in a fixed order. */
size_t nksr_assemble_split_bytes(const nksr_hier_t* h, int64_t total_rows);
int nksr_assemble(const nksr_hier_t* h, const nksr_siteset_t* sets, int nsets, float reg, int col_bits,
                  void* workspace, const int32_t* rowptr, const int32_t* indeg, const int32_t* samelow,
                  const int32_t* mir_off, int col_format, int32_t* cols_out, float* vals_out, float* diag_out,
                  uint64_t* mir_keys, float* mir_vals, float* b_out, void* split_scratch, size_t split_scratch_bytes, void* stream);
/* Mirrors stably sorted by destination row (low col_bits of the key) -> their CSR slots;
 * mirptr = exclusive scan of indeg. */
int nksr_place_mirrors(const uint64_t* keys_sorted, const float* vals_sorted, int64_t n, int col_bits,
                       const int32_t* rowptr, const int32_t* mirptr, int col_format, int32_t* cols_out, float* vals_out,
                       void* stream);

/* ---- PCG (the CG SpMV is the roofline kernel) -------------------- */
/* nnz-chunked streaming CSR SpMV (csrc/pcg.hip).  cols/vals live in a tile-interleaved physical layout
 * chosen by col_format (the same value must be given to nksr_assemble / nksr_place_mirrors, which
 * write it):
 *   0: 256-entry tiles, logical entry m of a tile at 4*(m%64) + m/64, int32 columns, zero-padded (valid
 *      column 0, value 0) to a multiple of 4096 entries;
 *   1: 192-entry tiles, entry m at 3*(m%64) + m/64, padded to a multiple of 4608 entries; the SpMV reads
 *      the columns as one 64-bit word per three entries (21 bits each, nksr_pack_cols21 converts the
 *      int32 array written by the assembly): 6.67 instead of 8 bytes per entry, requires M <= 2^21.
 *   2 (nksr_assemble / nksr_place_mirrors only): plain CSR order, nnz entries, no padding -- the small coarse-level block of the
 *      preconditioner (nksr_coarse_precond_t); the streaming SpMV does not take it.
 * The plan (first row of every chunk) lives in `workspace` (nksr_spmv_workspace_bytes) and must be built
 * once per matrix with nksr_spmv_plan. */
size_t nksr_spmv_workspace_bytes(int64_t nnz);
int nksr_spmv_plan(const int32_t* rowptr, int32_t M, int64_t nnz, int col_format, void* workspace, void* stream);
int nksr_spmv_csr(const int32_t* rowptr, const void* cols, const float* vals, int32_t M, int64_t nnz, int col_format,
                  const float* x, float* y, void* workspace, void* stream);
/* int32 columns in the col_format-1 layout (n_padded entries, a multiple of 192) -> packed 64-bit words
 * (n_padded / 3 of them) */
int nksr_pack_cols21(const int32_t* cols32, int64_t n_padded, uint64_t* packed_out, void* stream);
/* Experimental kernel variants for tools/spmv_probe.py (0 = default). */
int nksr_spmv_set_variant(int v);
/* Scratch bytes required by nksr_pcg_solve. */
size_t nksr_pcg_workspace_bytes(int32_t M, int64_t nnz);
/* Jacobi-PCG, x0 = 0.  info_out (host, 3 doubles, may be NULL): [0]=iterations [1]=relative residual (negated on a hard failure:
 * r.z <= 0 with the Jacobi preconditioner, or NaN) [2]=segments whose coarse-level block lost definiteness (r.z <= 0) and that went on
 * with Jacobi alone from the current iterate (a restart on the device, no error).
 * Checks convergence every `check_every` iterations (one stream sync each) -- syncs.
 * coarse_precond (struct nksr_coarse_precond_t, declared below; NULL = Jacobi only): the diagonal block of the coarse levels. */
struct nksr_coarse_precond_s;
int nksr_pcg_solve(const int32_t* rowptr, const void* cols, const float* vals, const float* diag, int32_t M,
                   int64_t nnz, int col_format, const float* b, float* x, float tol, int max_iter, int check_every,
                   void* workspace, const struct nksr_coarse_precond_s* coarse_precond, double* info_out, void* stream);
/* Live profiling of the SpMV launches inside nksr_pcg_solve (HIP events on the solve's stream).
 * Returns and resets the accumulated milliseconds / launch count, then sets the enable flag. */
int nksr_pcg_profile(int enable, double* ms_out, int64_t* launches_out);
/* Durations (milliseconds) of the individual applications behind the totals the last nksr_pcg_profile call returned (host array,
 * at most `capacity` written); returns how many there were. */
int64_t nksr_pcg_profile_samples(float* ms_out, int64_t capacity);
/* Bytes of the launches timed since the last call (then reset): the algorithmic figure -- CSR: 8 nnz + 12 M + 4 per launch
 * (SURVEY.md section 8d); matrix-free operator: its algorithmic minimum, 4 bytes per stored entry + 4 per row and level (row ->
 * cell) + 116 per unknown (stencil, x, y) -- and the bytes the physical layout streams (col_format, padding, partial blocks). */
int nksr_pcg_profile_bytes(double* algorithmic_out, double* physical_out);
/* The launches of the last nksr_pcg_profile_bytes call priced by SURVEY.md section 8d's formula (equal to the algorithmic figure for
 * the CSR SpMV; 2 x 8 bytes per stored entry + 12 M + 4 for the matrix-free operator, more than it moves). */
double nksr_pcg_profile_survey_bytes(void);

/* ---- coarse-level block preconditioner of the PCG (csrc/pcg.hip) -----------------------------------------------------
 * The unknowns of the levels >= c0 (the LAST n of the M: unknowns are level-major) take `steps` Jacobi-preconditioned
 * Chebyshev steps on their diagonal block A_cc instead of one Jacobi step; all finer unknowns keep Jacobi.  A_cc: plain CSR
 * (nksr_assemble with col_format 2 on the hierarchy whose fine levels have n = 0, hcap = 0 and whose coarse levels are
 * re-based to offset 0), local indices.  lambda_max: largest eigenvalue of D^-1 A_cc (nksr_coarse_lambda_max, x ~1.1);
 * the polynomial targets the interval [lambda_max / ratio, lambda_max]. */
#define NKSR_PC_MAX_STEPS 16
/* Independent diagonal blocks of one system ("segments": the chunks of a batched chunk solve -- the reference solves its
 * chunks one after the other, examples/recons_by_chunk.py:26-29; here all chunks of a rank share every launch).  The unknowns
 * of segment c are the nranges index ranges [lo[c * nranges + k], hi[c * nranges + k]) (one per hierarchy level, possibly
 * empty).  The PCG gives every segment its own dot products (fixed, segment-relative reduction order), alpha / beta, stopping
 * test and iteration count: the iterates of a segment do not depend on which other segments share the launch.
 * NULL wherever a segments pointer is taken = one segment [0, M). */
typedef struct {
    int32_t nseg, nranges;
    const int32_t* lo;         /* [nseg * nranges] device */
    const int32_t* hi;         /* [nseg * nranges] device */
    double* info;              /* device [nseg * 2] or NULL: iterations, relative residual (negated when the segment stopped on
                                * r.z <= 0) of every segment, refreshed at every convergence check */
} nksr_segments_t;
typedef struct nksr_coarse_precond_s {
    int32_t first, n, steps, format;   /* format 0: plain CSR (rowptr / cols / vals / diag);  1: packed, see below */
    float lambda_scale, ratio; /* the eigenvalue bound of segment c is lambda_scale * lambda[c] (safety margin, ~1.1)          */
    const float* lambda;       /* device [nseg]: nksr_coarse_lambda_max; <= 0 / non-finite: that segment keeps plain Jacobi     */
    const int32_t* row_seg;    /* device [n]: segment of every coarse row; NULL with one segment                                */
    const int32_t* rowptr;     /* [n + 1] */
    const int32_t* cols;       /* [nnz_c] local column indices */
    const float* vals;         /* [nnz_c] */
    const float* diag;         /* [n] diagonal of A_cc */
    float* work;               /* [3 n] floats ([4 n] with format 1) */
    float* coef;               /* [nseg * (1 + 2 * NKSR_PC_MAX_STEPS)] floats: the solve writes the Chebyshev coefficients here */
    /* format 1 (nksr_coarse_pack): the Jacobi-scaled block S = D^-1/2 A_cc D^-1/2 without its unit diagonal, 4 bytes per entry --
     * (half-precision value << 16) | column local to the row's segment -- over the coarse unknowns renumbered segment-major; needs
     * < 2^16 coarse unknowns per segment.  row_seg is then in the NEW order.  The preconditioner stays a fixed symmetric positive
     * operator (all a CG preconditioner must be); a Chebyshev step streams half the bytes. */
    const uint32_t* packed;    /* [packed_rowptr[n]] */
    const int32_t* packed_rowptr; /* [n + 1] new order */
    const float* dis;          /* [n] D^-1/2, new order */
    const int32_t* old_of_new; /* [n] coarse row (PCG order) of every new row */
    const int32_t* seg_base;   /* [nseg + 1] first new row of every segment */
    const float* gersh;        /* device [nseg] or NULL: Gershgorin bound of the same spectrum (nksr_coarse_gershgorin); the interval's upper
                                * end is min(lambda_scale * lambda[c], gersh[c]) */
} nksr_coarse_precond_t;
/* Gershgorin bound per segment of the Jacobi-scaled block described by pc (either format; work: n floats): gersh_out [nseg]. */
int nksr_coarse_gershgorin(const nksr_coarse_precond_t* pc, int32_t nseg, float* work, float* gersh_out, void* stream);
/* format-1 block from the plain CSR: new_of_old / old_of_new = the segment-major renumbering, row_seg_new / seg_base in the new
 * order, packed_rowptr = exclusive sum of the kept entries per row (nksr_coarse_pack_count) in the new order.  Off-diagonal entries
 * with |S_ij| < drop_tol are left out (0 keeps all; S_ij = S_ji bitwise, so the block stays symmetric).  Writes packed_out, dis_out. */
int nksr_coarse_pack_count(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n,
                           const int32_t* old_of_new, float drop_tol, int32_t* lens_out, void* stream);
int nksr_coarse_pack(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n,
                     const int32_t* old_of_new, const int32_t* new_of_old, const int32_t* row_seg_new, const int32_t* seg_base,
                     const int32_t* packed_rowptr, float drop_tol, uint32_t* packed_out, float* dis_out, void* stream);
/* eigenvalue bounds of a format-1 block (power iteration on S, work: 2 n floats): lambda_out [nseg] */
int nksr_coarse_lambda_max_packed(const nksr_coarse_precond_t* pc, int32_t nseg, int iters, float* work, float* lambda_out, void* stream);
/* power iteration (iters steps from the all-ones vector, work: 2 n floats): lambda_out (device, [nseg]) = ||v_k|| / ||v_{k-1}||
 * over the coarse rows of every segment (`first` = unknown index of coarse row 0; the ranges below it are skipped). */
int nksr_coarse_lambda_max(const int32_t* rowptr, const int32_t* cols, const float* vals, const float* diag, int32_t n, int iters,
                           float* work, float* lambda_out, const nksr_segments_t* segments, int32_t first, void* stream);

/* ---- matrix-free ("fused") operator and solve: reconstruct(..., fused_mode=True), examples/recons_waymo.py:33,
 *      recons_waymo_cpu.py:58, gis_app.py:40; KernelField.solve (csrc/fused.hip).  The system matrix is never built:
 *      y = (sum_s R_s^T R_s + reg I) x is applied from the dense-slot kernel rows, cell by cell. ------------------ */
typedef struct {
    int32_t depth, M, n_multi, n_big;
    int64_t rows_total;        /* rows of all site sets in ONE Morton-ordered list (level-0 key of the site; position rows 1 per site,
                                * gradient rows 3 per site); also the level stride of rows_all / row_cells                   */
    const float* rows_all;     /* LEVEL-MAJOR dense-slot rows [depth][rows_total][27], pre-multiplied by sqrt(w) (nksr_kernel_rows, level_stride, row_index) */
    const float* targets_all;  /* [rows_total] right-hand side values pre-multiplied by sqrt(w) (0 for rows without a target); may be NULL if unused */
    const int32_t* row_cells;  /* [depth][rows_total] GLOBAL unknown index of the row's level-d cell, -1 = none (nksr_kernel_rows) */
    const int32_t* nbr32;      /* [M, 32]: [0..26] GLOBAL unknown index of every neighbour voxel or -1; [27] block base, [28] / [29] first / last row of
                                * the cell, [30] first word / 4 of the cell's rows in the compact array, [31] mask of the existing neighbours (nksr_fused_tables) */
    const int32_t* nbrT;       /* [27, M]: the same neighbour indices SLOT-MAJOR (the second product's gather runs one lane per unknown) */
    const int32_t* item_begin; /* [nksr_fused_item_entries(rows_total)] first row of every work item of the sweep (nksr_fused_block_counts) */
    const int32_t* offsets;    /* [M + 1] partial blocks of a cell = [offsets[j], offsets[j + 1])                             */
    const int32_t* multi;      /* [n_multi] the cells with partial blocks (cells whose rows span several 256-row workgroups of the sweep):
                                * first the n_big cells with more than 16 blocks (one workgroup each in the per-cell sum), then the others */
    int64_t nblocks;           /* offsets[M]                                                                                 */
    uint64_t* nnz_counter;     /* device counter or NULL: nksr_fused_rhs_diag leaves the non-zero slots of rows_all here = the
                                * stored entries of G and Q (roofline accounting) */
    void* workspace;           /* nksr_fused_workspace_bytes(nblocks, M)                                                     */
    float* cell_sums;          /* [27, M] per-cell block sums, slot-major, ZERO-INITIALISED by the caller (cells without rows are never written) */
    const int32_t* item_seg;   /* batched chunks (nksr_segments_t), both or neither: [ceil(rows_total / 32) + 1] segment of every 32-row window
                                * of the row list (a segment's rows are padded to a multiple of 256) and                      */
    const int32_t* unknown_seg;/* [M] segment of every unknown: the solve skips the rows / unknowns of segments that have converged */
    /* FACTOR FORM (kernel_dim 4; nksr_kernel_factors): fac_vec != NULL replaces rows_all (which may then be NULL) -- the sweep rebuilds the
     * 27 slots of every row in registers from 16-byte records and the psi stencil of the row's cell.  A normal site owns FOUR rows
     * of the list (header + one per axis); the header row is all-zero. */
    const float* fac_vec;      /* [depth][rows_total][4] phi (position / header rows) or d phi / d x_a (gradient rows), times sqrt(w); 16-byte aligned */
    const float* fac_pos;      /* [rows_total][4] x * inv_w0 (3 floats) + the row's kind as int bits: 0 position, 1 header, 2 + a gradient row of axis a */
    const float* psi_all;      /* [M][4] psi of every unknown, levels concatenated (nksr_voxel_psi)                              */
    float inv_w0;              /* 1 / finest voxel size (fp32, as in nksr_hier_t)                                               */
    int32_t dense_from;        /* nksr_fused_rhs_diag / nksr_fused_expand_rows: the rebuilt rows of the levels >= dense_from are also written to */
    float* dense_out;          /* [depth - dense_from][rows_total][27] (NULL: not wanted) -- what the coarse-level block of the preconditioner is assembled from */
    /* COMPACT rows (round 6; nksr_fused_row_sizes, nksr_kernel_rows_merged): compact != 0 => rows_all holds, for every row, only the slots
     * of its cell's EXISTING neighbours (a quarter of the dense slots of a chunked scene are structural zeros).  Cell j owns the words
     * [4 nbr32[j][30], + rows_j popcount(nbr32[j][31])) in row order, slots in slot order, every block padded to 16 bytes with zeros;
     * words 0..3 of rows_all are zero.  rows_words = length of rows_all in words (without the tail padding). */
    int32_t compact;
    int64_t rows_words;
} nksr_fused_op_t;
/* A UNIT is a maximal run of rows that lie in the same cell at every level (the rows of one level-0 cell); work item i of the sweep =
 * the units that start in the 32-row window [32 i, 32 i + 32) = rows [item_begin[i], item_begin[i + 1]); eight items are a workgroup.
 * A cell whose rows lie inside one workgroup is finished by the sweep, a cell whose rows reach into k > 1 workgroups owns k partial
 * blocks.  rows_all / targets_all must be readable 320 rows past rows_total (unconditional loads).
 * nksr_fused_block_counts: span_out [3, M] (first / last row of every cell, -1 = none; first workgroup), item_begin_out [nksr_fused_item_entries],
 * counts_out [M + 1] (0 or k; last entry 0) -> exclusive scan = offsets -> nksr_fused_tables (nbr32_out [M, 32], nbrT_out [27, M]). */
int64_t nksr_fused_item_entries(int64_t rows_total);
int nksr_fused_block_counts(int32_t depth, int32_t M, int64_t rows_total, const int32_t* row_cells, int32_t* span_out, int32_t* item_begin_out,
                            int32_t* counts_out, void* stream);
int nksr_fused_tables(const nksr_hier_t* h, int64_t rows_total, const int32_t* item_begin, const int32_t* offsets, const int32_t* span,
                      const int32_t* rowbase4, int32_t* nbr32_out, int32_t* nbrT_out, void* stream);
/* COMPACT rows: sizes4_out [M + 1] (int64; last entry 0) = 16-byte units of the rows of every cell, (rows of the cell) x (its existing
 * neighbours), rounded up; rowbase4 [M] (int32; may be NULL: dense rows) = 1 + the exclusive scan of sizes4 (unit 0 is the zero block)
 * goes to nksr_fused_tables, which writes it to nbr32[j][30] next to the neighbour mask nbr32[j][31]. */
int nksr_fused_row_sizes(const nksr_hier_t* h, const int32_t* span, int64_t* sizes4_out, void* stream);
size_t nksr_fused_workspace_bytes(int64_t nblocks, int32_t M);
/* b = sum_s R_s^T t_s and diag = reg + sum_s diag(R_s^T R_s) (either may be NULL): one sweep over the rows serves both. */
int nksr_fused_rhs_diag(const nksr_fused_op_t* op, float reg, float* b_out, float* diag_out, void* stream);
/* Factor form only: the rebuilt rows of the levels >= op->dense_from written dense into op->dense_out (one more set-up sweep;
 * nksr_fused_rhs_diag does the same on its way when dense_out is set).  Scratch: the operator's workspace and cell_sums. */
int nksr_fused_expand_rows(const nksr_fused_op_t* op, void* stream);
/* y = (sum_s R_s^T R_s + reg I) x */
int nksr_fused_apply(const nksr_fused_op_t* op, float reg, const float* x, float* y, void* stream);
/* Jacobi-PCG with that operator; pcg_workspace: nksr_pcg_vector_workspace_bytes(M), or ..._seg(M, nseg, nranges) with segments.
 * Syncs like nksr_pcg_solve.  info_out (3 doubles): [0] = iterations (max over the segments), [1] = relative residual (max; negated if a
 * segment failed hard), [2] = segments that fell back to Jacobi (see nksr_pcg_solve). */
size_t nksr_pcg_vector_workspace_bytes(int32_t M);
size_t nksr_pcg_vector_workspace_bytes_seg(int32_t M, int32_t nseg, int32_t nranges);
int nksr_pcg_solve_fused(const nksr_fused_op_t* op, float reg, const float* diag, const float* b, float* x, float tol, int max_iter,
                         int check_every, void* pcg_workspace, const nksr_coarse_precond_t* coarse_precond /* or NULL: Jacobi */,
                         const nksr_segments_t* segments /* or NULL */, double* info_out, void* stream);

/* ---- chunk plumbing of reconstruct(chunk_size=...) (csrc/chunks.hip; examples/recons_by_chunk.py:29) ---------------------------
 * The chunk grid: chunk id = (cx * grid[1] + cy) * grid[2] + cz; along a split axis (grid[a] > 1) chunk j has the core [lo_j, hi_j),
 * solves the points with  lo_sel[a][j] <= x < hi_sel[a][j]  (= fp32(lo_j - band), fp32(hi_j + band)) and weighs
 * w = prod_a clamp((x - lo_w[a][j]) * inv_2ov, 0, 1) * clamp((hi_w[a][j] - x) * inv_2ov, 0, 1)  (lo_w = fp32(lo_j - ov), hi_w = fp32(hi_j + ov),
 * factors multiplied in the order up_x, dn_x, up_y, ...; an axis that is not split contributes nothing).  Candidates of a point: the
 * chunks within `reach` of its home chunk floor((x - origin) * inv_cs). */
typedef struct {
    int32_t grid[3];
    int32_t reach;
    float origin[3];
    float inv_cs, inv_2ov;
    const float* lo_sel[3];    /* device, [grid[a]] each; may be NULL for an axis that is not split */
    const float* hi_sel[3];
    const float* lo_w[3];
    const float* hi_w[3];
    const float* shift;        /* device [nchunk, 3]: translation of every chunk into its slot of the exploded frame (mode 1) */
} nksr_chunk_grid_t;
/* mode 0: (point, chunk) pairs of the solve's input; mode 1: pairs with a positive blend weight.  chunk_flag [nchunk] >= 0: the chunk
 * takes part.  Count pass -> exclusive scan (offsets, n + 1 entries) -> fill pass: pairs of a point in ascending chunk order;
 * mode 1 also writes the weight and the translated position x + shift[chunk] (one fp32 rounding). */
int nksr_chunk_pair_counts(const nksr_chunk_grid_t* grid, int mode, const float* xyz, int64_t n, const int32_t* chunk_flag,
                           int32_t* counts_out, void* stream);
int nksr_chunk_pair_fill(const nksr_chunk_grid_t* grid, int mode, const float* xyz, int64_t n, const int32_t* chunk_flag,
                         const int32_t* offsets, int64_t* pair_point_out, int32_t* pair_chunk_out, float* pair_w_out,
                         float* pair_xyz_out, void* stream);
/* f = sum_k w_k f_k / max(sum_k w_k, 1e-20) over the pairs [offsets[i], offsets[i + 1]) of query i, in that order (+ the same for the
 * gradient when pair_grad / grad_out are given). */
int nksr_chunk_blend(int64_t n, const int32_t* offsets, const float* pair_w, const float* pair_f, const float* pair_grad,
                     float* f_out, float* grad_out, void* stream);
/* Halo selection of a batch of chunks, one level (the payload of the rank exchange; chunking.pack_halos):
 * voxel i (keys ascending) belongs to chunk seg = the last of the ascending key ranges klo [nchunk] that starts at or before its key;
 * flag = 1 when its centre (ijk + 0.5) w - shift[seg] lies in one of the chunk's band intervals [tlo, thi] ([nchunk, 3, 2] each, an
 * unused interval = (+inf, -inf)) along some axis. */
/* Seam candidates of a rank's mesh piece (the merge on rank 0, nksr_amd/dist.py): vertex i lies on the lattice edge (vertex vkey[i],
 * axis[i]); flag = 1 when one of the four lattice cells around that edge belongs to another rank -- owner[chunk of the centre of the
 * cell's base voxel] != rank, base voxel = floor(cell / cells_per_voxel), centre = (base + 0.5) w0, chunk as nksr_chunk_pair_counts. */
int nksr_edge_seam_flags(const nksr_chunk_grid_t* grid, const int64_t* vkey, const int8_t* axis, int64_t n, int32_t cells_per_voxel, float w0,
                         const int32_t* owner, int32_t rank, uint8_t* flags_out, void* stream);
/* Which points lie in a core this rank owns (reach = 0) or within `reach` of one along the split axes: flag = 1 when
 * owner[chunk of (x + o)] == rank for an offset o in {-reach, 0, +reach} per split axis (the cells a rank meshes / evaluates). */
int nksr_points_owner_flags(const nksr_chunk_grid_t* grid, const float* xyz, int64_t n, float reach, const int32_t* owner, int32_t rank,
                            uint8_t* flags_out, void* stream);
int nksr_halo_band_flags(const int64_t* keys, const int32_t* ijk, int64_t n, const int64_t* klo, int32_t nchunk, const float* shift,
                         const float* tlo, const float* thi, float w, int32_t* seg_out, int32_t* flags_out, void* stream);

/* ---- grid-hash nearest neighbours (csrc/knn.hip) ----------------------------------------------------
 * Points Morton-sorted by a uniform grid of size `cell` (keys from nksr_point_keys with inv_w0 =
 * inv_cell); start/end = nksr_site_ranges of the occupied cells; hkeys/hvals = their hash. */
/* kNN-PCA normals (unoriented): nksr.get_estimate_normal_preprocess_fn, examples/recons_waymo.py:36,
 * recipe examples/recons_waymo_cpu.py:21-41.  valid_out[i]=0 where fewer than k points lie within
 * max_ring cells. */
int nksr_knn_pca_normals(const float* xyz_sorted, int64_t n, const int32_t* start, const int32_t* end,
                         const int64_t* hkeys, const int32_t* hvals, int32_t hcap, float cell, float inv_cell, int k,
                         int max_ring, float* normal_out, float* radius2_out, int32_t* valid_out,
                         int32_t* todo_work /* [n] ints: one wavefront per query (candidates through LDS); NULL: one thread per query */,
                         void* stream);
/* An octree over ONE Morton-sorted cloud (ext.sdfgen, reference ext/common/kdtree_cuda.cu: the role of its kd-tree): level l has cells
 * of size cell * 2^l, its cell keys are the level-0 keys >> 3 l (sorted unique); start / end = the point range of every cell, hkeys /
 * hvals / hcap the key -> cell hash of the level, and for l > 0 child [n_l + 1] / cmask [n_l] = the range of a cell's children on level
 * l - 1 and their occupied octants (nksr_knn_pyramid_level builds start / end / child / cmask of a level from the level below).
 * nksr_sdf_from_points_pyramid / nksr_knn_mean_dist_pyramid = nksr_sdf_from_points / nksr_knn_mean_dist for nb_points <= 32 over
 * EVERY scale in one launch: a query climbs to the first level with a point within one cell of it and takes its k nearest there
 * (kept sorted in registers; cells nearest first, descended with box pruning down to cells of <= leaf points), climbing on only when
 * fewer than k lie within max_ring rings; valid = 0: not even the coarsest level holds k points in reach. */
#define NKSR_KNN_LEVELS 12
typedef struct {
    const float* xyz_sorted;
    const int32_t* start[NKSR_KNN_LEVELS];
    const int32_t* end[NKSR_KNN_LEVELS];
    const int32_t* child[NKSR_KNN_LEVELS];
    const uint8_t* cmask[NKSR_KNN_LEVELS];
    const int64_t* hkeys[NKSR_KNN_LEVELS];
    const int32_t* hvals[NKSR_KNN_LEVELS];
    int32_t hcap[NKSR_KNN_LEVELS];
    int32_t levels, leaf;
    float cell, inv_cell;
} nksr_knn_pyramid_t;
int nksr_knn_pyramid_level(const int64_t* child_keys, int32_t n_child, const int32_t* child_start, const int32_t* child_end,
                           const int64_t* keys, int32_t n, int32_t* child_out, uint8_t* cmask_out, int32_t* start_out, int32_t* end_out,
                           void* stream);
int nksr_sdf_from_points_pyramid(const nksr_knn_pyramid_t* pyramid, const float* normal_sorted, const float* ref_std_sorted, const float* query,
                                 int64_t nq, int k, int max_ring, float stdv, int imls, float* sdf_out, float* grad_out, int32_t* valid_out,
                                 void* stream);
int nksr_knn_mean_dist_pyramid(const nksr_knn_pyramid_t* pyramid, int64_t n, int k, int max_ring, float* out, int32_t* valid_out, void* stream);
/* Signed distance of arbitrary queries to an oriented cloud from their k = nb_points nearest reference points: the training
 * ground truth ext.sdfgen.sdf_from_points(queries, ref_xyz, ref_normal, nb_points, stdv, compute_grad, imls, adaptive_knn)
 * (ext/sdfgen/sdf_from_points.cu:32-235; models/loss.py:85, dataset/av_gt_geometry.py:72).  imls = 0: nearest-neighbour magnitude
 * (|n.(x-p)| inside stdv * ref_std[p], |x-p| outside) with the sign voted by the k neighbours; imls != 0: IMLS weights
 * exp(-|x-p_k|^2 / stdv^2).  ref_std_sorted may be NULL (= 1); grad_out may be NULL.  valid_out[i] = 0: fewer than k points within
 * max_ring cells (retry on a coarser grid).  nksr_knn_mean_dist: mean distance of every reference point to its k nearest
 * (itself included) = ref_std for adaptive_knn = k. */
int nksr_sdf_from_points(const float* xyz_sorted, const float* normal_sorted, const float* ref_std_sorted, const int32_t* start,
                         const int32_t* end, const int64_t* hkeys, const int32_t* hvals, int32_t hcap, float cell, float inv_cell,
                         const float* query, int64_t nq, int k, int max_ring, float stdv, int imls, float* sdf_out, float* grad_out,
                         int32_t* valid_out, void* stream);
int nksr_knn_mean_dist(const float* xyz_sorted, int64_t n, const int32_t* start, const int32_t* end, const int64_t* hkeys,
                       const int32_t* hvals, int32_t hcap, float cell, float inv_cell, int k, int max_ring, float* out,
                       int32_t* valid_out, void* stream);
/* index (into the sorted cloud) of the nearest point of every query: fields.PCNNField,
 * examples/recons_colored_mesh.py:28 */
int nksr_nearest_index(const float* xyz_sorted, const int32_t* start, const int32_t* end, const int64_t* hkeys,
                       const int32_t* hvals, int32_t hcap, float cell, float inv_cell, const float* query, int64_t nq,
                       int max_ring, int32_t* index_out, void* stream);

/* ---- dual marching cubes (field.extract_dual_mesh, examples/recons_simple.py:27) ------- */
/* flags[i]=1 where voxel i and its +x,+y,+z,... 7 partners are all active */
int nksr_base_cell_flags(const int32_t* nbr, int32_t n, int32_t* flags, void* stream);
/* expand selected base voxels into U^3 lattice cell keys */
int nksr_base_cell_keys(const int32_t* ijk, const int32_t* sel, int64_t nsel, int upsample, int64_t* cell_keys, void* stream);
/* Lattice cells covered by the dual cells of the selected level-`level` voxels ((upsample << level)^3 keys each): the part of
 * the dual grid that comes from levels 1 .. adaptive_depth-1 where the finest level is absent
 * (LayerField(dec_svh, adaptive_depth), models/nksr_net.py:132; adaptive_depth 2: configs/carla/train.yaml:6). */
int nksr_level_cell_keys(const int32_t* ijk, const int32_t* sel, int64_t nsel, int level, int upsample, int64_t* cell_keys, void* stream);
/* 8 corner lattice keys per cell */
int nksr_cell_corner_keys(const int64_t* cell_keys, int64_t ncell, int64_t* corner_keys, void* stream);
/* lattice key -> position x = g*h + half_w0 */
int nksr_lattice_positions(const int64_t* vkeys, int64_t n, float h, float half_w0, float* xyz_out, void* stream);
/* sorted-key lower-bound lookup (exact match required; -1 otherwise) */
int nksr_sorted_lookup(const int64_t* sorted, int64_t n, const int64_t* q, int64_t nq, int32_t* idx_out, void* stream);
/* rank of every element of the ASCENDING list q in the ascending list `sorted`: rank_out[i] = #{ sorted < q[i] } (upper == 0) or
 * #{ sorted <= q[i] } (upper != 0) -- the merge of two sorted site lists without sorting them again (KernelField.solve: the shared
 * Morton-ordered row list of the position and the normal sites, models/nksr_net.py:105-112).  n < 2^31. */
int nksr_rank_sorted(const int64_t* sorted, int64_t n, const int64_t* q, int64_t nq, int upper, int32_t* rank_out, void* stream);
/* per cell: 8-bit sign configuration (bit c set iff f[corner c] > 0) and triangle count */
int nksr_cell_config(const int32_t* corner_idx, const float* f, int64_t ncell, int32_t* config, int32_t* ntri, void* stream);
/* flags[i]=1 where the cell's corners do not share a sign (MISE candidates) */
int nksr_cell_active_flags(const int32_t* config, int64_t ncell, int32_t* flags, void* stream);
/* ordered stream compaction: per-256-block counts, then (after an exclusive scan of the
 * counts) ordered scatter of the flagged indices -- wave ballot + popcount prefix */
int nksr_compact_block_counts(const int32_t* flags, int64_t n, int32_t* block_counts, void* stream);
int nksr_compact_scatter(const int32_t* flags, int64_t n, const int32_t* block_offsets, int32_t* sel, void* stream);
/* MISE hanging-vertex constraint: a refined vertex on a coarse edge / face gets the mean of the coarse
 * end points / face corners unless every coarse cell sharing that edge / face was refined (f_fine is updated in
 * place) -- closes T-junction cracks between refined and unrefined cells.  The coarse lattice vertices (key -> index into
 * f_coarse) and the refined coarse cells are given as nksr_hash_build tables. */
int nksr_mise_constrain(const int64_t* vkeys_fine, int64_t nv, float* f_fine, const int64_t* chash_keys, const int32_t* chash_vals,
                        int32_t chash_cap, const float* f_coarse, const int64_t* ahash_keys, const int32_t* ahash_vals,
                        int32_t ahash_cap, void* stream);
/* children of the flagged cells: out has 8 keys per selected cell */
int nksr_cell_children(const int64_t* cell_keys, const int32_t* sel, int64_t nsel, int64_t* child_keys, void* stream);
/* triangle emission: edge keys (lower vertex index*3+axis) [ntri_total,3] */
int nksr_mc_emit(const int32_t* corner_idx, const int32_t* config, const int32_t* tri_offset, int64_t ncell,
                 int64_t* edge_keys, void* stream);
/* mesh vertices from unique edge keys (vhash: nksr_hash_build table of vkeys) */
int nksr_mc_vertices(const int64_t* edge_keys, int64_t nedge, const int64_t* vkeys, const int64_t* vhash_keys, const int32_t* vhash_vals,
                     int32_t vhash_cap, const float* vpos, const float* f, float h, float* verts_out, void* stream);

/* ---- marching cubes on the ADAPTIVE dual graph: cells as large as their hierarchy level (field.extract_dual_mesh with
 * LayerField(dec_svh, adaptive_depth), reference models/nksr_net.py:132,214,284; specification: oracle/dual_adaptive.py) ----------
 * A primal cell is (lam, C): size 2^lam fine units, minimum corner C * 2^lam; its key = Morton(C + 2^20).  The cells of all sizes
 * form ONE table, smallest first, each size sorted by key (id = offset[l] + rank); nksr_cell_table_t names the per-size key hashes
 * (nksr_hash_build). */
#define NKSR_CELL_SIZES 12
typedef struct {
    int32_t nlev;
    int32_t lam[NKSR_CELL_SIZES];
    int32_t offset[NKSR_CELL_SIZES];
    int32_t hcap[NKSR_CELL_SIZES];
    const int64_t* hkeys[NKSR_CELL_SIZES];
    const int32_t* hvals[NKSR_CELL_SIZES];
} nksr_cell_table_t;
/* keys of k - 1 for the eight corners k of every cell of ONE size (fine coordinates): out has 8 keys per cell */
int nksr_adaptive_corner_keys(const int64_t* cell_keys, int64_t ncell, int lam, int64_t* corner_keys_out, void* stream);
/* the dual cell of every corner: cidx_out [ncorner, 8] = id of the cell that contains the fine voxel (k - 1) + o, o = (c >> 2, (c >> 1) & 1,
 * c & 1), or -1 (the smallest size is asked first) */
int nksr_adaptive_dual_cells(const int64_t* corner_keys, int64_t ncorner, const nksr_cell_table_t* table, int32_t* cidx_out, void* stream);
/* cell centres: fl(fl(C 2^lam * u) + 2^lam * (u / 2)) per axis (the lattice positions of nksr_lattice_positions in the uniform case) */
int nksr_adaptive_positions(const int64_t* cell_keys, const int32_t* cell_lam, int64_t ncell, float u, float* xyz_out, void* stream);
/* triangle emission: vertex names (A << 33) | (axis << 31) | B -- the cells the cube edge joins, A on the low side -- [ntri_total, 3] */
int nksr_mc_emit_pairs(const int32_t* corner_idx, const int32_t* config, const int32_t* tri_offset, int64_t ncell, int64_t* pair_keys,
                       void* stream);
/* mesh vertices from unique vertex names: pA + t (pB - pA), t = fA / (fA - fB), pB - pA exact from the integer coordinates */
int nksr_pair_vertices(const int64_t* pair_keys, int64_t npair, const int64_t* cell_keys, const int32_t* cell_lam, const float* cell_pos,
                       const float* f, float u, float* verts_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
