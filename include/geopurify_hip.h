/*
 * geopurify_hip.h -- C ABI of libgeopurify_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for GeoPurify's per-scene hot path.  The reference has no
 * FFI of its own: its hot path calls third-party engines from Python.  Each entry point below
 * replaces one of those call sites (cited as file:line relative to the reference tree); the Python
 * host layer (geopurify_amd/_lib.py) binds them with ctypes, see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all buffers (outputs and workspaces) are caller-allocated; *_workspace_bytes() gives sizes;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream unless
 *     it documents a synchronisation;
 *   - return value: 0 = ok, negative = error (GP_E*), gp_last_error() gives a message;
 *   - no exceptions, no hidden allocation on the launch path, no torch types.
 *   - rows are fp32 row-major with an explicit leading dimension (in elements) where noted.
 */
#ifndef GEOPURIFY_HIP_H
#define GEOPURIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GP_OK 0
#define GP_EINVAL (-22)   /* bad argument / shape */
#define GP_ENOMEM (-12)   /* workspace too small */
#define GP_EHIP (-5)      /* HIP runtime error */
#define GP_ERANGE (-34)   /* coordinate extent not representable */

#define GP_KNN_MAX_K 127  /* K+1 <= 128 */

int gp_version(void);
/* Tuning knobs for experiments (table in csrc/error.hip).  No knob reaches a PRODUCT kernel: the pooling / convolution  */
/* tuning bits (knobs 4 / 3) are honoured only by the *_tuning_kernel twins, the others select a launch shape or a slower, */
/* equally tested kernel.  Unknown keys and undefined values are GP_EINVAL.                                                */
int gp_debug_set(int32_t key, int32_t value);
/* Device buffers for experiments (0: pooling time stamps, 10 x uint64 per wave; NULL, 0 = off).  `bytes` is checked by    */
/* every launch that writes into the buffer.                                                                               */
int gp_debug_ptr(int32_t key, void *device_buffer, size_t bytes);
const char *gp_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* Rows 1-2: Voxelizer.voxelize + FNV-1 dedupe (dataset/voxelizer.py:103-121,                  */
/*           dataset/voxelization_utils.py:6-18,95-97: np.unique(return_index, return_inverse)) */
/* c = floor([x 1] @ rigid^T[:, :3]); c -= min(c); key = FNV-1(c) ; voxels in ascending key      */
/* order.  rigid_host: 16 doubles row-major (M_r @ M_v).  Outputs: coords_aug f64 [N,3] and inds */
/* i64 [N] hold nv valid rows; inds_reconstruct i64 [N]; *nv_dev receives the voxel count;       */
/* order i64 [N] (optional, may be NULL): point ids sorted by (key, id); seg_start i64 [N+1]     */
/* (optional): CSR offsets of each voxel's points in `order`.                                    */
size_t gp_voxelize_workspace_bytes(int64_t n);
int gp_voxelize_f64(const double *coords, int64_t n, const double *rigid_host,
                    double *coords_aug, int64_t *inds, int64_t *inds_reconstruct, int64_t *nv_dev,
                    int64_t *order, int64_t *seg_start,
                    void *workspace, size_t workspace_bytes, void *stream);
/* FNV-1 hash of integer-valued fp64 coordinates [n,3] -> uint64 [n] (voxelization_utils.py:6-18) */
int gp_fnv_hash_f64(const double *coords, int64_t n, uint64_t *hash, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 3: point -> pixel mapping with depth-consistency test                                    */
/* (models/utils/fusion_util.py:99-147 ScanNet, :45-82 Matterport).                              */
/* w2c_host: 16 doubles row-major world->camera (the host passes world_view_transform^T, or     */
/* inv(camera_to_world)); fx,fy,cx,cy: intrinsics at image_dim.  depth f64 [H,W] or NULL.        */
/* mapping i64 [N,3] rows (v,u,1) or (0,0,0).  weight f64 [N] optional (ScanNet variant).         */
int gp_project_points_f64(const double *coords, int64_t n, const double *w2c_host,
                          double fx, double fy, double cx, double cy,
                          const double *depth, int32_t width, int32_t height,
                          int32_t cut_bound, double vis_thres,
                          int64_t *mapping, double *weight, void *stream);
/* Depth "render" mode of the ScanNet mapper (fusion_util.py:126-130, compute_mapping(depth=<str>)):  */
/* depth f64 [H,W] = 999999 everywhere, then the minimum camera-space z over the points with z > 0.2   */
/* that project inside the cut bound; feed it to gp_project_points_f64 for the occlusion test.        */
int gp_render_depth_f64(const double *coords, int64_t n, const double *w2c_host,
                        double fx, double fy, double cx, double cy, int32_t width, int32_t height,
                        int32_t cut_bound, double *depth, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Internal voxel order + lattice grid (shared by kernel-map build and kNN).                     */
/* gp_morton_order: perm i32 [nv] = voxel rows sorted by the Morton code of (coords - min);      */
/* rank i32 [nv] = inverse permutation.  coords i32 [nv,3].                                      */
/* per-axis minimum / maximum of integer coordinates [nv,3] -> mm i32 [6] = (min x,y,z, max x,y,z); no host sync      */
int gp_minmax_i32(const int32_t *coords, int64_t nv, int32_t *mm, void *stream);
size_t gp_morton_order_workspace_bytes(int64_t nv);
int gp_morton_order(const int32_t *coords, int64_t nv, int32_t *perm, int32_t *rank,
                    void *workspace, size_t workspace_bytes, void *stream);
/* gp_grid_build: coords must already be in Morton order (as produced by perm).  The grid lives   */
/* in `grid` (gp_grid_bytes(nv, extent) bytes).  extent_host: 3 ints = max-min+1 per axis,        */
/* origin_host: 3 ints = min per axis.                                                           */
size_t gp_grid_bytes(int64_t nv, const int32_t *extent_host);
int gp_grid_build(const int32_t *coords, int64_t nv, const int32_t *origin_host,
                  const int32_t *extent_host, void *grid, size_t grid_bytes, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 8: torch_scatter.scatter_mean(src, index, dim=0) (models/affinity_module.py:1524-1535).   */
/* CSR form (deterministic, sums in ascending point id): seg_start i64 [nv+1], order i64 [n].     */
/* out[v, col0:col0+d] = mean over the voxel's points.  ld_src / ld_out in floats.                */
int gp_scatter_mean_csr(const float *src, int64_t ld_src, int32_t d, const int64_t *order,
                        const int64_t *seg_start, int64_t nv, const int32_t *row_map,
                        float *out, int64_t ld_out, int32_t col0, void *stream);
/* out[p, 0:d] = src[index[p], 0:d]   (final voxel->point gather, affinity_module.py:1589)        */
int gp_gather_rows(const float *src, int64_t ld_src, int32_t d, const int64_t *index, int64_t n,
                   const int32_t *row_map, float *out, int64_t ld_out, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 9: MinkowskiEngine submanifold convolution (models/affinity_module.py:36-66,1541-1546).   */
/* gp_kernel_map_build: nbr_map i32 [27,nv]; nbr_map[k][u] = row of voxel coords[u]+o_k or -1,   */
/* k = (dx+1)+3(dy+1)+9(dz+1).  Needs the grid built from the same Morton-ordered coords.         */
int gp_kernel_map_build(const void *grid, const int32_t *coords, int64_t nv, int32_t *nbr_map,
                        void *stream);
/* Y[u, :] = epilogue( sum_k X[nbr_map[k][u], :] @ W[k] ),  W fp32 [kv, cin, cout] row-major,      */
/* kv = 27 (nbr_map [27,nv]) or 1 (nbr_map NULL: identity).  epilogue: y = acc*scale + shift      */
/* (scale/shift fp32 [cout] or NULL), optional residual add (fp32 [nv, ld_res] or NULL), optional  */
/* ReLU.  cin % 8 == 0 (pad channels with zeros), cout % 128 == 0.                                */
int gp_sparse_conv(const float *x, int64_t ld_x, const int32_t *nbr_map, int64_t nv,
                   const float *w, int32_t kv, int32_t cin, int32_t cout,
                   const float *scale, const float *shift, const float *residual, int64_t ld_res,
                   int32_t relu, float *y, int64_t ld_y, void *stream);
/* Fast path of the same convolution on the f16 matrix cores with fp32-class accuracy (operands     */
/* split x = hi + lo in f16, products hi*hi + hi*lo + lo*hi accumulated in fp32; relative error of   */
/* a product <= 2^-21).  gp_conv_pairs_build compacts the kernel map once per scene: pair_in i32     */
/* [num_pairs] (input row per pair, ordered by (chunk of output rows, k, output row)),                  */
/* pair_pos i32 [kv,nv] (pair index or -1), seg_off i32 [nseg+1] (first pair of each (chunk,k) segment, */
/* nseg = num_chunks*kv; seg_off[nseg] = num_pairs), tile_start i32 [nseg+1] (256-pair tiles before     */
/* each segment).  The chunks are given by their row offsets chunk_row_off (DEVICE i32 [num_chunks+1], */
/* 0 = first, nv = last, ascending): equal heights, or the heights of gp_conv_chunk_plan.              */
/* gp_conv_weights_split: w fp32 [kv,cin,cout] -> w_hi/w_lo f16 [kv,cout,cin] of scale_pow2 * w.       */
/* gp_sparse_conv_f16x3: `partial` = 4 * num_pairs * cout bytes of workspace (the largest chunk's pairs */
/* when chunked) for the partial rows between the two phases -- fp32 rows on the register-staged path   */
/* (x fp32), 24-bit block floating point on the LDS-DMA path (x_hi / x_lo; 3 bytes per element + one    */
/* exponent byte per (pair row, 128 columns): rounded within 2^-22 of the quarter's largest magnitude,   */
/* the rounding of the f16 hi + lo split that follows; an Inf / NaN activation row makes the output      */
/* rows that gather it NaN -- so does ONE overflowing element of a partial row: its whole 128-column    */
/* quarter is marked, where fp32 rows kept Inf in that element and finite neighbours.  The 24-bit rows   */
/* use 32-bit byte offsets: a call whose largest chunk holds pairs * cout * 3 >= 4 GiB runs on fp32     */
/* partial rows instead, decided before the first launch); its contents are private to the call.        */
/* Epilogue as gp_sparse_conv (the                                                                       */
/* caller folds 1/scale_pow2 into `scale`).  cin % 32 == 0, cout % 256 == 0, |x| < 65504.              */
size_t gp_conv_pairs_workspace_bytes(int64_t nv, int32_t kv);
int gp_conv_pairs_build(const int32_t *nbr_map, int64_t nv, int32_t kv, int32_t num_chunks, const int32_t *chunk_row_off,
                        int32_t *pair_in, int32_t *pair_pos, int32_t *seg_off, int32_t *tile_start,
                        int32_t *tile_desc /* i32 [num_pairs/256 + nseg, 4]: {k, first pair, count, 0} per tile */,
                        void *workspace, size_t workspace_bytes, void *stream);
/* Chunk heights chosen from the kernel map so that every chunk's phase-1 launch -- sum over the offsets of ceil(pairs / 256)  */
/* row tiles, times col_tiles (= cout / 256) column tiles -- stays within target_tiles (the host passes 3 x the CU count - 16:   */
/* three rounds of one-tile workgroups; equal heights leave 4-8 % of the tile slots of their rounds empty).  Chunks close at multiples of        */
/* granule_rows (>= 64).  Outputs on the DEVICE: chunk_row_off i32 [max_chunks + 1], n_chunks i32 [1]; the caller reads them      */
/* back to size the pair arrays and to pass the host copy to gp_sparse_conv_f16x3.                                               */
size_t gp_conv_chunk_plan_workspace_bytes(int64_t nv, int32_t granule_rows);
int gp_conv_chunk_plan(const int32_t *nbr_map, int64_t nv, int32_t kv, int32_t granule_rows, int32_t col_tiles,
                       int32_t target_tiles, int32_t max_chunks, int32_t *chunk_row_off, int32_t *n_chunks, void *workspace,
                       size_t workspace_bytes, void *stream);
int gp_conv_weights_split(const float *w, int32_t kv, int32_t cin, int32_t cout, float scale_pow2,
                          void *w_hi, void *w_lo, void *stream);
/* Optional pre-split operands: x_hi/x_lo f16 [nv, ld_xh] (from gp_split_f16 or a previous layer's   */
/* y_hi/y_lo) select the LDS-DMA staging path (x may then be NULL); y_hi/y_lo f16 [nv, ld_yh] (or      */
/* NULL) receive the split output for the next layer; with them y may be NULL (no fp32 copy).          */
int gp_split_f16(const float *x, int64_t ld_x, int32_t d, int64_t n, void *hi, void *lo, int64_t ld_h,
                 void *stream);
/* Power-of-two pre-scaling of split operands, so that the f16 lo halves stay NORMAL numbers and x = hi + lo holds to  */
/* 2^-22 RELATIVE also for small magnitudes (unscaled, |x| < 2^-3 leaves lo subnormal: absolute error 2^-25).          */
/* gp_pow2_scale: scale2[0] = s = 2^k with amax(|x[0:n, 0:d]|) * s in [2^13, 2^14), scale2[1] = 1/s (device scalars, */
/* no host sync; workspace >= 4 bytes).  gp_split_f16_scaled: hi + lo = x * s with s = *scale (global, nullable) or,   */
/* when row_inv_scale != NULL, a per-row s(row) chosen the same way, row_inv_scale[row] = 1/s(row).  Exact (powers of  */
/* two); consumers multiply back: gp_pool_mfma_apply(out_scale), gp_sparse_conv_f16x3(x_row_inv_scale).                */
/* gp_split_f16_scaled with lo = NULL: INTERLEAVED rows into hi ([d / 32 steps][hi 32 | lo 32], ld_h >= 2 d): the operand  */
/* form of gp_sparse_conv_f16x3's plane_flags bit 0.                                                                      */
int gp_pow2_scale(const float *x, int64_t ld_x, int32_t d, int64_t n, float *scale2, void *workspace,
                  size_t workspace_bytes, void *stream);
/* dst_row (nullable, i32 [n]): row r of x is written to row dst_row[r] of hi / lo / row_inv_scale (a permutation: gp_rcb_order).   */
int gp_split_f16_scaled(const float *x, int64_t ld_x, int32_t d, int64_t n, void *hi, void *lo, int64_t ld_h,
                        const float *scale, float *row_inv_scale, const int32_t *dst_row, void *stream);
int gp_sparse_conv_f16x3(const float *x, int64_t ld_x, const void *x_hi, const void *x_lo, int64_t ld_xh,
                         const int32_t *pair_in, const int32_t *pair_pos,
                         const int32_t *seg_off, const int32_t *tile_start, const int32_t *tile_desc,
                         int32_t nseg, int64_t num_pairs, int64_t nv, int32_t kv,
                         const void *w_hi, const void *w_lo, int32_t cin, int32_t cout, float *partial,
                         const float *scale, const float *shift, const float *residual, int64_t ld_res,
                         int32_t relu, float *y, int64_t ld_y, void *y_hi, void *y_lo, int64_t ld_yh,
                         int32_t num_chunks, const int32_t *chunk_row_off_host, const int32_t *chunk_tile_off_host,
                         const int32_t *chunk_pair_off_host, const float *x_row_inv_scale, float *y_row_inv_scale,
                         const void *res_hi, const void *res_lo, int64_t ld_rh, const float *res_row_inv_scale, int32_t w_blocked,
                         int32_t plane_flags, void *stream);
/* plane_flags bit 4 (16): ONE offset (kv = 1) whose map holds every output row -- a gather-GEMM y[u] = x[pair_in[u]] @ w[0] (the training   */
/* sampler's anchors x points similarity): phase 1 stores fp32 rows straight into y (contiguous, ld_y = cout; no scale / shift / residual / */
/* ReLU / split output) and phase 2 is not run.                                                                                            */
/* plane_flags (mask): 1 = x_hi is ONE tensor of INTERLEAVED rows, [cin / 32 steps][hi 32 | lo 32] halfs per row (ld_xh >= 2 cin; x_lo  */
/* unused): the LDS-DMA kernel then stages a row and K step as ONE full 128-byte line instead of two half lines; 2 = y_hi receives the   */
/* output in that form (y_lo unused; what the next layer reads); 4 = the residual planes res_hi come in that form.  0: separate planes. */
/* w_blocked = 1: w_hi / w_lo come from gp_conv_weights_split_blocked -- the same halves as [kv][cout / 256][cin / 32][256][32]: a K   */
/* step's 16 KiB of a column tile contiguous (one 1-KiB run per LDS-DMA instruction instead of 16 half lines), row rho of a tile =     */
/* its column (rho & 128) | (rho & 15) << 3 | (rho >> 4 & 7).  cin % 32 == 0, cout % 256 == 0.  0: the [kv][cout][cin] halves above.    */
/* transpose_flip = 1: the data-gradient operand V[k] = W[kv - 1 - k]^T straight from the forward layer's w: cin / cout are V's (the      */
/* forward layer's cout / cin), w is read as [kv][cout][cin] with the offsets mirrored -- no flipped, transposed fp32 copy in between.  */
int gp_conv_weights_split_blocked(const float *w, int32_t kv, int32_t cin, int32_t cout, float scale_pow2,
                                  void *w_hi, void *w_lo, int32_t transpose_flip, void *stream);
/* res_hi / res_lo f16 [nv, ld_rh] (+ res_row_inv_scale fp32 [nv] or NULL): the residual as the split planes an earlier layer wrote  */
/* (its y_hi / y_lo / y_row_inv_scale) instead of fp32 rows -- (hi + lo) * inv, the value that layer's consumer multiplied with; the    */
/* producer then needs no fp32 copy (y = NULL).  `residual` and res_hi exclude each other.                                             */
/* Chunked execution (num_chunks >= 1, the chunks the pairs were built with): phase 1 / phase 2 alternate  */
/* per chunk so that `partial` (then sized for the largest chunk) stays in the Infinity Cache;             */
/* chunk_row_off_host = HOST copy of the row offsets, chunk_tile_off_host / chunk_pair_off_host =           */
/* tile_start / seg_off at the chunk boundaries, all [num_chunks+1]; num_chunks = 0: none of them.          */
/* in-place row L2 normalisation, F.normalize(p=2, dim=1, eps=1e-12) (affinity_module.py:1547)     */
int gp_l2norm_rows(float *x, int64_t ld, int32_t d, int64_t n, void *stream);
/* The student's 1x1x1 output convolution (hidden -> 128 embedding channels, affinity_module.py:66,71) on the pre-split rows   */
/* the last 3x3x3 layer writes, fused with the row normalisation above (:1547) when l2_normalize != 0:                        */
/* y[r, :] = out_scale * x_row_inv_scale[r] * sum_c (x_hi + x_lo)[r, c] * (w_hi + w_lo)[:, c]   (three exact f16 products per  */
/* element, fp32 accumulation, fixed order).  w_hi / w_lo f16 [cout, cin] from gp_conv_weights_split(kv = 1, scale 2^k),       */
/* out_scale = 2^-k; x_row_inv_scale nullable (unscaled planes).  cout = 128, cin a multiple of 64.                            */
/* e_hi / e_lo (nullable pair, f16 [nv, 128]): also / instead (y may then be NULL) the rows x plane_scale as hi + lo planes --   */
/* the operand of gp_affinity_cs_fragments (plane_scale 1024), written by the same epilogue instead of a separate split pass.    */
int gp_embed_head_f16x3(const void *x_hi, const void *x_lo, int64_t ld_x, const float *x_row_inv_scale, const void *w_hi,
                        const void *w_lo, int64_t nv, int32_t cin, int32_t cout, float out_scale, int32_t l2_normalize,
                        float *y, int64_t ld_y, void *e_hi, void *e_lo, float plane_scale,
                        const int32_t *e_dst_row /* nullable: plane row of input row r (gp_rcb_order's map); y keeps the input order */,
                        void *stream);

/* Row order of the POOLING operator (row 12's matrix-core kernels gather, per block of 128 consecutive rows, the union of the      */
/* rows' neighbours: compact blocks have smaller unions).  gp_rcb_order: inside chunks of chunk_rows (1024 or 2048) consecutive rows   */
/* of the Morton-ordered integer coords [nv, 3], recursive coordinate bisection into leaves of leaf_rows rows (every leaf but a       */
/* chunk's last is full).  sigma i32 [nv]: new position -> row, rho i32 [nv]: row -> new position (both permutations of 0 .. nv - 1   */
/* that keep every chunk in place).  gp_rows_renumber_i32: out[p, j] = rho[nbr[sigma[p], j]] for i32 [nv, k] neighbour lists.           */
/* No reference counterpart: an internal order, composed away before any output (models/affinity_module.py:1575-1589 sees none).        */
int gp_rcb_order(const int32_t *coords, int64_t nv, int32_t chunk_rows, int32_t leaf_rows, int32_t *sigma, int32_t *rho, void *stream);
int gp_rows_renumber_i32(const int32_t *nbr, int64_t nv, int32_t k, const int32_t *sigma, const int32_t *rho, int32_t *out,
                         void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 10: faiss.IndexFlatL2.search(K+1) on integer voxel coordinates, self dropped              */
/* (models/affinity_module.py:1551-1557).  Exact; canonical tie rule (d^2, id) ascending where id  */
/* = ids[row] (NULL: the row number).  nbr i32 [nv,K] holds ROW numbers of the given arrays,      */
/* emitted in (d^2, id) order.  coords Morton-ordered + grid as above.  K <= GP_KNN_MAX_K.         */
size_t gp_knn_workspace_bytes(int64_t nv);
int gp_knn_lattice(const void *grid, const int32_t *coords, const int32_t *ids, int64_t nv,
                   int32_t k, int32_t *nbr, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 11: cosine affinity + sharpened softmax (models/affinity_module.py:1559-1572).             */
/* w[i,j] = softmax_j( sharpen * <E_i, E_nbr[i,j]> ),  E fp32 [nv, ld_e], first d columns.         */
int gp_affinity_softmax(const float *e, int64_t ld_e, int32_t d, const int32_t *nbr, int32_t k,
                        int64_t nv, float sharpen, float *w, void *stream);
/* The same weights, written to w AND -- x 2^10, split hi + lo -- to element dst[i * k + j] of the pooling operator's fragment   */
/* arrays (dst, wa_hi, wa_lo from gp_pool_cs_structure): the value gp_pool_cs_fill would read back from w, so the operator has  */
/* the same bits and no fill pass runs between the student and the 19 applications.                                          */
int gp_affinity_softmax_scatter(const float *e, int64_t ld_e, int32_t d, const int32_t *nbr, int32_t k, int64_t nv,
                                float sharpen, float *w, const int32_t *dst, void *wa_hi, void *wa_lo, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 12: one application of the row-stochastic affinity operator, torch.sparse.mm(A, X)        */
/* (models/affinity_module.py:1575-1587) in ELL form: Y[i,0:d] = sum_j w[i,j] * X[nbr[i,j],0:d].   */
/* d % 4 == 0, ld_x/ld_y % 4 == 0.  X and Y must not alias.                                       */
int gp_pool_ell(const float *x, int64_t ld_x, const int32_t *nbr, const float *w, int32_t k,
                int64_t nv, int32_t d, float *y, int64_t ld_y, void *stream);

/* Fast path of the same operator, re-blocked once per scene for the 19 applications: tiles of r     */
/* Morton-adjacent rows (r in {4,8,16}); per tile the union of its rows' neighbours and a dense        */
/* [union, r] weight block.  count: tile_off i64 [ntiles+1] (exclusive scan; last = total entries,     */
/* read it back to size u_row i32 [total] and u_w f32 [total, r]); fill; apply = one application.       */
/* apply: d a multiple of 256 (r = 4, 8, 16), or d = 64 with r = 4 or 8 (a tile per wave).               */
size_t gp_pool_tiles_workspace_bytes(int64_t nv, int32_t r);
int gp_pool_tiles_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t r, int64_t *tile_off,
                        void *workspace, size_t workspace_bytes, void *stream);
int gp_pool_tiles_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t r,
                       const int64_t *tile_off, int32_t *u_row, float *u_w, void *stream);
int gp_pool_tiles_apply(const float *x, int64_t ld_x, const int64_t *tile_off, const int32_t *u_row,
                        const float *u_w, int32_t r, int64_t nv, int32_t d, float *y, int64_t ld_y,
                        void *stream);

/* Matrix-core variant (d = 512): blocks of block_rows (64 or 128) rows, the block's neighbour union    */
/* swept in steps of 32 rows on v_mfma_f32_16x16x32_f16 with split operands (x = hi + lo in f16;        */
/* hi*hi + hi*lo + lo*hi accumulated in fp32).  nblocks = ceil(nv / block_rows), nw = block_rows / 16.   */
/* bu_off i64 [nblocks+1] (padded union rows, multiples of 32), bu_n i32 [nblocks] (unpadded sizes),     */
/* bu_row i32 [total], wa_hi/wa_lo f16 [total/32 * nw * 64 * 8] (weights x 2^10 in MFMA A-fragment order).*/
/* apply: x_hi/x_lo f16 [*, ld_x] -> y_hi/y_lo f16 (nullable pair) and/or y_f32 (nullable).              */
/* min_steps: every row block is padded (zero weights) to at least this many 32-row steps; 0 unless the   */
/* operator is built for gp_pool_mfma_apply_persistent, which needs 9.                                    */
size_t gp_pool_mfma_workspace_bytes(int64_t nv, int32_t block_rows);
int gp_pool_mfma_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t block_rows, int32_t min_steps,
                       int64_t *bu_off, int32_t *bu_n, void *workspace, size_t workspace_bytes, void *stream);
int gp_pool_mfma_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t block_rows,
                      const int64_t *bu_off, const int32_t *bu_n, int64_t total_rows, int32_t *bu_row,
                      void *wa_hi, void *wa_lo, void *stream);
int gp_pool_mfma_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off,
                       const int32_t *bu_row, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                       int32_t block_rows, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32,
                       int64_t ld_yf, const float *out_scale, void *stream);
/* Persistent form of the same operator (one 512-thread workgroup per CU walks (row block, 256-column half) tiles;  */
/* both column groups of waves share every staged weight fragment; the LDS-DMA ring stays full across row blocks; */
/* the epilogue stores straight from the accumulators).  Extra requirements: min_steps = min over row blocks of   */
/* (bu_off[b+1]-bu_off[b])/32 must be >= 9; the output holds y_rows >= ceil(nv/block_rows)*block_rows rows (rows  */
/* >= nv receive zeros); exactly one of (y_hi,y_lo) / y_f32.  out_scale: optional device scalar multiplied into  */
/* the fp32 output (undoes a power-of-two pre-scaling of the split operands).  queue: 9 x uint32 of device memory, */
/* zero before the first launch and left zero by every launch (per-XCD tile counters: workgroups claim their next */
/* tile dynamically, which keeps neighbouring tiles together in L2); launches sharing a queue must be stream-     */
/* ordered; NULL = static tile lists.  Replaces the 19 torch.sparse.mm calls of                                    */
/* models/affinity_module.py:1584-1587 like gp_pool_mfma_apply.                                                    */
int gp_pool_mfma_apply_persistent(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off,
                                  const int32_t *bu_row, const void *wa_hi, const void *wa_lo, int64_t nv,
                                  int32_t d, int32_t block_rows, int32_t min_steps, void *y_hi, void *y_lo,
                                  int64_t ld_y, float *y_f32, int64_t ld_yf, int64_t y_rows,
                                  const float *out_scale, uint32_t *queue, void *stream);

/* Column-sliced matrix-core variant (d = 512; the default from round 3 on): blocks of rows x 256-column halves, every     */
/* wave owns all rows x 32 columns; the builder orders a block's union rows by the 16-row groups that use them and         */
/* stores one bit per (32-row step, group): all-zero 16 x 32 weight fragments are neither fetched nor multiplied.        */
/* rows_per_block (16..128, the same value for count, fill and apply; 128 unless there is a reason): the block height.    */
/* nblocks = ceil(nv / rows_per_block).  bu_off i64 [nblocks+1] (padded union rows, multiples of 32), bu_n i32 [nblocks],   */
/* bu_row i32 [total], bu_mask u32 [total/32] (bit g: group g of the step has a non-zero), wa_hi/wa_lo f16               */
/* [total/32 * 8 * 512] (weights x 2^10 in MFMA fragment order; only fragments whose bit is set are defined and read).   */
/* Same numerics and operand conventions as gp_pool_mfma_apply.  Replaces the 19 torch.sparse.mm calls of                */
/* models/affinity_module.py:1575-1589.  gp_pool_cs_count needs the neighbour lists only: a scheduler can run it (and the  */
/* host read-back of bu_off[nblocks] that sizes the arrays) before the affinity weights exist.                            */
size_t gp_pool_cs_workspace_bytes(int64_t nv, int32_t rows_per_block);
/* max_union (device i64, may be NULL): receives the largest block union.  The fill passes take it (max_union argument, 0 = not   */
/* known) to size their LDS tables -- 56 KiB instead of 152 KiB per workgroup on a ScanNet-shaped scene -- together with the      */
/* host read-back of bu_off[nblocks].                                                                                           */
int gp_pool_cs_count(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, int64_t *bu_off, int32_t *bu_n,
                     int64_t *max_union, void *workspace, size_t workspace_bytes, void *stream);
int gp_pool_cs_fill(const int32_t *nbr, const float *w, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                    int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, void *wa_hi, void *wa_lo, void *stream);
/* gp_pool_cs_fill without the weights: bu_row, bu_mask, zeroed fragments and dst i32 [nv, k] = the element of wa_hi / wa_lo that  */
/* (row, neighbour j) owns; gp_affinity_softmax_scatter completes the operator.  Needs the neighbour lists only (a scheduler runs */
/* it ahead, like gp_pool_cs_count).  total_rows * 128 must fit 32 bits (else: gp_pool_cs_fill).                               */
int gp_pool_cs_structure(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                         int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, void *wa_hi, void *wa_lo,
                         int32_t *dst, void *stream);
/* The structure for gp_affinity_cs_fragments: union rows, fragment masks and valid u32 [total_rows / 32 * 128 + 64]: bit p of  */
/* valid[step * 128 + row] = union row 32 step + p of the row's block is one of that row's neighbours (the last 64 words are     */
/* padding).  No fragment is touched.  Needs the neighbour lists only; their ids must be distinct within a row (k-NN lists are).  */
int gp_pool_cs_structure_valid(const int32_t *nbr, int64_t nv, int32_t k, int32_t rows_per_block, const int64_t *bu_off,
                               int64_t total_rows, int32_t max_union, int32_t *bu_row, uint32_t *bu_mask, uint32_t *bu_valid,
                               void *stream);
/* Row 11 (models/affinity_module.py:1559-1572) fused with the operator fill, on the matrix cores: e_hi / e_lo = the unit          */
/* embeddings x 2^10 as f16 planes [nv, 128] (gp_split_f16_scaled, scale 1024); every (row, union row) similarity of a non-empty   */
/* fragment is computed as hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_f16, the valid ones go through the row's softmax           */
/* (x sharpen), and every non-empty fragment of wa_hi / wa_lo is written whole (weights x 2^10, zeros elsewhere).  d = 128, k <= 96. */
/* Produces no [nv, k] weight matrix (gp_affinity_softmax does).                                                                  */
int gp_affinity_cs_fragments(const void *e_hi, const void *e_lo, int64_t nv, int32_t d, int32_t k, float sharpen,
                             const int64_t *bu_off, const int32_t *bu_row, const uint32_t *bu_mask, const uint32_t *bu_valid,
                             int32_t rows_per_block, void *wa_hi, void *wa_lo, void *stream);
int gp_pool_cs_apply(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                     const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                     int32_t rows_per_block, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
                     const float *out_scale, void *stream);
/* The same application through the persistent producer / consumer form of the kernel (cs_engine_kernel, one workgroup per */
/* CU); bit-identical results.  Its own entry point: the choice of kernel is an argument of the call, not process state.   */
int gp_pool_cs_apply_engine(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                            const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                            int32_t rows_per_block, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32, int64_t ld_yf,
                            const float *out_scale, void *stream);
/* One 256-column half (0 or 1) of the same application; the halves are independent (two streams can each carry one).     */
int gp_pool_cs_apply_half(const void *x_hi, const void *x_lo, int64_t ld_x, const int64_t *bu_off, const int32_t *bu_row,
                          const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv, int32_t d,
                          int32_t rows_per_block, int32_t half, void *y_hi, void *y_lo, int64_t ld_y, float *y_f32,
                          int64_t ld_yf, const float *out_scale, void *stream);
/* ALL `applications` (2..65535) of the operator in ONE launch (the T = 19 torch.sparse.mm calls of                          */
/* models/affinity_module.py:1575-1587 as one kernel): application t reads plane set (t even ? x : p) and writes the other   */
/* one, the last one writes y_f32 (x out_scale[0]) only -- the sequence, planes and bits of `applications` calls of           */
/* gp_pool_cs_apply ping-ponging between x and p; x_hi / x_lo are rewritten from application 1 on.  A row block's tile of     */
/* application t starts when the row blocks of its dependency list have published application t - 1 (per-block flags,         */
/* written-through stores, L1-bypassing gathers); workgroups wait only for workgroups with a smaller index.                   */
/*   gp_pool_cs_deps: dep i32 [nblocks * 64] from the operator's structure (bu_off, bu_row), scratch i32 [nblocks].           */
/*   flags u32 [gp_pool_cs_chain_flag_words()]: zeroed ONCE at allocation.  Word 0 is the ABORT word: set to 1 by the kernel   */
/*     if a workgroup waited 2 s for a dependency (the launch drains without computing, outputs invalid); the caller reads it   */
/*     at its next synchronisation point and treats non-zero as an error.                                                     */
/*   epoch: kept by the caller per flags array, each call at least `applications` above the previous call's.                  */
/* One flags array serves one launch at a time (do not share it between streams).                                             */
int gp_pool_cs_deps(const int64_t *bu_off, const int32_t *bu_row, int64_t nv, int32_t rows_per_block, int32_t *dep,
                    int32_t *scratch, void *stream);
size_t gp_pool_cs_chain_flag_words(int64_t nv, int32_t rows_per_block);
int gp_pool_cs_apply_chain(void *x_hi, void *x_lo, void *p_hi, void *p_lo, int64_t ld, const int64_t *bu_off,
                           const int32_t *bu_row, const uint32_t *bu_mask, const void *wa_hi, const void *wa_lo, int64_t nv,
                           int32_t d, int32_t rows_per_block, int32_t applications, float *y_f32, int64_t ld_yf,
                           const float *out_scale, const int32_t *dep, uint32_t *flags, uint32_t epoch, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Rows 5-7: 2D->3D lift (models/affinity_module.py:416-449, 495-646, 647-696).                   */
/* gp_lift_dense_accum: sum[pt[i], :] += feat2d[:, x[i], y[i]] ; cnt[pt[i]] += 1 for one view.     */
int gp_lift_dense_accum(const float *feat2d, int32_t d, int32_t height, int32_t width,
                        const int64_t *pt, const int64_t *x, const int64_t *y, int64_t n_v,
                        float *sum, int64_t ld_sum, float *cnt, void *stream);
/* gp_lift_dense_bilinear_accum: the LSeg path (affinity_module.py:404-433): feat f32 [d,h,w] is the network's    */
/* low-resolution map; the reference's F.interpolate(bilinear, align_corners=True) to [out_h,out_w] is evaluated   */
/* only at the sampled pixels (x = row, y = col of the full-size image), bit-identical to torch's CPU kernel.      */
int gp_lift_dense_bilinear_accum(const float *feat, int32_t d, int32_t h, int32_t w, int32_t out_h, int32_t out_w,
                                 const int64_t *pt, const int64_t *x, const int64_t *y, int64_t n_v,
                                 float *sum, int64_t ld_sum, float *cnt, void *stream);
/* gp_lift_dense_finish: out = sum / (cnt==0 ? 1e-6 : cnt); seen[p] = cnt > 1e-5 (u8).             */
int gp_lift_dense_finish(float *sum, int64_t ld_sum, int32_t d, const float *cnt, int64_t n,
                         uint8_t *seen, void *stream);
/* gp_lift_masks_view: per visible point pick the segment k* = argmax_q score[q]*sigmoid(m_q(x,y)) */
/* over queries with score>0, where m_q is the bicubic-antialias resize of pred_masks [Q,h,w] to    */
/* mask_shape evaluated only at the sampled pixel (separable taps tap_x0/tap_wx [W,4],              */
/* tap_y0/tap_wy [H,4] precomputed on the host); seg[i] = k* if sigmoid(m_k*) >= 0.5 else -1.       */
size_t gp_lift_masks_workspace_bytes(int32_t q, int32_t h, int32_t w);
int gp_lift_masks_view(const float *pred_masks, int32_t q, int32_t h, int32_t w,
                       const float *scores, const int32_t *tap_x0, const float *tap_wx,
                       const int32_t *tap_y0, const float *tap_wy, int32_t out_h, int32_t out_w,
                       const int64_t *x, const int64_t *y, int64_t n_v, int32_t *seg,
                       float *seg_logit, void *workspace, size_t workspace_bytes, void *stream);
/* gp_lift_masks_views: rows 6-7 up to the point -> (view, segment) lists for ALL views of a scene at once (entries      */
/* from gp_views_visible_lists; pred_masks f32 [nsrc,Q,h,w] and scores f32 [nsrc,Q] stacked over the source views,        */
/* nviews <= nsrc, nviews <= 128).  Per entry the segment of gp_lift_masks_view; then the in-view fill                    */
/* (affinity_module.py:604-625: an entry without a segment takes the one of the nearest entry with a segment in the       */
/* same view, (fp64 squared distance of the fp32 xyz [n,3], entry order) minimum = gp_nn1_masked_f64's rule); then the     */
/* CSR pv_start i64 [n+1], pv_view / pv_seg i32 [total] that gp_fuse_views_top3 reads, a point's entries in ascending      */
/* view order (= gp_pv_count + scan + gp_pv_fill called view by view).  seg i32 [total] out.  Entries of views with       */
/* keep == 0 take no part.  fill_cap: capacity, in fill queries (entries without a segment), of the in-view fill's       */
/* partial-result arrays: 1..total, or 0 = total.  ANY value gives the same results -- queries beyond the capacity are     */
/* answered by one block per 256 of them sweeping the view's whole reference range -- a capacity near the expected query   */
/* count keeps the fill fully parallel and the workspace small (192 B per unit of capacity).                               */
/* Preconditions: Q <= 1024 (a view's scores are sorted in LDS; more: GP_EINVAL -- lift view by view with                  */
/* gp_lift_masks_view, which takes any Q), and scores >= 0 (they are softmax maxima in the reference, :544): the kernel      */
/* visits a pixel's queries in descending score order and stops once the next 64 scores lie below the best                   */
/* score x sigmoid(logit) found, which bounds a candidate only while sigmoid <= 1 multiplies a non-negative score.            */
size_t gp_lift_masks_views_workspace_bytes(int32_t nsrc, int32_t q, int32_t h, int32_t w, int64_t total, int64_t n,
                                           int64_t fill_cap);
int gp_lift_masks_views(const float *pred_masks, int32_t nsrc, int32_t q, int32_t h, int32_t w, const float *scores,
                        const int32_t *tap_x0, const float *tap_wx, const int32_t *tap_y0, const float *tap_wy,
                        int32_t out_h, int32_t out_w, const float *xyz, int64_t n, const int64_t *ent_pt,
                        const int64_t *ent_x, const int64_t *ent_y, const int32_t *ent_view, const int64_t *view_off,
                        const uint8_t *keep, int32_t nviews, int64_t total, int64_t fill_cap, int32_t *seg, int64_t *pv_start,
                        int32_t *pv_view, int32_t *pv_seg, void *workspace, size_t workspace_bytes, void *stream);
/* gp_segment_tables: per view, f_seg[q,:] = normalize(mask_embed[q,:]) and                         */
/* logit_seg[q,c] = logit_scale * <f_seg[q], normalize(text[c])>  (affinity_module.py:627-630;       */
/* every point feature is a segment embedding).                                     */
int gp_segment_tables(const float *mask_embed, int32_t q, int32_t d, const float *text_norm,
                      int32_t c, float logit_scale, float *f_seg, float *logit_seg, void *stream);
/* point -> (view, segment) CSR, replacing the reference's per-point Python dict                       */
/* (affinity_module.py:633-639): gp_pv_count adds 1 to cnt[pt[i]] for one view; after all views an    */
/* exclusive scan gives pv_start; gp_pv_fill appends (view, seg[i]) to every visible point's slot      */
/* list (views must be filled in ascending order on one stream; cursor i32 [n] zero-initialised).     */
int gp_pv_count(const int64_t *pt, int64_t n_v, int64_t *cnt, void *stream);
size_t gp_scan_workspace_bytes(int64_t n);
int gp_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, void *workspace,
                          size_t workspace_bytes, void *stream);
int gp_pv_fill(const int64_t *pt, const int32_t *seg, int64_t n_v, int32_t view,
               const int64_t *pv_start, int32_t *cursor, int32_t *pv_view, int32_t *pv_seg,
               void *stream);
/* gp_fuse_views_top3: CSR over points (pv_start i64 [n+1]; pv_view i32, pv_seg i32 per slot in      */
/* ascending view order; seg -1 = zero feature): consensus class = argmax of the mean logits,        */
/* top-min(M,3) views by agreement, softmax weights, weighted sum of segment features                */
/* f_seg fp32 [V,Q,d], logit_seg fp32 [V,Q,c].  out fp32 [n, ld_out]; seen u8 [n].                    */
int gp_fuse_views_top3(const int64_t *pv_start, const int32_t *pv_view, const int32_t *pv_seg,
                       int64_t n, const float *f_seg, const float *logit_seg, int32_t q, int32_t d,
                       int32_t c, float *out, int64_t ld_out, uint8_t *seen, void *stream);
/* gp_nn1_fill_f64: exact 1-NN (fp64 distances on fp32 coordinates, lowest index on ties) from       */
/* query points to reference points; writes nn[i] = index into ref.  (sklearn KDTree k=1,           */
/* affinity_module.py:619-625,693-696; run/validation.py:425-430.)                                   */
size_t gp_nn1_workspace_bytes(int64_t n_ref, int64_t n_query);
int gp_nn1_f64(const float *ref_xyz, int64_t n_ref, const float *query_xyz, int64_t n_query,
               int64_t *nn, void *workspace, size_t workspace_bytes, void *stream);
/* masked form (no host synchronisation): references = points with ref_mask != 0, queries = points  */
/* with query_mask != 0, both subsets of xyz fp32 [n,3]; nn i64 [n] = index into xyz for queries,    */
/* -1 elsewhere.  If there is no reference point nn stays -1 everywhere.                            */
size_t gp_nn1_masked_workspace_bytes(int64_t n);
int gp_nn1_masked_f64(const float *xyz, int64_t n, const uint8_t *ref_mask, const uint8_t *query_mask,
                      int64_t *nn, void *workspace, size_t workspace_bytes, void *stream);
/* Loader glue (dataset/data_loader_ablation.py:257-264,348-351): ordered lists of the visible      */
/* points of one view from its mapping i64 [n,3]: pt (ascending ids), x = pixel row, y = pixel col;  */
/* *count_dev receives n_v.  Outputs must hold n entries.                                           */
size_t gp_visible_lists_workspace_bytes(int64_t n);
int gp_visible_lists(const int64_t *mapping, int64_t n, int64_t *pt, int64_t *x, int64_t *y,
                     int64_t *count_dev, void *workspace, size_t workspace_bytes, void *stream);
/* The same for ALL views of a scene in four launches (instead of V x {gp_project_points_f64, gp_visible_lists}):      */
/* params f64 [V,20] on the DEVICE = row-major world->camera matrix (16) | fx fy cx cy; depth f64 [V,H,W] or NULL.      */
/* Entries (view, point, pixel row, pixel col) come out view-major, ascending point id inside a view -- the               */
/* concatenation of the per-view lists: ent_pt / ent_x / ent_y i64 and ent_view i32 must hold V*n entries; view_off i64   */
/* [V+1] = first entry of each view (+ total); keep u8 [V] = the loader's view-drop rule (data_loader_ablation.py:       */
/* 254-255, 280-288): n_v != 0, n_v >= min_visible, n_v <= val_keep.  V <= 65535, V*n < 2^31.                            */
size_t gp_views_visible_lists_workspace_bytes(int64_t n, int32_t nviews);
int gp_views_visible_lists(const double *coords, int64_t n, const double *params, const double *depth, int32_t nviews,
                           int32_t width, int32_t height, int32_t cut_bound, double vis_thres, int64_t min_visible,
                           int64_t val_keep, int64_t *ent_pt, int64_t *ent_x, int64_t *ent_y, int32_t *ent_view,
                           int64_t *view_off, uint8_t *keep, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Row 13 + caller tail (run/validation.py:413-416, util/util.py:160-177).                        */
/* pred[p] = argmax_c <normalize(F[p]), text_norm[c]> (first max on ties); zero_row[p] = sum|F|==0  */
int gp_classify_argmax(const float *feat, int64_t ld, int32_t d, int64_t n, const float *text_norm,
                       int32_t c, float logit_scale, int64_t *pred, uint8_t *zero_row, void *stream);
/* gp_gather_rows (the final voxel -> point gather, affinity_module.py:1589) and gp_classify_argmax (run/validation.py:413-416) in   */
/* ONE pass: out[p, 0:d] = src[row_map[index[p]], 0:d] and pred / zero_row of those rows -- the per-point matrix is written once    */
/* and not read back.  Same bits and labels as the two calls.  d a multiple of 64 up to 512, c * d * 4 <= 64 KiB.                  */
int gp_gather_rows_classify(const float *src, int64_t ld_src, int32_t d, const int64_t *index, int64_t n, const int32_t *row_map,
                            float *out, int64_t ld_out, const float *text_norm, int32_t c, float logit_scale, int64_t *pred,
                            uint8_t *zero_row, void *stream);
/* arg-max over the first c columns of logits rows fp32 [n, ld] (e.g. gp_sparse_conv with kv=1 as the   */
/* exact-fp32 MFMA GEMM F @ T^T); zero_row from feat (optional).                                      */
int gp_rows_argmax(const float *logits, int64_t ld, int32_t c, int64_t n, const float *feat, int64_t ld_f,
                   int32_t d, int64_t *pred, uint8_t *zero_row, void *stream);
/* counts i64 [3,C] += (intersection, output, target) histograms with the ignore-id overwrite.      */
int gp_iou_hist_i64(const int64_t *pred, const int64_t *target, int64_t n, int32_t num_classes,
                    const int64_t *ignore_ids_host, int32_t num_ignore, int64_t *counts,
                    void *stream);

/* ------------------------------------------------------------------------------------------ */
/* training step of the student (models/affinity_module.py:1138-1237,                */
/* run/train.py:188-198,346-353).  Convolutions forward / dgrad reuse gp_sparse_conv_f16x3 (dgrad   */
/* = the same operator with weights V[k] = W[26-k]^T); these are the remaining pieces.              */
/* gp_col_stats: mean / biased variance of the rows (BatchNorm1d in training mode over voxel rows): ONE sweep, fp64 sums of x and  */
/* x^2 in a fixed order, var = max(E[x^2] - mean^2, 0) in fp64 (53-bit sums of 24-bit data: the cancellation costs                 */
/* log2(mean^2 / var) of ~29 spare bits).                                                                                            */
size_t gp_col_stats_workspace_bytes(int64_t nv, int32_t c);
int gp_col_stats(const float *y, int64_t ld, int64_t nv, int32_t c, float *mean, float *var,
                 void *workspace, size_t workspace_bytes, void *stream);
/* out = [relu]((y-mean)/sqrt(var+eps)*gamma + beta [+ residual]) (out nullable when the planes are asked for); out_hi/out_lo: optional split f16 */
/* copy for the next convolution; running_mean/var (nullable) <- (1-m)*running + m*batch (unbiased var). */
int gp_bn_train_apply(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, const float *var,
                      const float *gamma, const float *beta, float eps, const float *residual, int64_t ld_res,
                      int32_t relu, float *out, int64_t ld_out, void *out_hi, void *out_lo, int64_t ld_split,
                      float momentum, float *running_mean, float *running_var, void *stream);
/* dz = dout*mask; dgamma = sum dz*xhat; dbeta = sum dz.  mask: act > 0 (act = the layer's fp32 output); or act NULL and beta_mask given -- a */
/* layer without a residual -- recomputed from y as (y-mean)/sqrt(var+eps)*gamma + beta_mask > 0, the value gp_bn_train_apply evaluated (the  */
/* same float: the activation is then neither read here nor, with out = NULL in gp_bn_train_apply, ever written); both NULL: no mask.        */
/* dy = gamma/sqrt(var+eps)*(dz - dbeta/nv - xhat*dgamma/nv); dz_out (nullable) <- dz.                */
/* dy_scale2 (nullable, 2 floats on the device) <- [s, 1/s] of gp_pow2_scale(dy), taken inside the sweep that writes dy: */
/* the scale of the gradient's f16 split (gp_split_f16_scaled) without another pass over it.                           */
/* SPLIT FORM (dy = NULL, dy_hi / dy_lo f16 [nv + 1, ld_h] and dy_scale2 given; c % 4 == 0, 16-byte aligned rows): the sweep writes    */
/* hi + lo = dy * s itself, row nv zeroed (the weight gradient's padded pairs), dy_scale2 = [s, 1/s] with s the power of two of a BOUND  */
/* of max |dy| -- |gamma| / std x (max |dz| + |sum dz| / n + max |xhat| |sum dz xhat| / n) per column, from maxima taken in the           */
/* reduction pass -- so that no fp32 dy is written and no separate split pass reads it (bound / true maximum: a small factor).          */
/* workspace: gp_bn_train_backward_workspace_bytes(nv, c).                                            */
size_t gp_bn_train_backward_workspace_bytes(int64_t nv, int32_t c);
int gp_bn_train_backward(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y,
                         int64_t ld_y, const float *mean, const float *var, float eps, const float *gamma, const float *beta_mask,
                         int64_t nv, int32_t c, float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz,
                         float *dgamma, float *dbeta, float *dy_scale2, void *dy_hi, void *dy_lo, int64_t ld_h,
                         void *workspace, size_t workspace_bytes, void *stream);
/* SyncBatchNorm pieces (run/train.py:212-213 converts the student to MinkowskiSyncBatchNorm; geopurify_amd/sharding.py     */
/* all-reduces these small vectors over the ranks).  gp_col_sums_f64: mean == NULL -> out[col] = sum_r y[r][col], else         */
/* out[col] = sum_r (y[r][col] - mean[col])^2 (fp64, fixed order).  gp_bn_bwd_sums_f64: sums[0:c] = sum dz,                    */
/* sums[c:2c] = sum dz * xhat.  gp_bn_bwd_apply: the dy formula of gp_bn_train_backward with caller-supplied fp32 sums        */
/* [2c] and the row count n_total they were taken over.  workspace: gp_col_stats_workspace_bytes(nv, c).                     */
int gp_col_sums_f64(const float *y, int64_t ld, int64_t nv, int32_t c, const float *mean, double *out, void *workspace,
                    size_t workspace_bytes, void *stream);
int gp_bn_bwd_sums_f64(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                       const float *mean, const float *var, float eps, const float *gamma_mask, const float *beta_mask, int64_t nv, int32_t c,
                       double *sums, void *workspace, size_t workspace_bytes, void *stream);
int gp_bn_bwd_apply(const float *dout, int64_t ld_dout, const float *act, int64_t ld_act, const float *y, int64_t ld_y,
                    const float *mean, const float *var, float eps, const float *gamma, const float *beta_mask, const float *sums,
                    int64_t n_total, int64_t nv, int32_t c, float *dy, int64_t ld_dy, float *dz_out, int64_t ld_dz, float *dy_scale2,
                    void *stream);
/* InfoNCE (affinity_module.py:1219-1233) forward + backward: samples s -> voxel rows sample_to_voxel[s]; */
/* point_to_batch i64 [A*(2+Nn)] = sample ids of (anchors | positives | negatives row-major).           */
/* loss f32 device scalar; de f32 [nv, d] = d loss / d e (overwritten).                                  */
size_t gp_infonce_workspace_bytes(int64_t num_samples, int32_t d);
int gp_infonce_fwd_bwd(const float *e, int64_t ld_e, int64_t nv, int32_t d, const int64_t *sample_to_voxel,
                       int64_t num_samples, const int64_t *point_to_batch, int64_t num_anchors,
                       int32_t num_negatives, float temperature, float *loss, float *de, int64_t ld_de,
                       void *workspace, size_t workspace_bytes, void *stream);
/* Weight gradient of a 3x3x3 layer on the matrix cores: dW[k] = X[in_k]^T dY[out_k] (f16 hi/lo operands, fp32    */
/* accumulate).  x_hi/x_lo f16 [nv, ld_x >= cin_pad]; y_hi/y_lo f16 [nv+1, ld_y >= cout] with row nv all zero;       */
/* pair_in/pair_out i32: per offset its (input row, output row) pairs padded to a multiple of 32 with (0, nv);      */
/* segs i32 [num_segments,4] = {offset, first step (32 pairs), steps, 0} ordered by offset; seg_off i32 [kv+1].     */
/* dw f32 [kv, cin_out, cout] = inv_scale[0] * gradient (inv_scale: device scalar, nullable).                       */
size_t gp_conv_wgrad_workspace_bytes(int64_t num_segments, int32_t cin_pad, int32_t cout);
int gp_conv_wgrad_f16x3(const void *x_hi, const void *x_lo, int64_t ld_x, const void *y_hi, const void *y_lo,
                        int64_t ld_y, const int32_t *pair_in, const int32_t *pair_out, const int32_t *segs,
                        int64_t num_segments, const int32_t *seg_off, int32_t kv, int32_t cin_pad, int32_t cin_out,
                        int32_t cout, const float *inv_scale, float *dw, void *workspace, size_t workspace_bytes,
                        void *stream);
/* torch.optim.AdamW update of one flat fp32 tensor (run/train.py:198), step >= 1.                      */
int gp_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step, void *stream);
/* K nearest other points of each query row (faiss.IndexFlatL2.search(K+1)[:,1:], affinity_module.py:1157-1166), */
/* order (d^2 in fp64 of the fp32 coordinates, row id); *flag_dev != 0: degenerate duplicates, result invalid.   */
int gp_knn_points_f32(const float *xyz, int64_t n, const int64_t *queries, int64_t num_queries, int32_t k,
                      int64_t *out, int32_t *flag_dev, void *stream);
/* The sampler's selections on the anchors x points similarity (sample_contrastive_pairs_hybrid, affinity_module.py:1116-1124):    */
/* per row a of sim fp32 [num_anchors, >= n] (leading dimension ld): positive[a] = arg-max over the points other than anchor_idx[a]   */
/* (ties: the lowest index -- torch.argmax after the -inf mark); macro[a, 0:k] = the k points of lowest similarity other than the     */
/* anchor and the positive, ascending by (value, index) (torch.topk(largest=False) after the two +inf marks).  sim is not written.    */
/* 1 <= k < 1024, k + 2 <= n <= 3 145 728 (12 288 groups of at most 256 elements in LDS).  -0 counts as +0; NaNs order above +inf.        */
int gp_sampler_select(const float *sim, int64_t ld, int64_t num_anchors, int64_t n, const int64_t *anchor_idx, int32_t k,
                      int64_t *positive, int64_t *macro, void *stream);
/* F.normalize(x, p=2, dim=1) (affinity_module.py:1114) written as the f16 hi/lo planes of the similarity GEMM's operands:            */
/* hi + lo = x[r] / max(|x[r]|_2, eps) for r < n; rows n <= r < n_pad of the planes are zero.  d % 4 == 0, x 16-byte aligned.         */
int gp_normalize_split_f16(const float *x, int64_t ld_x, int32_t d, int64_t n, int64_t n_pad, float eps, void *hi, void *lo,
                           int64_t ld_h, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* decode of a fused-feature file on the device (dataset/feature_loader.py:113-192). */
/* feat [feat_rows, row_bytes] holds one row per True of mask_chunk u8 [n], in point order (fp16 or  */
/* fp32 elements: rows are copied as bytes); row_keep u8 [feat_rows] (nullable: the three-key form's  */
/* "mask"); vox_ind i64 [nv] = representative point of each voxel.  With p = vox_ind[v],             */
/* in = mask_chunk[p], r = #True in mask_chunk[0..p), keep = in && (row_keep ? row_keep[r] : 1):      */
/*   mode 0 (training forms):   mask_out[v] = keep; out[j] = feat[r] for the j-th kept voxel (compact, */
/*                              voxel order).                                                         */
/*   mode 1 (evaluation forms): mask_out[v] = keep; out[v] = in ? feat[r] : 0 for every voxel.        */
/* out holds nv rows in both modes.  n_sel (device i64 [2]): [0] = number of kept voxels, [1] = number */
/* of True in mask_chunk = the rows a well-formed file holds: the caller MUST compare it with         */
/* feat_rows (the reference's host code raises on a file whose mask and row count disagree; the       */
/* kernel alone would only mask the rows beyond the file's end).                                      */
size_t gp_fused_decode_workspace_bytes(int64_t n, int64_t nv);
int gp_fused_decode(const uint8_t *mask_chunk, int64_t n, const uint8_t *row_keep, const void *feat,
                    int64_t feat_rows, int64_t row_bytes, const int64_t *vox_ind, int64_t nv, int32_t mode,
                    void *out, uint8_t *mask_out, int64_t *n_sel, void *workspace, size_t workspace_bytes,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOPURIFY_HIP_H */
