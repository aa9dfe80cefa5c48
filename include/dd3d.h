/* libdd3d_hip.so -- C ABI of the MI355X (gfx950) kernels behind dd3d's gradient-matching step.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless the comment says HOST; the caller owns every buffer;
 *   - `stream` is a hipStream_t passed as void*; every call is stream-ordered, re-entrant and allocates nothing;
 *   - variable-size outputs are sized by their bound and the true count is written to a device int (`*_dev`) that the next
 *     call can read, so a chain of coordinate maps needs no host round trip;
 *   - scratch comes from the caller: `*_ws_bytes()` gives the size, `ws` must be 256-byte aligned;
 *   - return value: 0 = ok, < 0 = argument error (-1 bad argument, -2 size out of range, -3 workspace too small),
 *     > 0 = hipError_t.  Nothing throws, nothing calls exit().
 *   - feature rows are fp32 row-major; channel count C is 128 wherever stated; coordinates are int32 rows (b, c1, c2, c3).
 *
 * "replaces" cites the reference call site (paths relative to kghandour/dd3d) whose MinkowskiEngine / torch work the entry
 * point takes over.  MinkowskiEngine 0.5.4 itself is a third-party dependency of the reference (environment.yml:278).
 */
#ifndef DD3D_H
#define DD3D_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int dd3d_version(void);

/* ---------------------------------------------------------------------------------------------------------------------
 * (1) coordinate quantisation, voxel hash, kernel maps -- integer work, bit-exact
 * ------------------------------------------------------------------------------------------------------------------- */

/* power-of-two hash capacity for n keys (>= 2n) and the scratch needed by dd3d_unique_rows */
int64_t dd3d_table_capacity(int64_t n);
int64_t dd3d_unique_ws_bytes(int64_t n, int64_t cap);

/* q[N,4] = floor(coords[N,4]).  replaces: the quantisation inside ME.TensorField.sparse(), models/MEConv.py:247,199 */
int dd3d_floor_coords(const float* coords, int64_t N, int32_t* q, void* stream);

/* q[r] = (b, floor(c/stride)*stride ...) for the first *m_dev (or `bound` if null) rows.
 * replaces: the stride-2 coordinate map of ME.MinkowskiMaxPooling(kernel_size=2, stride=2), models/MEConv.py:227 */
int dd3d_stride_coords(const int32_t* C, const int32_t* m_dev, int64_t bound, int stride, int32_t* q, void* stream);

/* Unique rows of q in FIRST-OCCURRENCE order (winner = minimum input row).  Outputs: C[bound,4] (first *count_dev rows valid),
 * inv[bound] (voxel row of every input row), uidx[bound] (input row of every voxel), table[cap] (coordinate -> voxel row hash,
 * linear probing, -1 empty; reused by dd3d_kernel_map).  replaces: CoordinateMapManager insert/unique behind x.sparse(),
 * models/MEConv.py:247 */
int dd3d_unique_rows(const int32_t* q, const int32_t* n_dev, int64_t bound, int32_t* table, int64_t cap, int32_t* C, int32_t* inv,
                     int32_t* uidx, int32_t* count_dev, void* ws, int64_t ws_bytes, void* stream);

/* floor + unique in one call (q[N,4] is scratch that receives the floored coordinates) */
int dd3d_quantize_unique(const float* coords, int64_t N, int32_t* q, int32_t* table, int64_t cap, int32_t* C, int32_t* inv,
                         int32_t* uidx, int32_t* count_dev, void* ws, int64_t ws_bytes, void* stream);

/* per-voxel point lists, points ascending: vox_off[bound+1], vox_pts[bound] */
int64_t dd3d_csr_ws_bytes(int64_t n);
int dd3d_voxel_csr(const int32_t* inv, const int32_t* n_dev, int64_t bound, const int32_t* m_dev, int32_t* vox_off, int32_t* vox_pts,
                   void* ws, int64_t ws_bytes, void* stream);

/* nbr[K][ld]: row of `keys` found through `table` at Cq[o] + offset_k, or -1.  kernel_size 3: offsets {-1,0,1}*tensor_stride,
 * k = (d0+1) + 3(d1+1) + 9(d2+1), d0 = first spatial column; kernel_size 2: offsets {0,1}*tensor_stride, k = e0 + 2 e1 + 4 e2.
 * replaces: CoordinateMapManager.kernel_map behind ME.MinkowskiConvolution(kernel_size=3) and ME.MinkowskiMaxPooling,
 * models/MEConv.py:224,227 */
int dd3d_kernel_map(const int32_t* Cq, const int32_t* mq_dev, int64_t bound_q, const int32_t* table, int64_t cap, const int32_t* keys,
                    int kernel_size, int tensor_stride, int32_t* nbr, int64_t ld, void* stream);

/* canonical pair lists of a neighbour table: grouped by offset, sorted by out_row.  offs[K+1]; in_rows/out_rows hold <= K*bound */
int64_t dd3d_pairs_ws_bytes(int64_t bound, int K);
int dd3d_kernel_map_pairs(const int32_t* nbr, int64_t ld, const int32_t* m_dev, int64_t bound, int K, int32_t* in_rows, int32_t* out_rows,
                          int32_t* offs, void* ws, int64_t ws_bytes, void* stream);

/* seg_off[B+1]: first row of every cloud (rows must be grouped by ascending batch index, else *err_dev |= 1) */
int dd3d_segment_offsets(const int32_t* C, const int32_t* m_dev, int64_t bound, int B, int32_t* seg_off, int32_t* err_dev, void* stream);

/* Placement used by SparseTensor.dense(shape=[B,128,E0,E1,E2], min_coordinate=batch minimum), models/MEConv.py:250-254:
 * gmin[G,3] per-group minimum corner, site[v] = flat (E0,E1,E2) index of voxel v, vox_at[E0*E1*E2][B] inverse table (site-major, nullable).
 * cloud_group[B] (nullable = one group).  *err_dev |= 2 if a voxel falls outside the extent (ME raises there). */
int dd3d_dense_sites(const int32_t* C, const int32_t* m_dev, int64_t bound, const int32_t* cloud_group, int G, int B, int tensor_stride,
                     int E0, int E1, int E2, int32_t* gmin, int32_t* site, int32_t* vox_at, int32_t* err_dev, void* stream);

/* Collated real batch straight from a point pool resident in HBM: cloud b = pool cloud ids[b] (its first `cap` points, pool
 * order), rows (cloud_base + b, p) into coords[bound,4] and p into feats[bound,3] starting at row point_base, so a batch can be
 * appended behind clouds already in the buffers; pt_off = offsets of the WHOLE batch, entries [cloud_base, cloud_base+B] are
 * written; *n_dev = point_base + rows written.  *err_dev |= 4 if an id is outside [0,n_pool) or the rows do not fit into `bound`.
 * pool_xyz[.,3] / pool_off[n_pool+1] are built once per dataset.
 * replaces: get_images -> Mnist2Dreal.__getitem__ (argwhere(image==1)[:num_points]) -> minkowski_collate_fn / sparse_collate,
 * mnist_distill.py:200-208, utils/Mnist2D.py:50-59, utils/MinkowskiCollate.py:4-15 */
int dd3d_gather_clouds(const float* pool_xyz, const int32_t* pool_off, int n_pool, const int32_t* ids, int B, int cap, int64_t bound, int cloud_base,
                       int64_t point_base, float quant /* coords = p / quant (create_input_batch's quantization_size), feats = p */, float* coords, float* feats, int32_t* pt_off, int32_t* n_dev, int32_t* err_dev, void* stream);
/* dd3d_syn_head + dd3d_gather_clouds in ONE launch (what a captured step runs first): the gathering workgroups sum the point counts of
 * the clouds before them themselves (no one-workgroup offsets kernel), and `zero_words` ints at `zero` (nullable) are cleared on the way
 * (the map builder's per-step count / error words).  B >= 1.  Same outputs as the two calls, bit for bit.
 * replaces: get_images + Mnist2Dreal + minkowski_collate_fn for the real batch, clamp + get_syn_tensor + Mnist2Dsyn for the synthetic
 * one, mnist_distill.py:200-208,333-343, utils/Mnist2D.py:60-92 */
int dd3d_assemble_batch(float* xyz, int64_t n_syn_clouds, int npts, const int32_t* dest_cloud, int do_clamp, float lo, float hi, int zero_first,
                        float quant, const float* pool_xyz, const int32_t* pool_off, int n_pool, const int32_t* ids, int B, int cap, int64_t bound,
                        int cloud_base, int64_t point_base, float* coords, float* feats, int32_t* pt_off, int32_t* n_dev, int32_t* err_dev,
                        int32_t* zero, int zero_words, void* stream);

/* The whole coordinate pyramid of MEConvExp.forward in three launches, for batches whose clouds have at most
 * dd3d_cloud_maps_max_points() points each: one workgroup per cloud builds the voxel hashes of tensor strides 1, 2, 4 in LDS
 * (duplicates, neighbours and pooling children never leave a cloud because the batch index is part of every key).
 * coords[.,4] f32 rows grouped by cloud, pt_off[B+1] = first point of every cloud (device), bound >= pt_off[B], ld >= bound.
 * Outputs (all device, sized by bound; bit-identical to dd3d_floor_coords/unique_rows/voxel_csr/stride_coords/kernel_map/
 * segment_offsets/dense_sites): counts[3] rows at strides 1,2,4; seg[3][B+1] per-cloud row ranges; C1,C2,C4 coordinates; inv1,
 * uidx1, vox_off, vox_pts; par2/par4 (row at the coarser stride) and uidx2/uidx4 (first finer row); nbr1/nbr2 [27][ld] k=3 kernel
 * maps at strides 1 and 2; kid2/kid4 [8][ld] k=2 stride-2 pooling maps; optional dense placement of stride 4 (gmin[G,3], site,
 * vox_at[E0*E1*E2][B]; pass site = NULL to skip); optional group_row_off[3][G+1] = first row of every group of clouds at the three
 * strides, from group_cloud_off[G+1] (nullable pair); optional gmin_floor[G,3]: an upper bound of every group's minimum corner -- the
 * corner of the clouds of the group that are NOT in this call (a rank that holds part of a class's real batch: dense() places by the
 * minimum over the whole batch, MEConv.py:251-254).  With group_cloud_off given, cloud_group[b] must be the group whose range
 * [group_cloud_off[g], group_cloud_off[g+1]) holds cloud b (the groups' clouds are contiguous): the count scan then runs inside the
 * second pass (two launches instead of three).  *err_dev |= 1 batch column of a point != its cloud, |= 2 voxel outside the
 * dense extent, |= 8 a cloud has more than max_points points.
 * replaces: x.sparse(), the coordinate maps of both MinkowskiConvolution and both MinkowskiMaxPooling layers and the index math of
 * SparseTensor.dense(), models/MEConv.py:247,224,227,250-254 */
int dd3d_cloud_maps_max_points(void);
int64_t dd3d_cloud_maps_ws_bytes(int B);
int dd3d_cloud_maps(const float* coords, const int32_t* pt_off, int B, int max_points, int64_t bound, const int32_t* cloud_group, int G, int E0,
                    int E1, int E2, int32_t* counts, int32_t* seg, int32_t* C1, int32_t* C2, int32_t* C4, int32_t* inv1, int32_t* uidx1,
                    int32_t* vox_off, int32_t* vox_pts, int32_t* par2, int32_t* uidx2, int32_t* par4, int32_t* uidx4, int32_t* nbr1, int32_t* nbr2,
                    int32_t* kid2, int32_t* kid4, int64_t ld, int32_t* gmin, int32_t* site, int32_t* vox_at, const int32_t* group_cloud_off,
                    int32_t* group_row_off, const int32_t* gmin_floor, int32_t* err_dev, void* ws, int64_t ws_bytes, void* stream);
/* dd3d_cloud_maps + dd3d_voxel_mean_fwd for 3-channel point features in the same launches: F0[bound][3] = mean of feats3[N][3] over the
 * points of every stride-1 voxel (summed in ascending point order: the bits of dd3d_voxel_mean_fwd).  Same arguments otherwise.
 * replaces: x.sparse() including its feature average, models/MEConv.py:247 */
int dd3d_cloud_maps_f0(const float* coords, const int32_t* pt_off, int B, int max_points, int64_t bound, const int32_t* cloud_group, int G, int E0,
                    int E1, int E2, int32_t* counts, int32_t* seg, int32_t* C1, int32_t* C2, int32_t* C4, int32_t* inv1, int32_t* uidx1,
                    int32_t* vox_off, int32_t* vox_pts, int32_t* par2, int32_t* uidx2, int32_t* par4, int32_t* uidx4, int32_t* nbr1, int32_t* nbr2,
                    int32_t* kid2, int32_t* kid4, int64_t ld, int32_t* gmin, int32_t* site, int32_t* vox_at, const int32_t* group_cloud_off,
                    int32_t* group_row_off, const int32_t* gmin_floor, int32_t* err_dev, void* ws, int64_t ws_bytes, void* stream, const float* feats3, float* F0);

/* ---------------------------------------------------------------------------------------------------------------------
 * (2) sparse convolution, gather-GEMM with atomic-free output (fp32 MFMA for Cin = 128)
 * ------------------------------------------------------------------------------------------------------------------- */

/* Row counts: M / M_out / M_in is the number of rows, or -- when the nullable `m_dev` is given -- a BOUND, with the true count
 * read from the device (rows beyond it are not touched).  Entry points that take group_row_off[G+1] / seg_off[B+1] device arrays
 * always follow those (their chunking depends on group_row_off[G] only), so a bound may be passed there as well.  This is what
 * lets a whole matching step be captured once into a HIP graph and replayed on batches of different sizes.
 *
 * Y[M_out,Cout] = sum_k X[nbr[k'][o]] . W[k],  W is [K,Cin,Cout], k' = mirror ? K-1-k : k.  K = 27, Cout = 128, Cin in {3,128}.
 * replaces: ME.MinkowskiConvolution forward (ConvolutionForwardGPU), models/MEConv.py:224 via :248 */
int dd3d_conv_fwd(const float* X, int64_t M_in, int Cin, const float* W, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                  const int32_t* m_dev, int mirror, float* Y, void* stream);

/* Launch schedule of the Cin = Cout = 128 convolution over one kernel map: perm[M] = the rows sorted (stable radix sort) by their K-bit
 * pattern of present offsets, so that the 32 rows one wave multiplies together share a pattern and absent offsets are skipped for the
 * whole block; tile_map[dd3d_conv_schedule_blocks(M)] = workgroup -> tile (dd3d_conv_schedule_tile_rows(M) rows) of that order (-1 = none), the tiles ranked by
 * cost and dealt to the CUs round by round, heaviest tile to the least loaded CU, so that the most loaded CU carries the average.
 * Optional outputs (both or neither): nbr_sorted[K][ld] = the kernel map with its columns in pattern order, tile_mask[ceil(M / tile rows)]
 * = the offsets present in each tile; with them a workgroup of dd3d_conv128_ordered reads its part of the table coalesced.
 * M is a bound when m_dev is given.  One schedule serves the forward and the mirrored launch.  No counterpart in MinkowskiEngine (its
 * gather-GEMM-scatter compacts per offset instead); serves MEConv.py:224 like dd3d_conv_fwd. */
int dd3d_conv_schedule_tile_rows(int64_t M); /* 128 from 65536 rows on (k_conv128_rows), 32 below (k_conv128<32,1,4>) */
int64_t dd3d_conv_schedule_blocks(int64_t M);
int64_t dd3d_conv_schedule_ws_bytes(int64_t M);
int dd3d_conv_schedule(const int32_t* nbr, int64_t ld, int K, int64_t M, const int32_t* m_dev, int32_t* perm, int32_t* tile_map,
                       int32_t* nbr_sorted, uint32_t* tile_mask, void* ws, int64_t ws_bytes, void* stream);

/* dd3d_conv_fwd for Cin = Cout = 128, K = 27 along a schedule: bit-identical output (a row's sum does not depend on the rows it shares
 * a block with), fewer all-zero rows through the matrix cores, even load over the CUs.
 * replaces: ConvolutionForwardGPU / ConvolutionBackwardGPU (input gradient, mirror = 1 with the transposed kernel), MEConv.py:224 */
int dd3d_conv128_ordered(const float* X, int64_t M_in, const float* W, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* m_dev,
                         int mirror, const int32_t* perm, const int32_t* tile_map, const int32_t* nbr_sorted, const uint32_t* tile_mask, float* Y,
                         void* stream);

/* dd3d_conv_fwd for Cin = Cout = 128, K = 27 for launches of few rows (the synthetic rows of the second-order pass): the offsets of
 * every 32-row tile are dealt over up to nine workgroups (count chosen from the true row count), partial tiles go to ws and are added
 * in split order.  Deterministic; equals dd3d_conv_fwd to fp32 rounding (the sum over offsets is associated per split), not bit for bit.
 * replaces: ConvolutionForwardGPU / ConvolutionBackwardGPU (input gradient, mirror = 1 with the transposed kernel), MEConv.py:224 */
int64_t dd3d_conv128_split_ws_bytes(int64_t M_out);
int dd3d_conv128_split(const float* X, int64_t M_in, const float* W, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* m_dev,
                       int mirror, float* Y, void* ws, int64_t ws_bytes, void* stream);

/* dX[M,Cin] of a stride-1 convolution.  Cin = 128: pass the transposed kernel [K,Cout,Cin]; Cin = 3: pass W itself.
 * replaces: ConvolutionBackwardGPU (input gradient) reached by torch.autograd.grad / loss.backward, mnist_distill.py:349,359,365 */
int dd3d_conv_bwd_data(const float* dY, int64_t M, const int32_t* m_dev, int Cout, const float* W_or_Wt, int Cin, const int32_t* nbr, int64_t ld,
                       int K, float* dX, void* stream);

/* dW[G,K,Cin,Cout], one kernel gradient per group of rows [group_row_off[g], group_row_off[g+1]) (device array of G+1 ints).
 * Deterministic: fixed row chunks -> partial slabs -> fixed-order reduction.  replaces: ConvolutionBackwardGPU (kernel gradient) */
int64_t dd3d_conv_bwd_weight_ws_bytes(int64_t M_out, int G, int K, int Cin, int Cout);
int dd3d_conv_bwd_weight(const float* X, int Cin, const float* dY, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                         const int32_t* group_row_off, int G, float* dW, void* ws, int64_t ws_bytes, void* stream);

/* dd3d_conv_bwd_weight for Cin = Cout = 128, K = 27 over a batch of SMALL clouds (every cloud at most
 * dd3d_conv128_bwd_weight_clouds_max_rows() = 256 rows), the cloud-window form: seg_off[B+1] = first row of every cloud (rows grouped by
 * cloud, clouds grouped by group, as dd3d_cloud_maps / dd3d_segment_offsets deliver them); a neighbour never leaves its cloud, so a
 * workgroup keeps a window of whole clouds of X and dY in LDS and every product reads its operands from there: each operand row comes
 * from HBM/L2 once per quarter-workgroup instead of once per offset, and the partial sums stay in registers over ~M/64 rows.  Same
 * result as dd3d_conv_bwd_weight to fp32 rounding (another, equally fixed, summation order); bitwise reproducible; M_out may be a bound
 * (the chunking follows group_row_off[G], read on the device).  *err_dev |= 16 if a cloud has more rows than the window holds (its
 * pairs are then missing from dW: route such batches to dd3d_conv_bwd_weight), |= 1 if a neighbour lies outside its row's cloud.
 * replaces: ConvolutionBackwardGPU (kernel gradient) reached by torch.autograd.grad, mnist_distill.py:349,359 via models/MEConv.py:224 */
int dd3d_conv128_bwd_weight_clouds_max_rows(void);
int64_t dd3d_conv128_bwd_weight_clouds_ws_bytes(int64_t M_out, int G);
int dd3d_conv128_bwd_weight_clouds(const float* X, const float* dY, const int32_t* nbr, int64_t ld, int K, int64_t M_out, const int32_t* seg_off,
                                   int B, const int32_t* group_row_off, int G, float* dW, int32_t* err_dev, void* ws, int64_t ws_bytes,
                                   void* stream);

/* Sparse convolution for ANY channel widths 1 <= Cin, Cout <= dd3d_conv_any_max_channels() (128) and any K >= 1: the general-purpose
 * siblings of the three entry points above, for the shapes that only the reference's unused model variants name (models/MEConv.py:25-39,
 * 102-130: 1 -> 128, 128 -> 64 -> 32 -> 16 at kernel size 3; 16 -> 1 and 128 -> 128 at kernel size 1) and for research variants of the network.
 * Exact-fp32 MFMA on zero-padded 32-wide channel blocks, deterministic.  W is [K,Cin,Cout].  nbr == NULL is allowed for K == 1 only and means
 * "every row is its own neighbour" (a kernel_size = 1 convolution).  M / M_out may be a bound when m_dev is given.
 *   fwd:        Y[o]   = sum_k X[nbr[k][o]] . W[k]
 *   bwd_data:   dX[i]  = sum_k dY[nbr[K-1-k][i]] . W[k]^T   (stride-1 convolution over one coordinate set, odd kernel: K odd)
 *   bwd_weight: dW[k]  = sum_o X[nbr[k][o]]^T dY[o]          (ONE group; chunk partials in ws, added in chunk order)
 * replaces: ME.MinkowskiConvolution forward / backward (ConvolutionForwardGPU / ConvolutionBackwardGPU) for models/MEConv.py:25,32,39,102,109,116,130 */
int dd3d_conv_any_max_channels(void);
int dd3d_conv_any_fwd(const float* X, int64_t M_in, int Cin, const float* W, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                      const int32_t* m_dev, float* Y, void* stream);
int dd3d_conv_any_bwd_data(const float* dY, int64_t M, const int32_t* m_dev, int Cout, const float* W, int Cin, const int32_t* nbr, int64_t ld,
                           int K, float* dX, void* stream);
int64_t dd3d_conv_any_bwd_weight_ws_bytes(int K, int Cin, int Cout);
int dd3d_conv_any_bwd_weight(const float* X, int Cin, const float* dY, int Cout, const int32_t* nbr, int64_t ld, int K, int64_t M_out,
                             const int32_t* m_dev, float* dW, void* ws, int64_t ws_bytes, void* stream);

/* Wt[k][co][ci] = W[k][ci][co] */
int dd3d_transpose_kernel(const float* W, int K, int Cin, int Cout, float* Wt, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * (3) voxel mean, instance norm, affine + ReLU + max-pool -- bandwidth kernels
 * ------------------------------------------------------------------------------------------------------------------- */

/* F0[v] = mean of the features of the points of voxel v (ascending point order).  replaces: MinkowskiSPMMAverageFunction inside
 * TensorField.sparse(), models/MEConv.py:247 -- backward is the only route by which gradient reaches the synthetic xyz */
int dd3d_voxel_mean_fwd(const float* feats, int64_t N, int C, const int32_t* vox_off, const int32_t* vox_pts, int64_t M, const int32_t* m_dev,
                        float* F0, void* stream);
int dd3d_voxel_mean_bwd(const float* dF0, int64_t M, int C, const int32_t* inv, const int32_t* vox_off, int64_t N, float* dfeats, void* stream);

/* xhat = (Y - mean_b) / sqrt(var_b + 1e-8) per cloud b and channel (biased variance); inv_std[B,C].
 * replaces: MinkowskiInstanceNormFunction forward/backward, models/MEConv.py:225 */
int dd3d_inorm_fwd(const float* Y, const int32_t* seg_off, int B, int C, float* xhat, float* inv_std, void* stream);
int dd3d_inorm_bwd(const float* g, const float* xhat, const float* inv_std, const int32_t* seg_off, int B, int C, float* dY, void* stream);

/* out[o] = max over existing children of relu(xhat*w + b) (use_affine_relu = 0: plain max); arg = winning child row.
 * replaces: the affine of ME.MinkowskiInstanceNorm, ME.MinkowskiReLU and ME.MinkowskiMaxPooling forward, models/MEConv.py:225-227 */
int dd3d_affine_relu_pool_fwd(const float* xhat, const float* w, const float* b, int use_affine_relu, int C, const int32_t* kids, int64_t ld,
                              int Kp, int64_t M_out, const int32_t* m_dev, float* out, int32_t* arg, void* stream);

/* dd3d_inorm_fwd + dd3d_affine_relu_pool_fwd (affine + ReLU on) in one launch, one workgroup per cloud: identical outputs; the pooled
 * level re-reads its children from cache instead of HBM.  seg_out[B+1] = the clouds' row ranges at the pooled level.
 * replaces: MinkowskiInstanceNorm + MinkowskiReLU + MinkowskiMaxPooling forward, models/MEConv.py:225-227 */
int dd3d_inorm_arp_fwd(const float* Y, const int32_t* seg_off, int B, int C, float* xhat, float* inv_std, const float* w, const float* b,
                       const int32_t* kids, int64_t ld, int Kp, const int32_t* seg_out, float* pooled, int32_t* arg, void* stream);
/* The same launch WITHOUT the normalised rows in memory: it writes mean[B,C] next to inv_std[B,C] instead of xhat[M,C] (one full
 * [rows, C] write less per level); the pooled rows normalise their children as they gather them from Y, and dd3d_arp_inorm_bwd_y
 * does the same in the backward passes.  inv_std / pooled / arg: the same bits as dd3d_inorm_arp_fwd.  Y must stay alive for the backward.
 * replaces: MinkowskiInstanceNorm + MinkowskiReLU + MinkowskiMaxPooling forward, models/MEConv.py:225-227 (what the step runs) */
int dd3d_inorm_arp_fwd_y(const float* Y, const int32_t* seg_off, int B, int C, float* mean, float* inv_std, const float* w, const float* b,
                         const int32_t* kids, int64_t ld, int Kp, const int32_t* seg_out, float* pooled, int32_t* arg, void* stream);

/* MinkowskiAvgPooling(kernel_size=2, stride=2) of the reference's unused model variants (models/MEConv.py:28,35,105,112,126):
 * out[v][C] = mean of the existing children of v (kids[Kp<=8][ld], -1 = none; the divisor is the number of children that exist),
 * cnt[v] = that number (kept for the backward).  C % 4 == 0.  M_out may be a bound when m_dev (device int) is given.
 * Backward: every input row has exactly one parent, dX[i] = g[parent[i]] / cnt[parent[i]]. */
int dd3d_avg_pool_fwd(const float* X, int64_t M_in, int C, const int32_t* kids, int64_t ld, int Kp, int64_t M_out, const int32_t* m_dev, float* out,
                      int32_t* cnt, void* stream);
int dd3d_avg_pool_bwd(const float* g, const int32_t* cnt, const int32_t* parent, int64_t M_in, int64_t M_out, int C, float* dX, void* stream);
/* MinkowskiGlobalAvgPooling (models/MEConv.py:20,97): out[b][C] = mean over the rows [seg_off[b], seg_off[b+1]) of cloud b (zeros for an
 * empty cloud); backward dX[i] = g[b] / rows(b). */
int dd3d_global_avg_pool_fwd(const float* X, int64_t M, const int32_t* seg_off, int B, int C, float* out, void* stream);
int dd3d_global_avg_pool_bwd(const float* g, int64_t M, const int32_t* seg_off, int B, int C, float* dX, void* stream);

/* backward of the same three ops for every input row; gw_w/gw_b[G,C] are the per-group affine gradients (nullable pair);
 * ga_out (nullable) keeps the ReLU-masked pooled gradient; ga_prev + gw_extra[G,C] (nullable pair) add ga_prev*gw_extra[group]. */
int64_t dd3d_arp_bwd_ws_bytes(int64_t M_in, int G);
int dd3d_affine_relu_pool_bwd(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* w, const float* b,
                              int use_affine_relu, int C, int64_t M_in, const int32_t* group_row_off, int G, float* g_xhat, float* ga_out,
                              const float* ga_prev, const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream);
/* affine_relu_pool_bwd followed by inorm_bwd fused into one pass per cloud (the gradient w.r.t. xhat never goes through HBM):
 * dY[M_in,128] is the gradient w.r.t. the instance norm's INPUT.  cloud_group[B] nullable (one group); group_cloud_off[G+1]
 * needed when gw_w/gw_b are requested.  Same nullable pairs as dd3d_affine_relu_pool_bwd.  use_affine_relu is implied (w, b given).
 * replaces: backward of ME.MinkowskiInstanceNorm + MinkowskiReLU + MinkowskiMaxPooling, models/MEConv.py:225-227 */
int64_t dd3d_arp_inorm_bwd_ws_bytes(int B);
int dd3d_arp_inorm_bwd(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* inv_std, const float* w,
                       const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group, const int32_t* group_cloud_off, int G,
                       float* dY, float* ga_out, int ga_clouds /* ga_out is written for clouds < ga_clouds only; <= 0: all */, const float* ga_prev, const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes,
                       void* stream);
/* dd3d_arp_inorm_bwd over the norm's INPUT rows Y and the per-cloud mean / inv_std of dd3d_inorm_arp_fwd_y (a row is normalised as it
 * is read: the two roundings that a stored xhat row went through); every output has the same bits as dd3d_arp_inorm_bwd's.
 * replaces: backward of ME.MinkowskiInstanceNorm + MinkowskiReLU + MinkowskiMaxPooling, models/MEConv.py:225-227 (what the step runs) */
int dd3d_arp_inorm_bwd_y(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* Y, const float* mean, const float* inv_std,
                         const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                         const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                         const float* gw_extra, float* gw_w, float* gw_b, void* ws, int64_t ws_bytes, void* stream);
/* dd3d_arp_inorm_bwd without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot allocate -- code under HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc.  The per-cloud
 * partial sums (2 x 128 floats per cloud) live in a file-scope device array of the library, 8 MiB per GPU, held until the process
 * ends; its device address is taken once per device with hipGetSymbolAddress.  B > dd3d_arp_inorm_bwd_static_max_clouds() (8192)
 * returns -2 (use dd3d_arp_inorm_bwd with a caller workspace there).  Same kernels, same outputs bit for bit.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two threads) race -- serialise them.  Nothing needs clearing
 * between calls: every slot a call reads, it has written.
 * Set-up: the FIRST call on a device takes the array's address (hipGetDevice + hipGetSymbolAddress, host-side queries).  A caller that
 * will make its calls under graph capture makes one call OUTSIDE capture first, with that device current -- B = 0 (w, b non-null,
 * everything else may be null) launches nothing and returns 0 with the address taken (tests/test_zz_no_alloc.py captures the call
 * after exactly this set-up, in a fresh process). */
int dd3d_arp_inorm_bwd_static_max_clouds(void);
int dd3d_arp_inorm_bwd_static(const float* g_pool, const int32_t* arg, const int32_t* parent, const float* xhat, const float* inv_std,
                              const float* w, const float* b, int C, const int32_t* seg_off, int B, const int32_t* cloud_group,
                              const int32_t* group_cloud_off, int G, float* dY, float* ga_out, int ga_clouds, const float* ga_prev,
                              const float* gw_extra, float* gw_w, float* gw_b, void* stream);
int dd3d_scale_rows_by_group(const float* ga, const float* gw, const int32_t* group_row_off, int G, int64_t M, int C, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * (3b) classifier over the dense placement without the dense tensor
 * replaces: SparseTensor.dense() + view(-1,100352) + nn.Linear(100352,10) and their autograd, models/MEConv.py:254-256
 * strides[4] (HOST) = element strides of the weight-like operand over (group, output j, channel c, site s)
 * ------------------------------------------------------------------------------------------------------------------- */
/* dd3d_sparse_linear_fwd, dd3d_softmax_ce and dd3d_group_sum_rows(dlogits) in two launches instead of five, same bits: logits[B,J],
 * prob[B,J] (nullable), dlogits[B,J], cloud_loss[B] (scratch), loss_g[G] (nullable), bias_grad[G,J] (nullable) = the classifier-bias
 * gradient of every group.  counters: G ints, zero before the first call (left zero by every call); max_per_group >= clouds of the
 * largest group.
 * replaces: self.classifier(dense) + nn.CrossEntropyLoss + its backward into the logits and the bias, models/MEConv.py:255,
 * mnist_distill.py:346-349,355-358 */
int dd3d_classifier_head(const float* F, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                         const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, const float* bias, const int32_t* labels,
                         const int32_t* group_cloud_off, int G, int max_per_group, float* logits, float* prob, float* dlogits, float* cloud_loss,
                         float* loss_g, float* bias_grad, int32_t* counters, void* ws, int64_t ws_bytes, void* stream);

/* coords = int32 [M,4] voxel rows (column 0 = cloud index), M_bound = their bound; the true row count is seg_off[B] (device).
 * fwd runs as per-voxel partial products (one wave per voxel) + an ordered per-cloud sum: ws from dd3d_sparse_linear_ws_bytes. */
int64_t dd3d_sparse_linear_ws_bytes(int64_t M_bound, int J);
int dd3d_sparse_linear_fwd(const float* F, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                           const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, const float* bias, float* out, void* ws,
                           int64_t ws_bytes, void* stream);
int dd3d_sparse_linear_bwd_x(const float* dl, const int32_t* site, const int32_t* coords, int64_t M_bound, const int32_t* seg_off, int B, int C,
                             const int32_t* cloud_group, const float* Wl, const int64_t* strides, int J, int accumulate, float* dF, void* stream);
int dd3d_sparse_linear_bwd_w(const float* dl, const float* F, const int32_t* vox_at, int S, int B, const int32_t* group_cloud_off, int G, int C, int J,
                             float* dWl, const int64_t* strides, int64_t dWl_elems, void* stream);

/* site-major copy of a [G][J, C*S] operand: out[g][s][j][c] = in[g][j][c*S+s] (inverse = 0), or back (inverse = 1).  The site-major
 * form makes every voxel's C weights of one output contiguous (strides {J*C*S, C, 1, J*C}). */
int dd3d_classifier_layout(const float* in, int G, int J, int C, int S, int inverse, float* out, void* stream);

/* mean softmax cross-entropy per group of clouds and its first / second-order backward.
 * replaces: nn.CrossEntropyLoss + autograd, mnist_distill.py:302,344,355 */
int dd3d_softmax_ce(const float* logits, const int32_t* labels, const int32_t* cloud_group, const int32_t* group_cloud_off, int G, int B, int J,
                    float* prob, float* dlogits, float* cloud_loss, float* loss_g, void* stream);
int dd3d_group_sum_rows(const float* v, const int32_t* group_cloud_off, int G, int J, float* out, void* stream);

/* per_class[G] = a + b + c + d, added left to right (b, c, d nullable; the partial matching losses of a step, which are computed on
 * different streams) and *total = the sum over the groups, in one launch; per_class / total nullable.  Deterministic.
 * replaces: the running `loss +=` over the classes, mnist_distill.py:361 */
int dd3d_loss_total(const float* a, const float* b, const float* c, const float* d, int G, float* per_class, float* total, void* stream);
int dd3d_softmax_ce_bwd2(const float* prob, const float* u, const float* u_group /* [G,J] added to every cloud's u, nullable */,
                         const int32_t* group_cloud_off, int G, int J, int max_clouds_per_group, float* dz, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * (4) gradient distance -- one segmented reduction over all layers, and its backward
 * replaces: distance_wb / distance_mse / match_loss, distillation_loss.py:26-130
 * ------------------------------------------------------------------------------------------------------------------- */
#define DD3D_MAX_SEG 16
#define DD3D_METRIC_OURS 0
#define DD3D_METRIC_MODMSE 1
#define DD3D_METRIC_MSE 2
#define DD3D_METRIC_COS 3

typedef struct dd3d_gseg {
    const float* syn;  /* gradient of the synthetic batch, one layer */
    const float* real; /* gradient of the real batch, same shape */
    float* dsyn;       /* out (backward): d loss / d syn, or NULL */
    int64_t rows;      /* logical rows after the reference's reshape (distillation_loss.py:27-47) */
    int64_t cols;      /* logical columns */
    int32_t colmajor;  /* 0: rows contiguous ([rows,cols] row-major).  1: element (r,q) at q*rows + r, i.e. a MinkowskiEngine kernel
                          [K,Cin,Cout] viewed as permute(2,1,0) rows = Cout (distillation_loss.py:35-40) */
    int32_t pad_;
} dd3d_gseg;

/* segs is a HOST array of n <= DD3D_MAX_SEG descriptors.  G >= 1 independent groups (classes): every segment tensor is
 * [G, rows*cols] contiguous and loss_dev has G entries (one matching loss per class).  stats_dev needs
 * dd3d_match_loss_stats_floats() floats and is kept for the backward. */
int64_t dd3d_match_loss_ws_bytes(const dd3d_gseg* segs, int n, int G);
int64_t dd3d_match_loss_stats_floats(const dd3d_gseg* segs, int n, int G);
int dd3d_match_loss_fwd(const dd3d_gseg* segs, int n, int G, int metric, float* loss_dev, float* stats_dev, void* ws, int64_t ws_bytes,
                        void* stream);
int dd3d_match_loss_bwd(const dd3d_gseg* segs, int n, int G, int metric, const float* stats_dev, const float* gl_dev, void* ws, int64_t ws_bytes,
                        void* stream);
/* dd3d_match_loss_fwd followed by dd3d_match_loss_bwd, same outputs bit for bit: ONE launch when the list is small (at most 8 block-wide
 * passes in all: a row-major row of at most 2048 columns is one, a column-major segment one per 128 rows x 64 columns; at most 256 rows --
 * e.g. an instance-norm weight's gradient, [1,128] per class, or the first layer's kernel gradient) and there are at least 8 groups (one
 * workgroup per group), the three launches otherwise (longer lists and fewer groups are faster as wide launches: measured).  dd3d_match_loss_fwd takes the same one-launch form for such lists.  north_star (4): "a single segmented-reduction kernel" where a single kernel is the faster form.
 * replaces: match_loss + the backward through it, mnist_distill.py:360,365 */
int dd3d_match_loss_fwd_bwd(const dd3d_gseg* segs, int n, int G, int metric, float* loss_dev, float* stats_dev, const float* gl_dev, void* ws,
                            int64_t ws_bytes, void* stream);

/* x -= lr * (mom = first_step ? grad : momentum*mom + grad); optional clamp to [lo,hi] with column 0 of every xyz row forced to 0.
 * replaces: optimizer_img.step() and the next iteration's clamp, mnist_distill.py:369,311-317 */
int dd3d_sgd_momentum_clamp(float* x, const float* grad, float* mom, int64_t n, float lr, float momentum, int first_step, int do_clamp, float lo,
                            float hi, void* stream);

/* Prologue of an iteration in one launch: clamp all synthetic points in place (do_clamp: [lo,hi]; zero_first: first axis := 0) and write
 * the clouds this process owns (dest_cloud[c] = position among the owned clouds or -1; NULL = all, in order) into the head of the
 * merged batch: coords[r][1..3] = xyz / quant ([.,4] rows, column 0 untouched), feats[r] = xyz.
 * replaces: the clamp mnist_distill.py:311-317, get_syn_tensor :241-244, Mnist2Dsyn.__getitem__ Mnist2D.py:84-92, create_input_batch MEConv.py:260-267 */
int dd3d_syn_head(float* xyz, int64_t n_clouds, int npts, const int32_t* dest_cloud, int do_clamp, float lo, float hi, int zero_first,
                  float quant, float* coords, float* feats, void* stream);

/* Epilogue in one launch: grad_out (nullable, full size) = the owned clouds' gradient dfeats scattered back, zero elsewhere;
 * mom = momentum*mom + grad; xyz -= lr*mom.  replaces: image_syn.grad assembly + optimizer_img.step(), mnist_distill.py:365,369 */
int dd3d_sgd_scatter(float* xyz, float* mom, const float* dfeats, const int32_t* dest_cloud, int64_t n_clouds, int npts, float lr,
                     float momentum, float* grad_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * (6) evaluation block: synthetic clouds -> occupancy images
 * ------------------------------------------------------------------------------------------------------------------- */

/* img[cloud, 0, trunc(xyz[1]), trunc(xyz[2])] = 1 for every point; img is [n_clouds,1,H,W] fp32 and is NOT cleared here (the caller
 * passes zeros, as the reference does).  Negative indices count from the end (Python indexing); an index outside the image sets
 * bit 0 of *err (the reference raises IndexError).  replaces: populate_img, mnist_distill.py:163-168 */
int dd3d_populate_img(const float* xyz, int64_t n_clouds, int64_t pts_per_cloud, int H, int W, float* img, int32_t* err, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DD3D_H */
