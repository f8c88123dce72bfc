/*
 * mocopci_hip.h -- C ABI of libmocopci_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the point-set hot path of icdm-adteam/MoCoPCI.
 * Part 1 replaces, one for one, the nine functions the reference registers in its
 * pybind11 module `pointnet2_cuda` (pointnet2/src/pointnet2_api.cpp:10-24); the
 * integer argument order of every entry point is the reference wrapper's.
 * Part 2 is the Python-level second boundary (knn_point, index_points_group,
 * UpsampleFlow, ... in models/pointconv_util.py and models/m_models/mocopci.py)
 * expressed as fused kernels.
 *
 * Conventions (same contract as the reference extension):
 *   - all pointers are DEVICE pointers into contiguous fp32 / int32 buffers owned by
 *     the caller; the library never allocates or frees, reads no environment variable, and keeps no
 *     state between calls except two per-process conveniences: the per-device "kernel attribute set"
 *     flags (marked after the attribute call) and the opt-in launch timers of mcp_prof_*;
 *   - every call is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the null stream) and returns without synchronising;
 *   - re-entrant and thread-safe; the device is the calling thread's current device;
 *   - return value: 0 on success, otherwise a hipError_t value (launch failure) or
 *     MCP_ERR_* (argument validation).  The reference calls exit(-1) on launch
 *     failure (pointnet2/src/sampling_gpu.cu:248-252); this library never exits.
 */
#ifndef MOCOPCI_HIP_H
#define MOCOPCI_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mcp_stream_t; /* hipStream_t */

#define MCP_OK 0
#define MCP_ERR_BAD_ARG 10001     /* null pointer, non-positive dimension */
#define MCP_ERR_UNSUPPORTED 10002 /* a size outside what the kernels are built for (e.g. K > 32) */

#define MCP_ABI_VERSION 1

int mcp_abi_version(void);
/* Static string for a code returned by any mcp_* call. */
const char *mcp_error_string(int code);

/* ---------------- Part 1: pointnet2_cuda replacements ------------------------------ */

/* furthest_point_sampling_wrapper(b,n,m,points,temp,idx)   pointnet2/src/sampling.cpp:38-49,
 * kernel sampling_gpu.cu:93-253.  xyz (B,N,3); temp (B,N) scratch pre-filled with 1e10 by the
 * caller (pointnet2_utils.py:26), holds the final min-distances on return; idx (B,M) int32.
 * Bit-exact with the reference kernel's index sequence including its tie rule.
 * No memory beyond the arguments is used.  For 16384 < N <= 65536 the fast (tiled, spatially pruned) kernel needs scratch
 * for a sorted copy of the cloud: pass it through mcp_furthest_point_sampling_ws; this entry point, which keeps the reference
 * wrapper's exact argument list, then runs the plain streaming kernel (same indices, ~9x slower at 8 x 65536 -> 2048). */
int mcp_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx, mcp_stream_t stream);

/* Same operation with caller-provided scratch (any workspace is a caller pointer + a size query).
 * mcp_fps_workspace_bytes returns how many bytes this (b, n, m) can use -- 0 when the size needs none -- and
 * mcp_furthest_point_sampling_ws takes a device buffer of at least that size (NULL / too small: behaves like
 * mcp_furthest_point_sampling).  The buffer is only used during the call (stream-ordered). */
size_t mcp_fps_workspace_bytes(int b, int n, int m);
int mcp_furthest_point_sampling_ws(int b, int n, int m, const float *xyz, float *temp, int *idx, void *workspace,
                                   size_t workspace_bytes, mcp_stream_t stream);

/* The same sampling for a fresh start (temp = 1e10 everywhere, which is what every caller of the reference passes:
 * pointnet2_utils.py:24): the running distances live and die in the kernel, so no (b,n) buffer is filled or allocated.
 * sampled_xyz (b,m,3), optional (NULL to skip): the coordinates of the selected points -- the index_points_gather the callers
 * run next (mocopci.py:1378-1379) -- written by the same launch.
 * MCP_ERR_UNSUPPORTED where only the streaming kernel applies (n > 65536, or 16384 < n <= 65536 without workspace). */
int mcp_furthest_point_sampling_fresh(int b, int n, int m, const float *xyz, int *idx, float *sampled_xyz, void *workspace,
                                      size_t workspace_bytes, mcp_stream_t stream);

/* gather_points_wrapper(b,c,n,npoints,points,idx,out)      sampling.cpp:11-22, sampling_gpu.cu:8-44
 * points (B,C,N), idx (B,npoints) -> out (B,C,npoints). */
int mcp_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out, mcp_stream_t stream);

/* gather_points_grad_wrapper(b,c,n,npoints,grad_out,idx,grad_points)  sampling.cpp:25-35,
 * sampling_gpu.cu:46-83.  grad_points (B,C,N) must be zeroed by the caller (pointnet2_utils.py:67). */
int mcp_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx, float *grad_points,
                           mcp_stream_t stream);

/* group_points_wrapper(b,c,n,npoints,nsample,points,idx,out)  group_points.cpp:26-37,
 * group_points_gpu.cu:47-86.  points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample). */
int mcp_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                     mcp_stream_t stream);

/* group_points_grad_wrapper(b,c,n,npoints,nsample,grad_out,idx,grad_points)  group_points.cpp:11-23,
 * group_points_gpu.cu:8-44.  grad_points pre-zeroed by the caller (pointnet2_utils.py:190). */
int mcp_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                          float *grad_points, mcp_stream_t stream);

/* Deterministic forms of the three scatter-add backwards (the reference's kernels -- and the three *_grad entry
 * points above and below, which keep its exact argument lists -- use atomicAdd: sampling_gpu.cu:46-83, group_points_gpu.cu:8-44,
 * interpolate_gpu.cu:120-161).  The caller passes, per batch element, the stable sort of the T scatter positions by destination:
 * order (B,T) int32 = positions in sorted order, seg (B,N+1) int32 = CSR offsets of each destination in it.  Each destination's
 * addends are summed in ascending position order (a sequential loop's order): the same bits on every run; grad_points needs no
 * zero-fill.
 *   mcp_group_points_grad_sorted: grad_out (B,C,T) -> grad_points (B,C,N).  K6 with T = npoints*nsample, K3 with T = npoints.
 *   mcp_three_interpolate_grad_sorted: grad_out (B,C,n), weight (B,n,3), positions t = 3*p + k of idx (B,n,3) sorted by idx value
 *     (order (B,3n), seg (B,m+1)) -> grad_points (B,C,m). */
int mcp_group_points_grad_sorted(int b, int c, int n, int t, const float *grad_out, const int *order, const int *seg,
                                 float *grad_points, mcp_stream_t stream);
int mcp_three_interpolate_grad_sorted(int b, int c, int n, int m, const float *grad_out, const int *order, const int *seg,
                                      const float *weight, float *grad_points, mcp_stream_t stream);

/* ball_query_wrapper(b,n,m,radius,nsample,new_xyz,xyz,idx)   ball_query.cpp:16-28, ball_query_gpu.cu:9-67
 * new_xyz (B,M,3) centres, xyz (B,N,3) -> idx (B,M,nsample), pre-zeroed by the caller
 * (pointnet2_utils.py:218): first nsample hits in index order, padded with the first hit. */
int mcp_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                   mcp_stream_t stream);

/* QueryAndGroup.forward as one launch                         pointnet2/pointnet2_utils.py:231-264
 * (= ball_query_wrapper + a transposed copy of xyz + 2 x group_points_wrapper + subtraction + cat in the reference).
 * xyz (B,N,3), new_xyz (B,M,3) centres, features (B,C,N) or NULL -> out (B, CT, M, nsample), CT = (use_xyz or no features ? 3 : 0) +
 * (features ? C : 0): channels 0..2 = neighbour coordinates relative to the centre, then the grouped feature channels; neighbours =
 * the first nsample points within `radius` in index order, padded with the first hit; a centre with no hit groups point 0
 * (the reference's pre-zeroed idx).  nsample <= 64 (MCP_ERR_UNSUPPORTED above); features NULL needs use_xyz. */
int mcp_query_and_group(int b, int n, int m, int c, float radius, int nsample, int use_xyz, const float *xyz, const float *new_xyz,
                        const float *features, float *out, mcp_stream_t stream);

/* three_nn_wrapper(b,n,m,unknown,known,dist2,idx)            interpolate.cpp:14-24, interpolate_gpu.cu:9-74
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) SQUARED distances, idx (B,n,3). */
int mcp_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, mcp_stream_t stream);

/* three_interpolate_wrapper(b,c,m,n,points,idx,weight,out)   interpolate.cpp:27-39, interpolate_gpu.cu:77-117
 * points (B,C,M), idx/weight (B,n,3) -> out (B,C,n). */
int mcp_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out,
                          mcp_stream_t stream);

/* three_interpolate_grad_wrapper(b,c,n,m,grad_out,idx,weight,grad_points)  interpolate.cpp:42-56,
 * interpolate_gpu.cu:120-161.  grad_points (B,C,M) pre-zeroed by the caller (pointnet2_utils.py:146). */
int mcp_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight,
                               float *grad_points, mcp_stream_t stream);

/* ---------------- Part 2: fused layer operators ------------------------------------ */

/* Distance forms for mcp_knn. */
#define MCP_DIST_EXPANSION 0 /* -2 q.r + |q|^2 + |r|^2 : square_distance, mocopci.py:1130-1155 */
#define MCP_DIST_DIRECT 1    /* sum (q-r)^2 : pytorch3d.ops.knn_points (pointconv_util.py:910) */

/* knn_point(nsample, xyz, new_xyz) (mocopci.py:1158-1169 = pointconv_util.py:129-140) and
 * pytorch3d.ops.knn_points(p1,p2,K) fused: never materialises the (B,Q,N) distance matrix.
 * query (B,Q,3), ref (B,N,3) -> idx (B,Q,K) int32 and, if dist != NULL, dist (B,Q,K):
 * the K smallest under the lexicographic order (distance, index), ascending.  1 <= K <= 32.
 * If N < K the tail repeats the last valid entry. */
int mcp_knn(int b, int q, int n, int k, int dist_form, const float *query, const float *ref, int *idx, float *dist,
            mcp_stream_t stream);

/* Spatially pruned variant of mcp_knn for large clouds: identical results (same definition, same fp32
 * canon), but queries and references are given in Morton order and each tile of consecutive sorted references
 * has a bounding box, so a wave visits tiles by ascending lower bound and stops early.
 *   mcp_morton_codes: xyz (B,N,3), box (B,6) = per-batch (min xyz, max xyz) -> codes (B,N) int32 (30 bit);
 *                     the caller sorts by code (any stable or unstable sort) to get a permutation.
 *   mcp_tile_boxes:   sorted_xyz (B,N,3) -> boxes (B,ceil(N/tile),6), tile = mcp_knn_tile_size().
 *   mcp_knn_pruned:   query_sorted (B,Q,3) with qperm (B,Q) = original row of each sorted query (NULL = identity),
 *                     ref_sorted (B,N,3) with rperm (B,N) = original index of each sorted reference, boxes as above
 *                     -> idx (B,Q,K) ORIGINAL reference indices at ORIGINAL query rows (+ dist).  1 <= K <= 32, N <= 65536.
 *   mcp_build_cloud:  all of the above in one launch for N <= 16384 (bbox, isotropic Morton keys, block radix sort, gather,
 *                     tile boxes): xyz (B,N,3) -> sorted_xyz (B,N,3), perm (B,N) int32, boxes (B,ceil(N/tile),6). */
int mcp_knn_tile_size(void); /* references per box tile (64): boxes arrays have ceil(N / tile) rows */
int mcp_build_cloud(int b, int n, const float *xyz, float *sorted_xyz, int *perm, float *boxes, mcp_stream_t stream);
int mcp_morton_codes(int b, int n, const float *xyz, const float *box, int *codes, mcp_stream_t stream);
int mcp_tile_boxes(int b, int n, const float *sorted_xyz, float *boxes, mcp_stream_t stream);
int mcp_knn_pruned(int b, int q, int n, int k, int dist_form, const float *query_sorted, const int *qperm,
                   const float *ref_sorted, const int *rperm, const float *boxes, int *idx, float *dist, mcp_stream_t stream);

/* knn_point_cosine(nsample, xyz, new_xyz) (pointconv_util.py:111-153) on channel-last features:
 * qfeat (B,Q,C), rfeat (B,N,C) -> idx (B,Q,K) (and dist if non-NULL): the K smallest of
 * d = 1 - <q^,r^>, x^ = x / sqrt(sum x^2 + 1e-8), under the order (d, index), ascending.
 * C in {64,128,256}, 1 <= K <= 16.  workspace: B*(Q+N)*C floats (normalised copies), caller-owned. */
int mcp_knn_cosine(int b, int q, int n, int c, int k, const float *qfeat, const float *rfeat, int *idx, float *dist,
                   float *workspace, mcp_stream_t stream);

/* index_points_group / index_points_gather (mocopci.py:1190-1215) without the permute copies:
 * points (B,N,C) channel-last, idx (B,T) -> out (B,T,C) (T = S*K or S), whole C*4-byte rows. */
int mcp_group_rows(int b, int n, int c, int t, const float *points, const int *idx, float *out, mcp_stream_t stream);
/* grouping + centre offset + LeakyReLU, the first layer of cross() when it is not fused into mcp_cross_volume
 * (pointconv_util.py:762-770): points (B,N,C), idx (B,S,K), centre (B,S,C) -> out (B,S,K,C) = leaky(points[idx] + centre).
 * C % 4 == 0, 16-byte aligned pointers. */
int mcp_group_rows_add_leaky(int b, int n, int c, int s, int k, float slope, const float *points, const int *idx,
                             const float *centre, float *out, mcp_stream_t stream);
/* its backward (channel-last counterpart of group_points_grad, group_points_gpu.cu:49-75): grad_out (B,T,C), idx (B,T)
 * -> grad_points (B,N,C) += scatter; the caller zero-initialises grad_points. */
int mcp_group_rows_grad(int b, int n, int c, int t, const float *grad_out, const int *idx, float *grad_points, mcp_stream_t stream);

/* The same gradient as a DETERMINISTIC segmented reduction: the caller sorts the T gather positions of each batch
 * element by destination row with a stable sort and passes the permutation order (B,T) int32 and the CSR offsets seg (B,N+1)
 * int32 (seg[b][d] .. seg[b][d+1] = the slice of order[b] that gathers row d).  grad_points (B,N,C) is fully written (rows
 * nobody gathered are zero), summed in the sorted order: bit-identical from run to run, no atomics. */
int mcp_group_rows_grad_sorted(int b, int n, int c, int t, const float *grad_out, const int *order, const int *seg,
                               float *grad_points, mcp_stream_t stream);

/* The operands of the *_grad_sorted entry points from a gather list: idx (B,T) int32 with values in [0,n) -> order (B,T) = the gather
 * positions 0..T-1 of each batch element sorted by (destination, position) -- a stable sort's result, whatever order the kernel's
 * atomics are served in -- and seg (B,n+1) = the CSR offsets.  A counting sort (count / scan / fill / per-row rank: five launches)
 * in place of a general key-value sort; positions whose value is outside [0,n) are left out (seg[b][n] = the number kept).
 * workspace: mcp_scatter_segments_workspace_bytes(b,t,n) caller-owned bytes. */
size_t mcp_scatter_segments_workspace_bytes(int b, int t, int n);
int mcp_scatter_segments(int b, int t, int n, const int *idx, int *order, int *seg, void *workspace, size_t workspace_bytes,
                         mcp_stream_t stream);

/* UpsampleFlow.forward (mocopci.py:1485-1502) / the interpolation half of PointWarping (:1472-1479):
 * dense (B,N,3), sparse (B,S,3), feat (B,S,C) channel-last -> out (B,N,C);
 * 3-NN in expansion form, weights 1/max(||d||,1e-10) normalised.  idx3 (int32) and w3 (B,N,3) are
 * caller-provided outputs (the library holds no workspace); callers can reuse them for further
 * feature tensors on the same (dense, sparse) pair via mcp_interp3_apply.  mcp_interp3_weights is the middle step on
 * its own: idx3 (B,N,3) from any exact 3-NN search (mcp_knn or mcp_knn_pruned, expansion form) -> w3 (B,N,3). */
int mcp_interp3(int b, int n, int s, int c, const float *dense, const float *sparse, const float *feat, float *out, int *idx3,
                float *w3, mcp_stream_t stream);
int mcp_interp3_weights(int b, int n, int s, const float *dense, const float *sparse, const int *idx3, float *w3,
                        mcp_stream_t stream);
int mcp_interp3_apply(int b, int n, int s, int c, const float *feat, const int *idx3, const float *w3, float *out,
                      mcp_stream_t stream);
/* backward of mcp_interp3_apply w.r.t. feat (channel-last counterpart of three_interpolate_grad, interpolate_gpu.cu:126-150):
 * grad_out (B,N,C) -> grad_feat (B,S,C) += w3 * grad_out at idx3; the caller zero-initialises grad_feat. */
int mcp_interp3_apply_grad(int b, int n, int s, int c, const float *grad_out, const int *idx3, const float *w3, float *grad_feat,
                           mcp_stream_t stream);

/* Deterministic backward of mcp_interp3_apply (autograd over the blend of UpsampleFlow / PointWarping, mocopci.py:1480-1481,
 * :1500-1501; the reference's K9, interpolate_gpu.cu:120-161, adds with atomicAdd): grad_w3 (B,N,3) = <grad_out[b,p,:],
 * feat[b, idx3[b,p,j], :]> and grad_feat (B,S,C) = the w3-weighted scatter of grad_out, every destination row's addends in ascending
 * position 3p + j -- (order (B,3N), seg (B,S+1)) = mcp_scatter_segments of idx3 viewed as (B, 3N) with n = S.  Either output may be
 * NULL (then feat / order / seg are not read).  Sums in a fixed order: bit-reproducible; no (B,N,3,C) intermediate. */
int mcp_interp3_apply_grad_sorted(int b, int n, int s, int c, const float *feat, const int *idx3, const float *w3, const float *grad_out,
                                  const int *order, const int *seg, float *grad_feat, float *grad_w3, mcp_stream_t stream);

/* MultiFrameEstimatier.knn_group + fusion (mocopci.py:798-819) after the two neighbour searches:
 * p1 (B,N,3) centres, p2 (B,N,3) gathered set, idx (B,N,64) int32 into p2 (32 self-neighbours of p1
 * followed by 32 neighbours of p1 in p2); per neighbour [d, |d|] -> 4->64->64->128 (1x1 conv + eval
 * BatchNorm folded into w,b by the caller + ReLU) -> channel max -> softmax over the 64 neighbours ->
 * weighted sum of neighbour coordinates -> out (B,N,3).  w1 (64,4), w2 (64,64), w3 (128,64) row-major.
 * nb must be 64 (the reference's fixed k = 32 + 32).  idx2 == NULL: idx is the (B,N,64) list; otherwise idx and idx2 are its
 * two halves as separate (B,N,32) lists, exactly as the two searches produce them (no concatenation pass). */
int mcp_fusion(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
               const float *w2, const float *b2, const float *w3, const float *b3, float *out, mcp_stream_t stream);

/* Backward of mcp_fusion (the reference differentiates mocopci.py:803-819 with autograd over the materialised (B,128,N,64)
 * activations; its only hand-written backward pieces are the atomicAdd scatters of group_points_gpu.cu:8-44).  Same arguments as
 * mcp_fusion, plus grad_out (B,N,3) = dL/dout.  Writes
 *   grad_p1 (B,N,3)       dL/dp1;
 *   grad_nb (B,N,64,3)    dL/d(p2[idx]) per gathered neighbour, in the order of the neighbour list(s) -- the caller scatters it into
 *                         dL/dp2 with mcp_group_rows_grad_sorted (deterministic) or any scatter-add of its own;
 *   grad_weights          mcp_fusion_grad_floats() = 12800 floats: dW1 (64,4) | db1 (64) | dW2 (64,64) | db2 (64) | dW3 (128,64) | db3 (128),
 *                         with respect to the (BatchNorm-folded) w, b the caller passed.
 * The layer is re-evaluated inside the kernel (nothing of size B x N x 64 x C is read or written except grad_nb); weight
 * gradients are summed per wave, per workgroup and over workgroups in fixed orders: results repeat bit for bit.
 * workspace: mcp_fusion_grad_workspace_bytes(b, n) bytes of device memory (the workgroups' partial weight gradients); w3 16-byte
 * aligned.  ReLU and the channel max take torch's subgradients (0 at 0; one arg-max channel), |r| has gradient 0 at r = 0. */
int mcp_fusion_grad_floats(void);
size_t mcp_fusion_grad_workspace_bytes(int b, int n);
int mcp_fusion_grad(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
                    const float *w2, const float *b2, const float *w3, const float *b3, const float *grad_out, float *grad_p1, float *grad_nb,
                    float *grad_weights, void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* The fusion layer in net.train() mode (train.py:130): its three Conv2d + BatchNorm2d + ReLU layers (mocopci.py:749-755, :810-816)
 * normalise with the statistics of the batch they are given -- one call here = one reference call = one set of statistics over
 * all b * n * 64 (point, neighbour) rows.  Arguments as mcp_fusion with the RAW conv weights (w, b), then
 *   bn    mcp_fusion_bn_floats() = 1024 floats, per layer (64, 64, 128 channels) mean | rstd | gamma | beta: the caller fills gamma
 *         and beta (the BatchNorm weight / bias), this call fills mean and rstd = 1 / sqrt(var + eps);
 *   var   64 + 64 + 128 floats: the biased batch variances (the caller's running-estimate update scales them by n / (n - 1)).
 * Four passes (statistics of layer 1, 2, 3, then the layer itself), nothing of size rows x channels is written.
 * workspace: mcp_fusion_bn_workspace_bytes(b, n) bytes. */
int mcp_fusion_bn_floats(void);
size_t mcp_fusion_bn_workspace_bytes(int b, int n);
int mcp_fusion_bn_forward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
                          const float *w2, const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var, float *out,
                          void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* Backward of mcp_fusion_bn_forward (one reference call; bn as that call left it).  grad_out (B,N,3).  The BatchNorm terms
 * dz = (gamma / sigma)(dy' - mean(dy') - zhat mean(dy' zhat)) need sums over all rows between the layers: four passes, the layer
 * re-evaluated in each as far as it is needed; dy2' and dy1' (rows x 64) pass through caller-provided scratch.  Scratch, rows = b n 64:
 * row_c (rows int32), row_dy, row_a (rows floats), dy2, dy1 (rows x 64 floats, 16-byte aligned).  Writes grad_p1 (B,N,3), grad_nb (B,N,64,3)
 * (scattered into dL/dp2 by the caller, as for mcp_fusion_grad), grad_weights (12800 floats in mcp_fusion_grad's layout; the conv
 * biases' entries are 0: a bias in front of a batch-statistics BatchNorm has no gradient) and grad_affine (512 floats: dgamma1 |
 * dbeta1 | dgamma2 | dbeta2 | dgamma3 | dbeta3).  All sums in fixed orders.  workspace: mcp_fusion_bn_grad_workspace_bytes(b, n). */
size_t mcp_fusion_bn_grad_workspace_bytes(int b, int n);
int mcp_fusion_bn_backward(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
                           const float *w2, const float *b2, const float *w3, const float *b3, const float *bn, const float *grad_out, int *row_c,
                           float *row_dy, float *row_a, float *dy2, float *dy1, float *grad_p1, float *grad_nb, float *grad_weights,
                           float *grad_affine, void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* The pair that keeps what the backward's first pass would otherwise re-evaluate the whole layer for: per (point, neighbour) row
 * (rows = b * n * 64) the neighbour's score (the maximum over the layer-3 channels, floored at 0), the channel it sits at and zhat3 there
 * -- save_c (int32), save_z, save_s (floats), caller-owned, 12 bytes per row.  mcp_fusion_bn_forward_save = mcp_fusion_bn_forward + these
 * outputs (same out / bn / var bit for bit); mcp_fusion_bn_backward_saved = mcp_fusion_bn_backward with a first pass that reads them
 * (same gradients bit for bit). */
int mcp_fusion_bn_forward_save(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
                               const float *w2, const float *b2, const float *w3, const float *b3, float eps, float *bn, float *var, float *out, int *save_c,
                               float *save_z, float *save_s, void *workspace, size_t workspace_bytes, mcp_stream_t stream);
int mcp_fusion_bn_backward_saved(int b, int n, int nb, const float *p1, const float *p2, const int *idx, const int *idx2, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *w3, const float *b3, const float *bn, const float *grad_out,
                                 const int *save_c, const float *save_z, const float *save_s, int *row_c, float *row_dy, float *row_a, float *dy2, float *dy1,
                                 float *grad_p1, float *grad_nb, float *grad_weights, float *grad_affine, void *workspace, size_t workspace_bytes,
                                 mcp_stream_t stream);

/* Cost-volume cross() after its neighbour searches (pointconv_util.py:750-781, :894-922, :1126-1161):
 * xyz1 (B,N1,3), xyz2 (B,N2,3), points1 (B,N1,D), points2 (B,N2,D) channel-last (16-byte aligned),
 * idx (B,N1,32) int32 into set 2 (16 feature-cosine + 16 xyz neighbours) -> out (B,N1,D) = max over the 32
 * neighbours of LeakyReLU(wmlp . LeakyReLU(points2[idx] + points1 + wpos.(xyz2[idx]-xyz1) + bpos) + bmlp).
 * The layer's weights -- wpos (D,3), bpos (D) = the Conv2d 3->D; wmlp (D,D), bmlp (D) = the single Conv2d D->D of
 * the mlp list (every layer MoCoPCI builds has exactly one) -- are packed ONCE per layer by mcp_cross_pack into
 * the MFMA-operand image (mcp_cross_packed_floats(D) floats, 16-byte aligned, caller-owned).  D in {64,128,256}, k = 32.  Neighbour lists: idx (B,N1,32), or with idx2 != NULL
 * the 16 + 16 halves as two (B,N1,16) lists (feature-space neighbours, then coordinate-space neighbours).  bmap != NULL (B int32):
 * the batch replicates / selects a smaller one -- element bb of the tensors flagged in `shared` (1 points1, 2 points2, 4 the
 * first index list) is read from element bmap[bb] of the unreplicated tensor (Multiframe_Attention's three flow iterations share
 * their features, mocopci.py:191-197); xyz1, xyz2 and idx2 are per element.  The map is kept in LDS: b <= 1024 when one is passed
 * (MCP_ERR_UNSUPPORTED beyond). */
int mcp_cross_packed_floats(int d);
int mcp_cross_pack(int d, const float *wpos, const float *bpos, const float *wmlp, const float *bmlp, float *packed,
                   mcp_stream_t stream);
int mcp_cross_volume(int b, int n1, int n2, int d, int k, const float *xyz1, const float *xyz2, const float *points1,
                     const float *points2, const int *idx, const int *idx2, const int *bmap, int shared, const float *packed, float *out, mcp_stream_t stream);

/* Backward of mcp_cross_volume for one cross layer given by its own weights (the reference differentiates pointconv_util.py:765-781
 * with autograd over three materialised (B,D,32,N1) tensors; its hand-written backward pieces are the atomicAdd scatters of
 * group_points_gpu.cu:8-44).  xyz1, xyz2, points1, points2, idx / idx2 as mcp_cross_volume (no batch map); wpos (D,3), bpos (D),
 * wmlp (D,D), bmlp (D) row-major; grad_out (B,N1,D) = dL/dout.  Writes
 *   grad_xyz1 (B,N1,3), grad_points1 (B,N1,D);
 *   grad_dir (B,N1,32,3) and grad_rows (B,N1,32,D): dL/d(xyz2[idx]) and dL/d(points2[idx]) per gathered neighbour, in the order of
 *       the neighbour list(s) -- the caller scatters them with mcp_group_rows_grad_sorted (deterministic);
 *   grad_weights: mcp_cross_grad_floats(d) floats = dWpos (D,3) | dbpos (D) | dWmlp (D,D) | dbmlp (D).
 * The layer is re-evaluated in the kernel; the arg-max neighbour of a channel is the lowest list position among equal maxima;
 * all sums run in fixed orders (results repeat bit for bit).  workspace: mcp_cross_grad_workspace_bytes(b, n1, d) bytes.
 * D = 64 and 128 (the level-1 and level-2 cost volumes, 97 % of the layer's backward time at the training shape); D = 256:
 * MCP_ERR_UNSUPPORTED, mcp_cross_grad_floats returns 0 (the caller differentiates the unfused layer).  workspace 16-byte aligned. */
int mcp_cross_grad_floats(int d);
size_t mcp_cross_grad_workspace_bytes(int b, int n1, int d);
int mcp_cross_grad(int b, int n1, int n2, int d, int k, const float *xyz1, const float *xyz2, const float *points1, const float *points2,
                   const int *idx, const int *idx2, const float *wpos, const float *bpos, const float *wmlp, const float *bmlp,
                   const float *grad_out, float *grad_xyz1, float *grad_dir, float *grad_points1, float *grad_rows, float *grad_weights,
                   void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* PointConv / PointConvD up to the final Linear (mocopci.py:1218-1266, :1289-1300, :1330-1335):
 * s_xyz (B,N,3), new_xyz (B,S,3) centres, s_points (B,N,D) channel-last, idx (B,S,32) int32 into the
 * N source points; WeightNet Conv2d 3->8->8->8 + ReLU as w0 (8,3), w1 (8,8), w2 (8,8) and biases.
 * out (B,S,(3+D)*8), 16-byte aligned: out[b,s,c*8+m] = sum_k [dxyz|feat][k][c] * weightnet(dxyz[k])[m],
 * i.e. the tensor the reference feeds to self.linear.  k must be 32. */
int mcp_pointconv_agg(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points,
                      const int *idx, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                      const float *b2, float *out, mcp_stream_t stream);

/* Backward of mcp_pointconv_agg (the reference differentiates mocopci.py:1330-1335 with autograd over the materialised (B,S,32,3+D)
 * grouping; its hand-written backward pieces are the atomicAdd scatters of group_points_gpu.cu:8-44).  Arguments as
 * mcp_pointconv_agg, plus grad_out (B,S,(3+D)*8) = dL/dout, 16-byte aligned.  Writes
 *   grad_new_xyz (B,S,3);
 *   grad_gxyz (B,S,32,3) and grad_rows (B,S,32,D): dL/d(s_xyz[idx]) and dL/d(s_points[idx]) per gathered neighbour -- the caller
 *       scatters them with mcp_group_rows_grad_sorted (deterministic);
 *   grad_weights: mcp_pointconv_agg_grad_floats() = 176 floats: dW0 (8,3) | db0 (8) | dW1 (8,8) | db1 (8) | dW2 (8,8) | db2 (8).
 * The WeightNet is re-evaluated per (centre, neighbour) pair; all sums run in fixed orders (results repeat bit for bit).
 * workspace: mcp_pointconv_agg_grad_workspace_bytes(b, s) bytes.  k = 32; d a multiple of 4, 4 <= d <= 256. */
int mcp_pointconv_agg_grad_floats(void);
size_t mcp_pointconv_agg_grad_workspace_bytes(int b, int s);
int mcp_pointconv_agg_grad(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points, const int *idx,
                           const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *grad_out, float *grad_new_xyz, float *grad_gxyz, float *grad_rows, float *grad_weights,
                           void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* PointConv / PointConvD after the sampling as ONE launch (models/m_models/mocopci.py:1330-1342 and :1381-1393: group + WeightNet +
 * matmul + Linear((3+D)*8 -> C_out) + LeakyReLU): the (B,S,(3+D)*8) aggregate stays on the compute unit.  Arguments as
 * mcp_pointconv_agg, then `packed`: the mcp_linear_pack image of the Linear's (C_out, (3+D)*8) weight (one piece) with its bias;
 * `slope`: the activation's negative slope (0.1 in the reference; 1 = none); out (B,S,C_out).
 * (D, C_out) = (32, 32) or (64, 64), k = 32; anything else returns MCP_ERR_UNSUPPORTED (use mcp_pointconv_agg + mcp_linear).
 * From 16384 rows (B*S) up the result is bit-identical to mcp_pointconv_agg followed by mcp_linear (same operand split, same
 * summation order); below that mcp_linear sums K in four parts and the two differ in the last bits. */
int mcp_pointconv_linear(int b, int n, int s, int d, int k, const float *s_xyz, const float *new_xyz, const float *s_points,
                         const int *idx, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                         const float *b2, const float *packed, int c_out, float slope, float *out, mcp_stream_t stream);

/* Multi-head attention with a tiny head dim (8 or 16) in exact fp32, flash style (no (N,N) score tensor):
 * the attention core of InterFrameAttentionInterpretation (mocopci.py:650-667) and CrossAttention (:72-86).
 * q (BF,Nq,*), k/v (BF,Nk,*) are read in place from the projection outputs: element [bf, token, head*hd + d]
 * at ptr + (bf*N + token)*stride + head*hd + d (strides in floats, multiples of 4; pointers 16-byte aligned);
 * out (BF,Nq,*) likewise.  softmax(q.k^T * scale) . v per (bf, head). */
int mcp_attention_small(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                        const float *v, int v_stride, float scale, float *out, int out_stride, mcp_stream_t stream);

/* Same contract for wide heads, hd in {32, 64, 256} (EI cross-former of level 3, mocopci.py:72-86 at dim 256 / 8 heads;
 * Cross_Frame_Att, mocopci.py:499-522, whose head slots are C = 256 wide): S = QK^T and PV both on fp32 MFMA, online softmax,
 * K/V streamed through LDS in 32-key tiles; nothing of size Nq x Nk is written. */
int mcp_attention_wide(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                        const float *v, int v_stride, float scale, float *out, int out_stride, mcp_stream_t stream);

/* Both kernels behind one entry point (hd in {8, 16, 32, 64, 256}), with a rotation of the key / value batch: batch element bf of
 * the queries attends to keys / values of batch element (bf + kv_batch_shift) mod BF (0 <= kv_batch_shift < BF).  EI_Crossformer
 * (mocopci.py:147-151) runs Injector(x1 -> x2) and Extractor(x2 -> x1) on the two halves of one stacked batch: one launch with
 * kv_batch_shift = BF / 2 instead of two. */
int mcp_attention(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                  const float *v, int v_stride, int kv_batch_shift, float scale, float *out, int out_stride, mcp_stream_t stream);

/* mcp_attention_small with attention dropout (net.train(): mocopci.py:660-662 / :80-82 drop entries of the softmax matrix): out =
 * dropout(softmax(q k^T scale), drop_p) v per head; the softmax is normalised before the mask, kept entries are scaled by
 * 1 / (1 - drop_p).  The mask is a counter-based hash of (seed, batch, head, query, key): the caller draws `seed` from its generator
 * per call; the backward regenerates the mask from the same seed.  Not the reference generator's mask for a given torch seed (no
 * kernel can be: the reference materialises the matrix and draws it with torch's Philox stream), the same distribution. */
int mcp_attention_small_dropout(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                                int v_stride, float scale, float drop_p, unsigned seed, float *out, mcp_stream_t stream);

/* The same forward that also writes the rows' log-sum-exp (log2 domain; lse (BF, heads, Nq) floats) for mcp_attention_small_grad_lse: a
 * training forward keeps it and the backward skips the statistics pass (the Q K^T products and exponentials of the whole score matrix a
 * second time).  drop_p = 0: no mask, the result of mcp_attention_small bit for bit. */
int mcp_attention_small_lse(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                            int v_stride, float scale, float drop_p, unsigned seed, float *out, float *lse, mcp_stream_t stream);

/* Backward of mcp_attention_small / mcp_attention_small_dropout (head_dim 8 / 16; the reference differentiates the materialised
 * softmax of mocopci.py:72-86, :650-667 with autograd).  q, k, v, strides, scale as the forward (no key / value batch shift);
 * drop_p, seed as the forward's (0: no dropout); out (BF,Nq,heads*hd) the forward's output, grad_out its gradient, both dense.
 * Writes grad_q (BF,Nq,heads*hd) and grad_kv (BF,Nk,2*heads*hd) laid out [dK | dV] like the reference's kv projection.  Three
 * kernels (row statistics, dQ, dK/dV); nothing of size Nq x Nk is written; fixed summation orders.
 * workspace: mcp_attention_small_grad_workspace_bytes(bf, nq, heads) bytes. */
size_t mcp_attention_small_grad_workspace_bytes(int bf, int nq, int heads);
int mcp_attention_small_grad(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                             int v_stride, float scale, float drop_p, unsigned seed, const float *out, const float *grad_out, float *grad_q,
                             float *grad_kv, void *workspace, size_t workspace_bytes, mcp_stream_t stream);
/* mcp_attention_small_grad with the log-sum-exp the forward kept (mcp_attention_small_lse): one elementwise pass for D = dO . O, then the
 * dQ and dK / dV kernels.  Same workspace size. */
int mcp_attention_small_grad_lse(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                                 int v_stride, float scale, float drop_p, unsigned seed, const float *out, const float *grad_out, const float *lse,
                                 float *grad_q, float *grad_kv, void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* Row normalisation with the additions in front of it (nn.LayerNorm semantics: biased variance, eps inside the root):
 *     z = x[r] (+ y[r]) (+ bias);   out[r] = (z - mean z) * rsqrt(var z + eps) (* gamma + beta)
 * x, y, out (rows, c) with row strides in floats; y, bias, gamma, beta may be NULL; c <= 1024.  Replaces the LayerNorm launches
 * of EI_Crossformer (mocopci.py:100-145: query_norm / feat_norm / ffn_norm) -- their affine parts fold into the Linear behind
 * them -- and the residual additions in front of ffn_norm. */
int mcp_add_layernorm(long long rows, int c, const float *x, long long x_stride, const float *y, long long y_stride,
                      const float *bias, const float *gamma, const float *beta, float eps, float *out, long long out_stride,
                      mcp_stream_t stream);

/* Inputs of Multi_Frame_Att (mocopci.py:200-208, :551-557) in one pass: per output row r (a (sample, frame) pair),
 *     x[r]  = fea[src_self[r]] + te_self[r]                    the flow embedding plus its time code
 *     xn[r] = scale * x[r] + shift                             norm1 in eval mode (per-channel affine)
 *     xr[r] = scale * (fea[src_partner[r]] + te_partner[r]) + shift     the attention partner, frame R-1-f of the flipped stack
 * fea (members, n, c) holds the computed members in any order; src_* (rows) index it; te_* (rows, c); x / xn / xr (rows, n, c).
 * c a multiple of 4, pointers 16-byte aligned. */
int mcp_mfa_prepare(int rows, int n, int c, const float *fea, const int *src_self, const int *src_partner, const float *te_self,
                    const float *te_partner, const float *scale, const float *shift, float *x, float *xn, float *xr, mcp_stream_t stream);

/* chamfer_loss (models/utils.py:36-45 -> pytorch3d chamfer_distance defaults): per-point squared
 * nearest distance both ways.  x (B,N,3), y (B,M,3) -> dxy (B,N), dyx (B,M); the caller takes the means. */
int mcp_chamfer_nn(int b, int n, int m, const float *x, const float *y, float *dxy, float *dyx, mcp_stream_t stream);

/* Per-point Linear (1x1 convolution) with fused epilogue for the tall-skinny shapes of the caller graph (Conv1d wrapper
 * mocopci.py:1111-1127, the Linear layers of :438-468 / :821-1059, and the concatenations in front of them, :186-187, :846-847):
 *     out[r, 0:n] = act( sum_i W_i . x_i[r] + b ) [+ res[r]],   act(v) = v > 0 ? v : slope * v   (slope 1: none, 0: ReLU, 0.1: LeakyReLU)
 * The input is given as nseg <= 3 pieces x[i] (rows, k_seg[i]) with row strides x_stride[i] (floats; 16-byte aligned rows,
 * k_seg[i] a multiple of 4): the pieces of what the reference concatenates, read in place.  W is (n, sum k_seg) row-major over
 * the concatenated K axis; n <= 128, or 129..192, or 193..256, or wider in column blocks of 128 (n <= 2048, ceil(n/32) a multiple of 4).  mcp_linear_pack prepares (W, b) once into
 * mcp_linear_packed_floats(n, nseg, k_seg) caller-owned floats (0 = unsupported shape); b may be NULL. */
int mcp_linear_packed_floats(int n, int nseg, const int *k_seg);
/* Narrow-output Linear with the activation on its input: out[r, 0:n] = b + W . act(x[r]), act(v) = v > 0 ? v : in_slope v; n <= 4,
 * k in {256, 512, 1024}; W (n, k) row-major, x rows 16-byte aligned with stride x_stride floats.  The tail of Mlp_T where only the
 * flow is read (PReLU, then fc2 and mapping_xyz folded into one 4C -> 3 map, mocopci.py:1561-1565 with :566-567 / :510-511). */
int mcp_linear_narrow(long long rows, int k, int n, const float *x, int x_stride, const float *w, const float *b, float in_slope,
                      float *out, int out_stride, mcp_stream_t stream);
int mcp_linear_pack(int n, int nseg, const int *k_seg, const float *w, const float *b, float *packed, mcp_stream_t stream);
int mcp_linear(long long rows, int n, int nseg, const float *const *x, const int *x_stride, const int *k_seg, float slope,
               const float *packed, const float *res, int res_stride, float *out, int out_stride, mcp_stream_t stream);
/* mcp_linear with the kernel chosen as for `policy_rows` rows (the few-row split-K form sums K in another order than the full-K
 * form): a caller that computes a SUBSET of the rows of a tall product -- the sampled rows of a PointConvD whose every candidate
 * row another code path computes (mocopci_amd/model.py: speculative / deferred forms of one forward) -- gets the same bits. */
int mcp_linear_as(long long rows, long long policy_rows, int n, int nseg, const float *const *x, const int *x_stride, const int *k_seg,
                  float slope, const float *packed, const float *res, int res_stride, float *out, int out_stride, mcp_stream_t stream);

/* Weight and bias gradient of a tall Linear y = act(x W^T + b) (autograd over mocopci.py:1111-1127 and the per-point Linears of the
 * caller graph under train.py:162): dw (n,k) = gz^T x, db (n) = column sums of gz, for gz (rows,n) = gy * act'(z) formed by the
 * caller and x (rows,k), both row-major with the given row strides (floats).  The rows are
 * the contraction axis of f32-input MFMAs; workgroup partials are added in workgroup order (bit-reproducible; the number of
 * workgroups depends on `rows` alone).  n <= 256 and ceil(n/32) ceil(k/32) <= 64 (widths that are not multiples of 32, odd strides:
 * read element by element, zero-padded); otherwise MCP_ERR_UNSUPPORTED / a workspace size of 0.  db may be NULL.  workspace: mcp_linear_wgrad_workspace_bytes(rows, n, k) caller-owned bytes. */
size_t mcp_linear_wgrad_workspace_bytes(long long rows, int n, int k);
int mcp_linear_wgrad(long long rows, int n, int k, const float *gz, int gz_stride, const float *x, int x_stride, float *dw, float *db,
                     void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* PReLU (one slope) followed by dropout on the hidden activation of Mlp_T under net.train() (mocopci.py:1558-1565, :1592-1595: `self.act`
 * then `self.drop`), and its backward, as one elementwise pass each:
 *     out[i]    = m_i * (z[i] > 0 ? z[i] : a z[i]),      m_i = 1 / (1 - drop_p) for a kept element, 0 for a dropped one
 *     grad_z[i] = grad_out[i] m_i (z[i] > 0 ? 1 : a),    grad_slope[0] = sum_i grad_out[i] m_i min(z[i], 0)
 * `slope` points at the layer's slope ON THE DEVICE (the live parameter).  The mask is a counter-based hash of (seed, i): the caller
 * draws `seed` from its generator per call and passes the same seed to the backward, which regenerates the mask -- only z is kept
 * between the passes (the reference keeps z, the mask and the dropped activation).  Same distribution as the reference's mask, not the
 * same draws for a given torch seed (see mcp_attention_small_dropout).  grad_slope is summed in a fixed order (workgroup count a
 * function of `total` alone): bit-reproducible.  drop_p outside [0, 1): MCP_ERR_BAD_ARG.
 * workspace: mcp_prelu_dropout_grad_workspace_bytes(total) caller-owned bytes. */
int mcp_prelu_dropout(long long total, const float *z, const float *slope, float drop_p, unsigned seed, float *out, mcp_stream_t stream);
size_t mcp_prelu_dropout_grad_workspace_bytes(long long total);
int mcp_prelu_dropout_grad(long long total, const float *z, const float *slope, const float *grad_out, float drop_p, unsigned seed,
                           float *grad_z, float *grad_slope, void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* Fused two-layer per-point MLP (Mlp_T of Multi_Frame_Att, mocopci.py:1558-1565 inside :551-575, and the flow heads
 * trans_block / trans_block_2 -> mapping_xyz, :566-567 / :510-511):
 *     out[r, 0:cout] = (res ? res[r] : 0) + b2 + W2 . act(W1 . x[r] + b1),   act(v) = v > 0 ? v : slope * v   (PReLU with one slope)
 * x (rows, cin) with row stride x_stride floats (16-byte aligned rows), W1 (hidden, cin), W2 (cout, hidden); the (rows, hidden)
 * activation is never written.  Supported (the shapes where it beats the BLAS chain): cin 64 (cout <= 64), cin 128 (cout <= 32 or 97..128);
 * hidden a multiple of 32.  The weights are prepared once by mcp_mlp2_pack into mcp_mlp2_packed_floats(cin, hidden, cout) caller-owned
 * floats (0 = unsupported shape).  Depthwise k=1 convolutions and eval-mode BatchNorms around the first layer are affine and are
 * folded into (W1, b1) by the caller. */
int mcp_mlp2_packed_floats(int cin, int hidden, int cout);
int mcp_mlp2_pack(int cin, int hidden, int cout, const float *w1, const float *b1, const float *w2, const float *b2, float *packed,
                  mcp_stream_t stream);
int mcp_mlp2(long long rows, int cin, int hidden, int cout, float slope, const float *x, int x_stride, const float *res, int res_stride,
             const float *packed, float *out, int out_stride, mcp_stream_t stream);

/* Point-Transformer vector attention (TransformerBlock.forward, models/pointT_layer2.py:58-77; d_model 64, k 16) after the
 * neighbour search and the q/k/v projections: xyz (B,N,3), q/kf/vf (B,N,64) channel-last (16-byte aligned) with a common row
 * stride of qkv_stride floats (64 for separate tensors, 192 when they are slices of one packed projection), idx (B,N,16)
 * -> out (B,N,64) = sum_j softmax_j(fc_gamma(q_i - k_j + delta_j) / 8) * (v_j + delta_j), delta_j = fc_delta(xyz_i - xyz_j).
 * fc_delta = Linear(3,64) ReLU Linear(64,64) as (wd1,bd1,wd2,bd2); fc_gamma = Linear(64,64) ReLU Linear(64,64) as
 * (wg1,bg1,wg2,bg2); packed once per block by mcp_ptblock_pack (mcp_ptblock_packed_floats() floats, caller-owned). */
int mcp_ptblock_packed_floats(void);
int mcp_ptblock_pack(const float *wd1, const float *bd1, const float *wd2, const float *bd2, const float *wg1, const float *bg1,
                     const float *wg2, const float *bg2, float *packed, mcp_stream_t stream);
int mcp_ptblock_attention(int b, int n, int c, int k, int qkv_stride, const float *xyz, const float *q, const float *kf,
                          const float *vf, const int *idx, const float *packed, float *out, mcp_stream_t stream);

/* Backward of mcp_ptblock_attention for a block given by its own weights wd1 (64,3), bd1, wd2 (64,64), bd2 (fc_delta), wg1, bg1, wg2,
 * bg2 (fc_gamma) (the reference differentiates pointT_layer2.py:64-75 with autograd over five (B,N,16,64) tensors).  xyz, q, kf,
 * vf, idx, qkv_stride as mcp_ptblock_attention; grad_out (B,N,64).  Writes
 *   grad_q (B,N,64); grad_xyz_c (B,N,3): the centre's share of dL/dxyz;
 *   grad_xyz_rows (B,N,16,3), grad_k_rows, grad_v_rows (B,N,16,64): per gathered neighbour, for the caller's deterministic scatter
 *       (mcp_group_rows_grad_sorted) into dL/dxyz, dL/dk, dL/dv;
 *   grad_weights: mcp_ptblock_grad_floats() floats = dWd1 (64,3) | dbd1 | dWd2 (64,64) | dbd2 | dWg1 | dbg1 | dWg2 | dbg2.
 * The block is re-evaluated in the kernel; all sums run in fixed orders.  workspace: mcp_ptblock_grad_workspace_bytes(b, n),
 * 16-byte aligned.  c = 64, k = 16. */
int mcp_ptblock_grad_floats(void);
size_t mcp_ptblock_grad_workspace_bytes(int b, int n);
int mcp_ptblock_grad(int b, int n, int c, int k, int qkv_stride, const float *xyz, const float *q, const float *kf, const float *vf, const int *idx,
                     const float *wd1, const float *bd1, const float *wd2, const float *bd2, const float *wg1, const float *bg1, const float *wg2,
                     const float *bg2, const float *grad_out, float *grad_xyz_c, float *grad_xyz_rows, float *grad_q, float *grad_k_rows,
                     float *grad_v_rows, float *grad_weights, void *workspace, size_t workspace_bytes, mcp_stream_t stream);

/* Approximate Earth Mover's Distance (metric of test.py:90): approxmatch + matchcost of
 * models/EMD/cuda/emd_kernel.cu:29-162, :204-247 (emd_cuda.approxmatch_forward / matchcost_forward, emd.py:11-12).
 * xyz1 (B,N,3), xyz2 (B,M,3) -> cost (B) = sum_{l,k} match[l][k] |xyz2[l]-xyz1[k]|^2.  match (B,M,N) is written
 * only if non-NULL (the cost is accumulated while the transfers are produced).  workspace: B*(3N+2M) floats. */
int mcp_emd(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *cost, float *workspace,
            mcp_stream_t stream);

/* ---------------- Instrumentation (bench.py roofline leg) --------------------------- */
/* mcp_prof_enable(mask): every launch of a kernel whose id bit (1 << MCP_KERNEL_*) is set in mask is bracketed
 * by hipEvents recorded on the launch stream; mask 0 disables and clears.  mcp_prof_collect(id, ...) synchronises
 * that kernel's events and returns its launch count and total milliseconds.  Off by default. */
#define MCP_KERNEL_FPS 1
#define MCP_KERNEL_KNN 2
#define MCP_KERNEL_GROUP_ROWS 3
#define MCP_KERNEL_INTERP3 4
#define MCP_KERNEL_KNN_COSINE 5
#define MCP_KERNEL_FUSION 6
#define MCP_KERNEL_CROSS 7
#define MCP_KERNEL_POINTCONV 8
#define MCP_KERNEL_ATTENTION 9
#define MCP_KERNEL_PTBLOCK 10
#define MCP_KERNEL_MLP 11
#define MCP_KERNEL_LINEAR 12
int mcp_prof_enable(int kernel_mask);
int mcp_prof_collect(int kernel_id, int *launches, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MOCOPCI_HIP_H */
