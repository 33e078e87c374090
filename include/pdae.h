/*
 * pdae.h -- C ABI of the MI355X-native Point-DAE hot path (libpdae_hip.so).
 *
 * This is the drop-in boundary for the pretraining step's native operators.
 * The reference (YBZh/Point-DAE) reaches its CUDA kernels through pybind11
 * torch extensions; every entry below names the reference host function it
 * replaces (file:line under the reference tree) and keeps that function's
 * argument meaning.  Differences that hold for every entry:
 *
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     to contiguous row-major buffers owned by the caller.  Nothing is
 *     allocated, retained or freed by the library.
 *   - every entry takes an explicit stream (a hipStream_t passed as void*);
 *     the reference launches chamfer/emd on the legacy default stream
 *     (chamfer.cu:159, emd_kernel.cu:188) and pointnet2 on torch's current
 *     stream (sampling_gpu.cu:183).  Calls are asynchronous and re-entrant.
 *   - return value: PDAE_OK (0) or a negative pdae_status; the reference
 *     either printf()s (chamfer.cu:166-169) or exit(-1)s (cuda_utils.h:32-41).
 *   - outputs are fully written by the call (the reference relies on
 *     torch::zeros in its host wrappers; here the zero-fill, where the
 *     semantics need one, is part of the call).
 *
 * Arithmetic contract (shared with oracle/pdae_oracle.c): fp32, squared
 * distances evaluated as ((dx*dx + dy*dy) + dz*dz) with every operation
 * rounded (no FMA contraction), indices by the reference's tie rules.
 */
#ifndef PDAE_H
#define PDAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pdae_stream_t; /* hipStream_t */

enum pdae_status {
  PDAE_OK = 0,
  PDAE_ERR_BAD_ARG = -1,     /* null pointer / negative size / unsupported size */
  PDAE_ERR_LAUNCH = -2,      /* hipGetLastError() != hipSuccess after launch    */
  PDAE_ERR_UNSUPPORTED = -3  /* shape outside what the kernels implement         */
};

/* Library identification: returns "pdae-hip gfx950 <abi version>". */
const char* pdae_version(void);
/* Text for the last failing status on this thread (hipGetErrorString or arg). */
const char* pdae_last_error(void);

/* Contexts.  The library's mutable state -- the deterministic-mode workspace, the parked (deferred) reductions, the
 * GEMM arithmetic -- lives in a context; every entry point works on the calling thread's CURRENT context, which is
 * the process default context until the thread sets one (the model of the CUDA / HIP runtime's current device; the
 * reference's extensions keep no state at all, so it has nothing to bind here).  A host that drives several streams
 * from several threads gives each thread its own context (and its own workspaces): nothing is shared between them.
 * One context must not be used by two threads at once.  Everything else the library holds (per-kernel launch
 * attributes set at first use) is immutable after its first write. */
typedef struct pdae_ctx* pdae_ctx_t;
int pdae_ctx_create(pdae_ctx_t* out);
int pdae_ctx_destroy(pdae_ctx_t ctx);          /* not while it is current on the calling thread */
int pdae_ctx_set_current(pdae_ctx_t ctx);      /* NULL: back to the process default context */
pdae_ctx_t pdae_ctx_current(void);             /* NULL when the thread is on the default context */

/* Deterministic mode.  The reference's reductions (cuDNN batch-norm statistics, cuBLAS split-K
 * weight gradients, ATen layer_norm backward: models/PointCAE_transformer.py:37-51, 94-147) and,
 * by default, this library's end in float atomics whose order of arrival varies from launch to
 * launch.  With a device workspace registered here (>= 1 MiB; 64 MiB covers the pretraining step
 * at batch 128) the batch-norm statistics, the embedder's weight / bias gradients, the LayerNorm
 * parameter gradients and the column sums store per-block partials and add them in block order:
 * bit-identical results from run to run and from hipGraph replay to eager launch, for one small
 * extra launch per reduction.  The workspace is shared by all launches of its CONTEXT: one stream at a time per context.
 * workspace == NULL switches the mode off.  A reduction that needs more than `bytes` fails with
 * PDAE_ERR_UNSUPPORTED.  Not covered (LDS float atomics, kept): group_points_grad,
 * three_interpolate_grad and the large-cloud Chamfer gradient of the PointNet++ configuration. */
int pdae_set_deterministic(void* workspace, size_t bytes);
int pdae_deterministic(void);   /* 1 when a workspace is registered */

/* Deferred column reductions.  Between deferred_begin and deferred_flush the LayerNorm backward entries park their
 * per-block parameter-gradient partials in `workspace` instead of finishing with float atomics; deferred_flush adds
 * the partials of all parked calls (up to 48) in block order into their outputs with ONE launch.  The outputs
 * (dgamma, dbeta, dbias) are complete only after the flush: for callers that consume them later, such as the
 * hipGraph-replayed step, whose gather copy follows the whole backward (graph_step.py).  A call that does not fit
 * the workspace simply behaves as usual.  Host-side state: one stream / one thread at a time. */
int pdae_deferred_begin(void* workspace, size_t bytes);
int pdae_deferred_flush(pdae_stream_t stream);
/* rows_wgrad parks its partial-tile reduction in the same window (the dW / db outputs and the workspace of the call
 * must then stay untouched until the flush, which adds the tiles of up to 8 parked launches per reduction launch).
 * deferred_hold(1) ... deferred_hold(0) brackets calls whose results the caller reads right away. */
int pdae_deferred_hold(int hold);

/* ------------------------------------------------------------------------
 * Farthest point sampling.
 * Replaces furthest_point_sampling_kernel_wrapper(b, n, m, dataset, temp, idxs)
 *   extensions/pointnet2/_ext_src/src/sampling_gpu.cu:178-229 (kernel :72-176),
 *   host alloc sampling.cpp:67-88 (third-party pointnet2_ops twin called from
 *   utils/misc.py:18).
 * dataset (b,n,3) f32 -> idxs (b,m) i32.  idxs[:,0]=0; points with
 * x*x+y*y+z*z <= 1e-3 are never selected; ties resolved as the reference's
 * 512-thread strided scan + lower-tid tree reduce.  The reference's `temp`
 * scratch (b,n) lives in registers here and is not part of the ABI.
 * If `centres` is non-null it receives dataset[idxs] as (b,m,3) f32 (the
 * gather_operation + 2 transposes of utils/misc.py:19 fused).
 */
int pdae_furthest_point_sampling(int b, int n, int m, const float* dataset,
                                 int32_t* idxs, float* centres /*nullable*/,
                                 pdae_stream_t stream);

/* gather_points_kernel_wrapper(b, c, n, npoints, points, idx, out)
 *   sampling_gpu.cu:25-33: out[b,c,j] = points[b,c,idx[b,j]].            */
int pdae_gather_points(int b, int c, int n, int npoints, const float* points,
                       const int32_t* idx, float* out, pdae_stream_t stream);
/* gather_points_grad_kernel_wrapper, sampling_gpu.cu:52-60: scatter-add of
 * grad_out (b,c,npoints) into grad_points (b,c,n); grad_points is zero-filled
 * by this call (sampling.cpp:50-52 does it with torch::zeros).            */
int pdae_gather_points_grad(int b, int c, int n, int npoints,
                            const float* grad_out, const int32_t* idx,
                            float* grad_points, pdae_stream_t stream);

/* query_ball_point_kernel_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx)
 *   ball_query_gpu.cu:49-57 (kernel :12-47), host ball_query.cpp:11-35.
 * idx (b,m,nsample) i32: first nsample indices k (ascending) with d2 < r*r;
 * short lists padded with the first hit; empty ball -> zeros.              */
int pdae_ball_query(int b, int n, int m, float radius, int nsample,
                    const float* new_xyz, const float* xyz, int32_t* idx,
                    pdae_stream_t stream);

/* group_points_kernel_wrapper(b, c, n, npoints, nsample, points, idx, out)
 *   group_points_gpu.cu:33-43: out[b,c,j,k] = points[b,c,idx[b,j,k]].     */
int pdae_group_points(int b, int c, int n, int npoints, int nsample,
                      const float* points, const int32_t* idx, float* out,
                      pdae_stream_t stream);
/* group_points_grad_kernel_wrapper, group_points_gpu.cu:69-78; grad_points
 * (b,c,n) is zero-filled by this call (group_points.cpp:49-51).           */
int pdae_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float* grad_out, const int32_t* idx,
                           float* grad_points, pdae_stream_t stream);

/* QueryAndGroup.forward (extensions/pointnet2/pointnet2_utils.py:345-361) in ROW layout, the set-abstraction MLP's input:
 *   out[(b*np + p)*ns + s] = [ xyz[b, idx[b,p,s]] - new_xyz[b,p] | 0 | features[b*N + idx[b,p,s], 0:C] ]  (4 + C floats;
 *   features (B*N, C) rows, nullable with C = 0; C a multiple of 4; the zero column keeps the reduction length of the
 *   row GEMMs a multiple of 4).  One pass instead of grouping_operation x 2 + subtract + cat.
 * sa_group_rows_grad: dfeatures[b*N + j, c] = sum of dout[row, 4 + c] over the rows of cloud b whose idx is j (the
 *   backward of the two grouping_operation calls, group_points_gpu.cu:69-78, for the feature part; coordinates carry no
 *   gradient on this path).  Fully written (no zero-fill needed).  A cloud's rows are ordered by source point with a
 *   counting sort in LDS (integer atomics), then summed per point in registers: no float atomics; the order of a point's
 *   rows -- hence of its fp32 sum -- is the sort's order of arrival.  N <= 4096, 4 (2 N + 1 + np ns) <= 150 KB, C <= 1024. */
int pdae_sa_group_rows(int B, int N, int np, int ns, int C, const float* xyz, const float* new_xyz,
                       const int32_t* idx, const float* features /*nullable*/, float* out, pdae_stream_t stream);
int pdae_sa_group_rows_grad(int B, int N, int np, int ns, int C, const int32_t* idx, const float* dout,
                            float* dfeatures, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * k nearest neighbours.  Replaces KNN_CUDA 0.2 `knn(ref, query, k)` as used
 * through KNN(k, transpose_mode=True)(ref, query) at
 * models/PointCAE_transformer.py:59,76 (third-party wheel, not in the
 * reference tree; semantics restated in oracle/pdae_oracle.c).
 * ref (b,n,3), query (b,g,3) -> idx (b,g,k) int64 ascending by squared
 * distance, earlier index first on ties; dist (b,g,k) f32 = sqrtf(d2),
 * nullable.  Requires 1 <= k <= min(n, 64).
 * If `nbr` is non-null it receives ref[idx] - query as (b,g,k,3) f32, i.e.
 * the flat-index gather + centre subtraction of Group.forward
 * (models/PointCAE_transformer.py:79-85) fused into the same launch.
 */
int pdae_knn(int b, int n, int g, int k, const float* ref, const float* query,
             int64_t* idx, float* dist /*nullable*/, float* nbr /*nullable*/,
             pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Feature propagation (PointNet++ FP modules; segmentation / Point-M2AE heads).
 * Replaces three_nn_kernel_wrapper / three_interpolate_kernel_wrapper /
 *   three_interpolate_grad_kernel_wrapper, extensions/pointnet2/_ext_src/src/
 *   interpolate_gpu.cu:12-146 (Python: pointnet2_utils.py:118-204).
 * three_nn: unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED distances to
 *   the three nearest known points, ascending, idx (b,n,3) i32; strict `<`
 *   insertion as the reference (ties keep the earlier index; fewer than three
 *   known points leave +inf / index 0).  The Python wrapper returns sqrt(dist2).
 * three_interpolate: points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) =
 *   sum_k points[b,c,idx[b,n,k]] * weight[b,n,k].
 * three_interpolate_grad: grad_out (b,c,n) -> grad_points (b,c,m), fully written
 *   (the reference zero-fills and scatters with global atomics).
 */
int pdae_three_nn(int b, int n, int m, const float* unknown, const float* known,
                  float* dist2, int32_t* idx, pdae_stream_t stream);
int pdae_three_interpolate(int b, int c, int m, int n, const float* points,
                           const int32_t* idx, const float* weight, float* out,
                           pdae_stream_t stream);
int pdae_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                const int32_t* idx, const float* weight,
                                float* grad_points, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * In-forward patch corruption.  Fuses, for PointCAE_transformer.forward
 * (models/PointCAE_transformer.py:680-684):
 *     neighborhood += center; (t_nb, t_c) = corrupt_data(neighborhood, center);
 *     neighborhood -= center;  t_nb -= t_c
 * where corrupt_data (datasets/corrupt_util_tensor.py:706-727) applies 1-3
 * per-sample linear maps ('translate' / 'scale_nonorm' multiply by a 3-vector,
 * :59-116; 'rotate' / 'reflection' / 'shear' right-multiply by a 3x3 matrix,
 * :139-342).  The maps are drawn on the host with the reference's RNG calls and
 * passed as steps (nsteps, b, 10) f32: [kind (0 = multiply, 1 = matrix), then
 * 3 factors or 9 row-major matrix entries].
 * nbr (b,g,k,3) centre-subtracted, center (b,g,3) ->
 * gt_nbr (b,g,k,3) = (nbr + c) - c, t_nbr (b,g,k,3), t_center (b,g,3).
 */
int pdae_patch_affine(int b, int g, int k, int nsteps, const float* nbr,
                      const float* center, const float* steps, float* gt_nbr,
                      float* t_nbr, float* t_center, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Point-M2AE token pyramid: index bookkeeping of the hierarchical grouping.
 * Replaces the tensor chains of Group.forward, models/Point_M2AE_modules.py:239-247 (the flat neighbour index
 *   `idx + arange(B).view(-1,1,1) * num_points` that Group returns next to neighbourhood and centre) and of
 *   H_Encoder.forward, models/Point_M2AE.py:112-117 (multi-scale masking pushed one level down).  FPS and kNN of a
 *   level are pdae_furthest_point_sampling / pdae_knn.
 * flatten_group_index: idx (b, g*k) i64 local indices into clouds of n points -> flat (b*g*k) i64 = idx + cloud*n.
 * mask_propagate: parent_masked (parents) u8, flat_idx (parents*k) i64 into `children` finer tokens ->
 *   child_masked (children) u8, fully written: 1, except 0 at every child of a visible parent -- and at flat child 0
 *   whenever some parent is masked (the reference multiplies the indices of masked parents by 0 and scatters them
 *   too; kept bug-for-bug).
 */
int pdae_flatten_group_index(int b, int n, int gk, const int64_t* idx, int64_t* flat, pdae_stream_t stream);
int pdae_mask_propagate(int parents, int k, int children, const uint8_t* parent_masked, const int64_t* flat_idx,
                        uint8_t* child_masked, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Loader-side corruption on the device.  Replaces corrupt_dropout_local
 *   datasets/corrupt_util.py:590-612, which the reference runs per item in its
 *   DataLoader workers (ShapeNet55Dataset.__getitem__ :90-119).
 * xyz (b,p,3) f32; for cloud i, nclusters[i] <= 8 clusters c = 0..: the seed is
 * the seed_rank[i*8+c]-th surviving point in index order, the sizes[i*8+c]
 * survivors nearest to it (squared distance, ties by lower index) are dropped.
 * alive (b,p) u8 receives 1 for survivors.  The draws are inputs: the host side
 * draws them with the reference's distributions.  p <= 16384.
 */
int pdae_dropout_local(int b, int p, const float* xyz, const int32_t* nclusters,
                       const int32_t* seed_rank, const int32_t* sizes,
                       unsigned char* alive, pdae_stream_t stream);

/* The other loader-side stages (ShapeNet55Dataset.__getitem__ :90-119 = augment_data 'norm' -> random_sample ->
 * corrupt_data -> random_sample); the random draws are inputs (the host side draws them with the reference's
 * distributions; tests/golden/make_loader_fixtures.py records the live reference's).  Clouds live in (b, stride, 3)
 * buffers so that added points can be appended behind the p original ones.
 * pipeline_norm_affine: replaces _pc_normalize (corrupt_util.py:7-17; when `normalise`), the maps of 'affine_r3'
 *   (:1062-1070 -> corrupt_tranlate :130-140, corrupt_scale_nonorm_2p :82-92, corrupt_rotate_360 :241-263,
 *   corrupt_reflection :390-409, corrupt_shear_p5 :412-428) as nmaps[i] <= 3 maps of 12 floats (3x3 M row-major, t):
 *   y = x M + t applied in order, and corrupt_jitter (:179-191): y += sigma[i] * noise[i,k,:] (noise may be NULL).
 * pipeline_add_global: replaces corrupt_add_global (:830-841) with _sample_points_inside_unit_sphere (:42-56):
 *   u (b, nmax, 3) = the radius / cos(theta) / phi uniforms; count[i] points are written behind row p0.
 * pipeline_add_local: replaces corrupt_add_local (:844-870): added point j of cloud i = cloud point seed[i,j] +
 *   sigma[i,j] * noise[i,j,:], divided by its squared norm when that exceeds 1 (as the reference does).
 * pipeline_density: replaces density (:875-897): alive[i,k] &= dist01(xyz, view[i]) * gate[i] < r[i,k].
 * pipeline_subset: replaces ShapeNet.random_sample (:76-88): out (b, n, 3) = the n alive points of smallest key, in
 *   ascending key order (ties by index) -- with keys = ranks of a permutation this IS pc[permutation[:n]]; with fewer
 *   than n survivors they repeat cyclically (the reference refills by sampling with replacement, then shuffles).
 *   alive may be NULL (all alive).  p <= 16384, n <= 16384.
 */
int pdae_pipeline_norm_affine(int b, int p, int out_stride, int normalise, const float* xyz, const int32_t* nmaps,
                              const float* maps, const float* sigma, const float* noise, float* out,
                              pdae_stream_t stream);
int pdae_pipeline_add_global(int b, int nmax, int stride, int p0, const int32_t* count, const float* u, float* xyz,
                             pdae_stream_t stream);
int pdae_pipeline_add_local(int b, int nmax, int stride, int p0, const int32_t* count, const int32_t* seed,
                            const float* sigma, const float* noise, float* xyz, pdae_stream_t stream);
int pdae_pipeline_density(int b, int p, int stride, const float* xyz, const float* view, const float* gate,
                          const float* r, unsigned char* alive, pdae_stream_t stream);
int pdae_pipeline_subset(int b, int p, int stride, int n, const float* xyz, const unsigned char* alive,
                         const float* keys, float* out, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Chamfer distance.  Replaces chamfer_cuda_forward(xyz1, xyz2)
 *   extensions/chamfer_dist/chamfer.cu:147-171 (kernel :15-145) and
 * chamfer_cuda_backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2)
 *   chamfer.cu:203-229 (kernel :173-201); pybind names chamfer.forward /
 *   chamfer.backward (chamfer_cuda.cpp:36-39).
 * xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n), idx1 (b,n) i32 [nearest in xyz2],
 * dist2 (b,m), idx2 (b,m) i32 [nearest in xyz1]; lowest index wins ties.
 * backward: grad_xyz1 (b,n,3), grad_xyz2 (b,m,3) fully written.
 */
int pdae_chamfer_forward(int b, int n, const float* xyz1, int m,
                         const float* xyz2, float* dist1, float* dist2,
                         int32_t* idx1, int32_t* idx2, pdae_stream_t stream);
int pdae_chamfer_backward(int b, int n, const float* xyz1, int m,
                          const float* xyz2, const int32_t* idx1,
                          const int32_t* idx2, const float* grad_dist1,
                          const float* grad_dist2, float* grad_xyz1,
                          float* grad_xyz2, pdae_stream_t stream);
/* ChamferDistanceL2 (extensions/chamfer_dist/__init__.py:29-44: `torch.mean(dist1) + torch.mean(dist2)`) without the
 * reduction / expand / divide launches around the two kernels: mean_sum2 writes mean(a) + mean(b) into out[0] (two
 * small launches, fixed summation order);
 * chamfer_backward_mean is chamfer.backward for grad_dist1 = grad_loss[0] / (b n), grad_dist2 = grad_loss[0] / (b m)
 * (what autograd's mean backward would materialise), grad_loss a DEVICE scalar. */
int pdae_mean_sum2(long long na, const float* a, long long nb, const float* b, float* workspace /* 256 floats */,
                   float* out, pdae_stream_t stream);
int pdae_chamfer_backward_mean(int b, int n, const float* xyz1, int m, const float* xyz2, const int32_t* idx1,
                               const int32_t* idx2, const float* grad_loss, float* grad_xyz1, float* grad_xyz2,
                               pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Approximate earth mover's distance.  Replaces ApproxMatchForward /
 * MatchCostForward / MatchCostBackward, extensions/emd/cuda/emd_kernel.cu
 * :168-194, :256-280, :369-398 (kernels :25-158, :200-243, :286-355); pybind
 * names emd_cuda.approxmatch_forward / matchcost_forward / matchcost_backward
 * (emd.cpp:23-27).
 * xyz1 (b,n,3), xyz2 (b,m,3); match (b,m,n) f32; cost (b) f32;
 * temp is scratch of b*2*(n+m) floats (the reference allocates it,
 * emd_kernel.cu:183).
 */
int pdae_emd_approxmatch(int b, int n, int m, const float* xyz1,
                         const float* xyz2, float* match, float* temp,
                         pdae_stream_t stream);
int pdae_emd_matchcost(int b, int n, int m, const float* xyz1,
                       const float* xyz2, const float* match, float* cost,
                       pdae_stream_t stream);
int pdae_emd_matchcost_grad(int b, int n, int m, const float* grad_cost,
                            const float* xyz1, const float* xyz2,
                            const float* match, float* grad1, float* grad2,
                            pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * DGCNN encoder (csrc/dgcnn.hip).  Replaces, for Point_CAE_DGCNN_FCOnly (models/PointCAE_DGCNN.py:146-231), the
 * reference's feature-space kNN + edge features (models/dgcnn_util.py:7-34: matmul, topk, advanced-index gather,
 * cat of (x_j - x_i, x_i) as a (B, 2C, N, 20) tensor) and the Conv2d -> BatchNorm2d -> LeakyReLU(0.2) -> max over the
 * 20 neighbours of every EdgeConv (:99-131), and conv5's BatchNorm1d -> LeakyReLU -> max over the points (:132-136).
 * Activations are rows (points) x channels; b clouds of n points; R = b n.
 *   rows_sqnorm:  xx[r] = |x_r|^2 (C % 4 == 0).   rows_pad: out (R, cp) = x (R, c) with zero columns appended.
 *   edge_weight_stack: ws (2 co, kp) = [W1; W2 - W1], zero-padded from cin to kp columns, from the Conv2d weight
 *                 w (co, 2 cin) = [W1 | W2] over cat(x_j - x_i, x_i); edge_weight_unstack: its transpose,
 *                 dw (co, 2 cin) = [dWs_top - dWs_bottom | dWs_bottom].
 *   gram_topk:    idx (b, n, k) int32, ids within the cloud: the k largest of pd[i][j] = (-xx_j + 2 g_ij) - xx_i
 *                 (dgcnn_util.knn's expression -- xx (B, 1, N) broadcasts over the columns first -- each operation
 *                 rounded), best first, the lower id on equal values;
 *                 gram (b, n, n) = X_b X_b^T from pdae_rows_gemm_batched.  k <= 64.
 *   xyz_topk:     the same idx for the FIRST EdgeConv, whose features are the points themselves: x4 (b n, 4) = xyz rows
 *                 zero-padded to 4 columns (rows_pad), g_ij = fma(z z', fma(y y', x x')) computed in the kernel -- no
 *                 (b, n, n) Gram matrix; n <= 6144 (a block keeps its cloud in LDS).  pd_out (nullable, b n n): the -pd values the selection saw (tests).
 *   knn_reverse:  the reverse graph: rev_start (b, n+1), rev_src (b, n k): the points that list point s as a
 *                 neighbour are rev_src[b][rev_start[b][s] .. rev_start[b][s+1]), ascending.  n <= 4096.
 *   edge_gather_stats: pq (R, 2 co) = [p | q], p = W1 x, q = (W2 - W1) x, so that the conv output of edge (r, j) is
 *                 e = p[idx[r][j]] + q[r].  ONE pass over the gathered rows: esel (R, co) the winning e of every
 *                 (point, channel) -- max over j where gamma > 0, min where gamma < 0 (y = lrelu(bn(e)) is monotone
 *                 in e with the sign of gamma), the first edge where gamma = 0 (torch.max's first occurrence) --
 *                 sel (R, co) uint16 the winner's id within the cloud, psum (R, co) = sum_j p[idx[r][j]], and
 *                 sums (2 co doubles) = sum e, sum e^2 over all R k edges (part: pdae_edge_parts() x 2 co doubles of
 *                 scratch; added in block order).  co a power of two in 16..1024, n <= 65535.
 *                 -> pdae_bn_finalize(co, R k, sums, ...) gives scale / shift / mean / invstd and the running estimates.
 *   bn_lrelu_rows: out[r][c] = lrelu(e[r][c] scale[c] + shift[c]), slope 0.2; out2 (nullable, row stride ld2): a
 *                 second copy into a wider row-major tensor (the concatenated features conv5 reads).
 *   bn_lrelu_backward_reduce: g = (d1 + d2) * (y > 0 ? 1 : 0.2), y = e scale + shift (d1 contiguous, d2 with row
 *                 stride ld2, either nullable); sums (2 C doubles) = sum g, sum g xhat, xhat = (e - mean) invstd;
 *                 dbeta = sum g, dgamma = sum g xhat as floats (nullable).  part: pdae_edge_parts() x 2 C doubles.
 *   edge_backward: dpq (R, 2 co) = [dp | dq], the gradient of pq under training-mode BatchNorm over the R k edges:
 *                 d e[r][j] = scale (dy[r][j] - c1 - xhat[r][j] c2), dy = g at the winners and 0 elsewhere,
 *                 c1 = sums[0] / (R k), c2 = sums[1] / (R k); dq[r] = sum_j d e[r][j] (closed form from psum),
 *                 dp[s] = sum over the edges arriving at s (a gather over the reverse graph: no atomics).
 *   cloud_pool_stats: y (R, C) -> ysel (b, C) the winning row value per cloud and channel (max / min / first by
 *                 the sign of gamma), arow (b, C) int32 its row within the cloud, sums (2 C doubles) = sum y, sum y^2.
 *                 A cloud's rows are split over rs = pdae_cloud_pool_splits(b, n) blocks; scratch: pv, pr (b, rs, C)
 *                 the ranges' winners, part b rs x 2 C doubles.
 *   cloud_pool_backward: dy[r][c] = scale ((r == arow[b][c] ? g[b][c] : 0) - c1 - xhat[r][c] c2), sums from
 *                 bn_lrelu_backward_reduce over the (b, C) winners, c1 / c2 = sums / R.
 */
int pdae_rows_sqnorm(int R, int C, const float* x, float* xx, pdae_stream_t stream);
int pdae_rows_pad(long long R, int c, int cp, const float* x, float* out, pdae_stream_t stream);
int pdae_edge_weight_stack(int co, int cin, int kp, const float* w, float* ws, pdae_stream_t stream);
int pdae_edge_weight_unstack(int co, int cin, int kp, const float* dws, float* dw, pdae_stream_t stream);
/* ..._multi: the same for n <= 8 layers in ONE launch (HOST arrays of sizes and device pointers): the encoder stacks its
 * four EdgeConv weights before the first layer and unstacks their gradients after the last one. */
int pdae_edge_weight_stack_multi(int n, const int* co, const int* cin, const int* kp, const float* const* w,
                                 float* const* ws, pdae_stream_t stream);
int pdae_edge_weight_unstack_multi(int n, const int* co, const int* cin, const int* kp, const float* const* dws,
                                   float* const* dw, pdae_stream_t stream);
int pdae_gram_topk(int b, int n, int k, const float* gram, const float* xx, int32_t* idx, pdae_stream_t stream);
int pdae_xyz_topk(int b, int n, int k, const float* x4, const float* xx, int32_t* idx, float* pd_out /*nullable*/,
                  pdae_stream_t stream);
int pdae_knn_reverse(int b, int n, int k, const int32_t* idx, int32_t* rev_start, int32_t* rev_src,
                     pdae_stream_t stream);
int pdae_edge_parts(void);
int pdae_edge_gather_stats(int b, int n, int k, int co, const float* pq, const int32_t* idx, const float* gamma,
                           float* esel, unsigned short* sel, float* psum, double* part, double* sums,
                           pdae_stream_t stream);
int pdae_bn_lrelu_rows(long long R, int C, const float* e, const float* scale, const float* shift, float* out,
                       float* out2 /*nullable*/, int ld2, pdae_stream_t stream);
int pdae_bn_lrelu_backward_reduce(long long R, int C, const float* d1 /*nullable*/, const float* d2 /*nullable*/,
                                  int ld2, const float* e, const float* scale, const float* shift, const float* mean,
                                  const float* invstd, float* g, double* part, double* sums,
                                  float* dgamma /*nullable*/, float* dbeta /*nullable*/, pdae_stream_t stream);
int pdae_edge_backward(int b, int n, int k, int co, const float* g, const float* pq, const unsigned short* sel,
                       const float* psum, const int32_t* rev_start, const int32_t* rev_src, const float* scale,
                       const float* mean, const float* invstd, const double* sums, float* dpq, pdae_stream_t stream);
int pdae_cloud_pool_splits(int b, int n);
int pdae_cloud_pool_stats(int b, int n, int C, const float* y, const float* gamma, float* ysel, int32_t* arow,
                          float* pv, int32_t* pr, double* part, double* sums, pdae_stream_t stream);
int pdae_cloud_pool_backward(int b, int n, int C, const float* y, const float* g, const int32_t* arow,
                             const float* scale, const float* mean, const float* invstd, const double* sums,
                             float* dy, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense layers.  The reference runs nn.Linear / 1x1 nn.Conv1d through
 * cuBLAS / cuDNN in fp32 (patch embedder models/PointCAE_transformer.py:24-51,
 * qkv / proj :113-137, fc1 / fc2 :94-110, pos_embed :329-333, increase_dim
 * :653-658).  Here: fp32-input MFMA (exact fp32 accumulation), row-major
 * activations (rows, channels), weights in torch's (out, in) layout.
 *   forward:          Y[M,N]  = act(X[M,K] . W[N,K]^T + bias[N]); act 0 none,
 *                     1 ReLU, 2 GELU(erf); bias nullable; K % 4 == 0.
 *   backward_data:    dX[M,K] = dY[M,N] . W[N,K], given Wt = W^T as [K,N].
 *   backward_weight:  dW[N,K] = dY^T . X, dbias[N] = column sums of dY
 *                     (nullable); both are overwritten (zero-filled, then
 *                     accumulated with fp32 atomics over M-splits).
 */
int pdae_linear_forward(int M, int N, int K, const float* X, const float* W,
                        const float* bias /*nullable*/, int act, float* Y,
                        pdae_stream_t stream);
int pdae_linear_backward_data(int M, int N, int K, const float* dY,
                              const float* Wt, float* dX, pdae_stream_t stream);
int pdae_linear_backward_weight(int M, int N, int K, const float* dY,
                                const float* X, float* dW,
                                float* dbias /*nullable*/, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense layers of the Transformer blocks (csrc/rows_gemm.hip): Attention.qkv /
 * proj (models/PointCAE_transformer.py:113-137), Mlp.fc1 / act / fc2 (:94-110),
 * pos_embed (:329-333), increase_dim (:653-658), and their backward.  Small-M
 * GEMMs (M = B * T tokens): a family of tile shapes, a per-shape plan, optional
 * split-K into slabs, no atomics (results are bit-identical run to run).
 *
 *   rows_gemm:  Y[M,N] = epi(X[M,K] . op(W))
 *       w_kn = 0: W is [N,K] (torch's (out,in) layout: a Linear's forward)
 *       w_kn = 1: W is [K,N] (the same (out,in) weight as the data-gradient
 *                 operand: dX[M,in] = dY[M,out] . W[out,in]; no transposed copy)
 *       epi  = 0: Y = acc (+ bias, nullable)
 *              1: Y = ReLU(acc + bias)                              (w_kn = 0)
 *              2: z = acc + bias, Y = GELU(z), Z = GELU'(z)   fc1 + nn.GELU in
 *                 one pass; Z is the factor the backward needs     (w_kn = 0)
 *              3: Y = acc * Z                   backward of 2: X = the gradient
 *                 of fc2's output, W = fc2's weight, Z from the forward: Y is
 *                 the gradient of fc1's output                     (w_kn = 1)
 *              4: Y = Z > 0 ? acc : 0      backward through a ReLU whose output
 *                 Z the forward kept (epi 1, or fold_input): the data gradient
 *                 arrives already masked, relu'(0) = 0 as ATen  (either layout)
 *       cfg    : tile shape 0..7, or -1 = planned per shape
 *       splits : 1, or S in 2..8 = split the reduction: Y is then S slabs
 *                [S][M][N] of partial products (epi 0, no bias) which the
 *                consumer adds up (residual_layernorm_forward / layernorm_backward
 *                take a slab count); -1 = 1.
 *       stream_blocks : 0, or P (a multiple of 8) = stream-K: the (tile, k-tile)
 *                units are dealt in equal contiguous ranges to P blocks; the q-th
 *                piece of a tile goes to slab q of Y, unused slabs of a tile are
 *                zero-filled (`splits` = number of slabs, as planned).
 *   rows_gemm_plan: the (cfg, splits, stream_blocks) rows_gemm would pick;
 *       may_split = 0 keeps splits = 1 and stream_blocks = 0; 1 allows up to 4 slabs (what the LayerNorm
 *       kernels add on their way); 2..8 up to that many (a caller with its own consumer: slab_sum_epi).
 *   slab_sum_epi: Y (M,N) = epi(sum_s slabs[s] + bias) in slab order, epi 0 store | 1 ReLU | 4 Z > 0 ? . : 0 --
 *       the Linear layers of the coarse heads (models/PointCAE_DGCNN.py:160-166 recfc, PointCAE_pointnetv2.py:151-156
 *       folding1, PointCAE_transformer.py coarse_pred) run on 32-128 rows: their reductions are split 8 ways.
 *   rows_wgrad: weight gradients of a GROUP of nprob <= 8 Linear layers that
 *       share M in one launch: dW_p[N_p,K_p] = dY_p[M,N_p]^T . X_p[M,K_p],
 *       db_p[N_p] = column sums of dY_p (db or db[p] nullable).  Pointer arrays
 *       are HOST arrays of device pointers.  (tile, 32-row chunk) work units are
 *       dealt in equal contiguous ranges to one residency of the chip; the
 *       partial tiles go to `workspace` and are added in block order by a second
 *       launch; rows_wgrad_workspace returns the workspace size in floats.
 */
int pdae_rows_gemm(int M, int N, int K, const float* X, const float* W, int w_kn,
                   const float* bias /*nullable*/, int epi, float* Z /*epi 2,3,4*/,
                   float* Y, int cfg, int splits, int stream_blocks,
                   pdae_stream_t stream);
/* The masked patches' share of the embedder backward by algebra (point_dae_amd/patch_embed.py; Encoder,
 * models/PointCAE_transformer.py:37-51, behind MaskTransformer.forward :424-469 which drops the masked tokens).
 * Rows of a group whose token is dropped carry no activation gradient into BatchNorm-2's backward, so their
 * conv-output gradient is the correction alone, dh = u + v * h (u, v per channel).  Entries:
 *   bnrelu_backward_listed: S (the sums = dbeta, dgamma) from the listed groups' compact dA, dA overwritten
 *       in place with the listed groups' conv-output gradient, gsum their row sums (row = position in the
 *       list, or the group id when gsum_by_group), uv[2][C].
 *   masked_group_sums: dgb[groups[cg]] = v * hs[cg] + 32 * xe[cg]: the row sums of dh over a masked group
 *       (hs = the group's summed conv output without its bias term, xe = u + v * gb).
 *   group_sum_listed: out[cg] = sum of the 32 rows of X's group groups[cg].
 *   linear_backward_weight_listed: dW[N,K] = sum_m dY[rowA(m)]^T X[rowB(m)], whole 32-row groups gathered on
 *       either operand (lists nullable = compact operand); dY = X with one list = a Gram matrix.
 *   group_gemm_scatter: Y[c_groups[m/32]*32 + m%32] = X[a_groups[m/32]*32 + m%32] . W[N,K]^T + gbias[m/32]
 *       (a_groups, gbias nullable); rows of Y outside the listed groups are not touched. */
int pdae_bnrelu_backward_listed(int G, int C, float* dA, const float* X, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* gamma, float* S,
                                float* gsum /*nullable*/, int gsum_by_group, float* uv /*nullable*/,
                                int n_listed, const int32_t* groups, pdae_stream_t stream);
int pdae_masked_group_sums(int n_listed, int C, const float* hs, const float* xe, const float* v,
                           const int32_t* groups, float* dgb, pdae_stream_t stream);
int pdae_group_sum_listed(int n_listed, int C, const float* X, const int32_t* groups, float* out,
                          pdae_stream_t stream);
int pdae_linear_backward_weight_listed(int M, int N, int K, const float* dY, const int32_t* a_groups,
                                       const float* X, const int32_t* b_groups, float* dW,
                                       float* dbias /*nullable*/, pdae_stream_t stream);
int pdae_group_gemm_scatter(int M, int N, int K, const float* X, const int32_t* a_groups, const float* W,
                            const float* gbias, float* Y, int ldy, const int32_t* c_groups,
                            pdae_stream_t stream);

/* Set-abstraction levels of the PointNet++ encoder (Point_CAE_PointNetv2;
 * extensions/pointnet2/pointnet2_modules.py PointnetSAModule: SharedMLP of
 * Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU layers, then F.max_pool2d over nsample)
 * on rows = (cloud, centre, sample):
 *   conv_stats: Y[M,N] = act(X[M,K]) . W[N,K]^T and this layer's batch statistics
 *       (stats [8][2][N] partial sums / sums of squares for bn_finalize); act = the
 *       previous layer's BatchNorm + ReLU, relu(x*scale[k] + shift[k]), applied while
 *       X is staged (scale = shift = NULL: X as is).  The normalised activations are
 *       never stored.  Backward of a layer: bnrelu_backward, then
 *       bnrelu_linear_backward_weight / rows_gemm (w_kn) as in the patch embedder.
 *   bnrelu_group_max: out[g][c] = max_j relu(y[g*ns+j][c]*scale[c] + shift[c]),
 *       arg[g][c] = the first j attaining it (ns <= 256).
 *   group_max_scatter_n: dense[g*ns+j][c] = (j == arg[g][c]) ? grad[g][c] : 0. */
int pdae_conv_stats(int M, int N, int K, const float* X, const float* scale /*nullable*/,
                    const float* shift /*nullable*/, const float* W, float* Y, float* stats,
                    pdae_stream_t stream);
int pdae_bnrelu_group_max(long long G, int ns, int C, const float* y, const float* scale,
                          const float* shift, float* out, unsigned char* arg, pdae_stream_t stream);
int pdae_group_max_scatter_n(long long G, int ns, int C, const float* grad, const unsigned char* arg,
                             float* dense, pdae_stream_t stream);
/* pool_bn_backward: ReLU + BatchNorm backward of a level's LAST layer straight through the max-pool:
 * grad[G][C] = gradient of the pooled output `out`, arg from bnrelu_group_max, y the raw conv output;
 * -> dy[G*ns][C] = gradient of y, S[2][C] = (dbeta, dgamma).  The gradient of relu(bn(y)) is non-zero
 * only at the arg-max rows, so the sums gather one y element per (group, channel) and the dense
 * scatter + two sweeps over it (group_max_scatter_n, bnrelu_backward) are replaced by one read of y
 * and one write of dy.  Fixed summation order (per-block partials in `workspace`, C/4 must divide 256;
 * pool_bn_backward_workspace gives its size in floats). */
long long pdae_pool_bn_backward_workspace(long long G, int C);
int pdae_pool_bn_backward(long long G, int ns, int C, const float* grad, const unsigned char* arg,
                          const float* out, const float* y, const float* mean, const float* invstd,
                          const float* gamma, float* S, float* workspace, float* dy, pdae_stream_t stream);

/* First layer of the FoldingNet stage of Point_CAE_PointNetv2 (csrc/folding.hip;
 * models/PointCAE_pointnetv2.py:157-167: folding2[0] over [grid(2) | coarse point(3) |
 * global feature(1024)] for `cells` grid cells x `coarse` points x `clouds`).  The
 * conv is linear in the three column blocks: a[clouds,C] (feature block + bias),
 * p[clouds*coarse,C] (coarse-point block), gd[cells,C] (grid block) come from three
 * small GEMMs and
 *   fold_input:      h[((b*coarse + c)*cells + g), :] = ReLU((a[b] + p[b,c]) + gd[g])
 *   fold_input_grad: dpre = the gradient of the pre-activation, already masked by the
 *       ReLU (rows_gemm epi 4 of the next layer's data gradient):
 *       dp[b,c] = sum_g dpre,  dgd_part[blk][g] = sum over block blk's 64 (b,c) pairs --
 *       the caller adds the fold_input_grad_parts(clouds, coarse) partial sets in
 *       order (no atomics) and reduces dp over c for da.  C <= 1024. */
/*   fold_out_backward: the stage's last layer (C -> 3 outputs zero-padded to 4, weight W[4][C]) backwards in one
 *       pass over the kept middle activation h2: d2[r] = (h2[r] > 0) ? dy[r] . W : 0 and the per-block partials
 *       part[blk][4][C] of dW = dy^T h2 (fold_out_backward_parts(rows) sets; the caller adds them in order). */
int pdae_fold_input(int clouds, int coarse, int cells, int C, const float* a, const float* p,
                    const float* gd, float* h, pdae_stream_t stream);
/*   fold_input_rows: the first layer with a per-ROW term instead of the per-cloud / per-cell ones (the published
 *       variant's second folding stage, models/PointCAE_transformer.py:1050-1059, where the first fold's points are the
 *       extra input): h[r] = ReLU(row[r] + p[r / cells]); its backward is fold_input_grad's dp (clouds = 1). */
int pdae_fold_input_rows(long long pairs, int cells, int C, const float* row, const float* p, float* h,
                         pdae_stream_t stream);
int pdae_fold_input_grad_parts(int clouds, int coarse);
int pdae_fold_input_grad(int clouds, int coarse, int cells, int C, const float* dpre, float* dp,
                         float* dgd_part, pdae_stream_t stream);
int pdae_fold_out_backward_parts(long long rows);
int pdae_fold_out_backward(long long rows, int C, const float* dy /*[rows][4]*/, const float* h2, const float* W,
                           float* d2, float* part, pdae_stream_t stream);

/* Batched Y_b[M,N] = X_b[M,K] . W_b[N,K]^T, b < batch, element strides between the
 * problems (W_b = X_b: the Gram matrices behind DGCNN's feature-space kNN,
 * models/dgcnn_util.py:7-12).                                                  */
int pdae_rows_gemm_batched(int batch, int M, int N, int K, const float* X,
                           long long strideX, const float* W, long long strideW,
                           float* Y, long long strideY, pdae_stream_t stream);
int pdae_rows_gemm_plan(int M, int N, int K, int w_kn, int may_split, int* cfg,
                        int* splits, int* stream_blocks);
int pdae_slab_sum_epi(int S, int M, int N, const float* slabs /*[S][M][N]*/, const float* bias /*nullable*/, int epi,
                      const float* Z /*epi 4*/, float* Y, pdae_stream_t stream);
/* GEMM arithmetic of the row-GEMM family (rows_gemm, rows_wgrad*; the Linear layers the reference runs through
 * cuBLAS sgemm: models/PointCAE_transformer.py:94-158).  Both are fp32 in, fp32 out, fp32 accumulation:
 *   PDAE_GEMM_BF16X3  (default) every operand element is split EXACTLY into three bf16 terms, x = h + m + l, and a
 *       product is the sum of the six largest of the nine exact bf16 x bf16 partial products (hh, hm, mh, hl, lh, mm;
 *       the dropped ones are below 2^-25 of the product) accumulated in fp32 on v_mfma_f32_32x32x16_bf16, the hh
 *       terms and the small terms in separate accumulators.  Error against fp64 at or below the fp32-input MFMA
 *       kernels' on every shape of the models (tests/test_gpu_rows3.py), 2.67x their matrix-pipe ceiling.  Reductions
 *       with K % 32 != 0 run on the fp32-input kernels in either mode.
 *   PDAE_GEMM_F32MFMA v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain (the reference's arithmetic class bit for bit
 *       up to summation order).  Also selected by the environment variable PDAE_GEMM=f32mfma at first use.
 * Set before planning: pdae_rows_gemm_plan / the workspace queries answer for the arithmetic in force. */
enum { PDAE_GEMM_F32MFMA = 0, PDAE_GEMM_BF16X3 = 1 };
int pdae_set_gemm_arith(int arith);
int pdae_gemm_arith(void);
int pdae_rows_wgrad_workspace(int M, int nprob, const int* Ns, const int* Ks,
                              long long* floats);
int pdae_rows_wgrad(int M, int nprob, const float* const* dY,
                    const float* const* X, float* const* dW,
                    float* const* db /*nullable*/, const int* Ns, const int* Ks,
                    float* workspace, pdae_stream_t stream);
/* The same with a row count PER layer: the weight gradients of layers that do not share their rows -- every Linear of
 * every Transformer block of a stack -- in ONE launch (nprob <= 48): work = (128 x 128 output tile, 32-row chunk) units
 * of all layers in one list, dealt in equal contiguous ranges to one residency of the chip, partial tiles added in
 * block order by the reduction launch that follows.  A step's backward thereby issues its blocks' weight gradients as
 * two large launches (decoder stack, encoder stack) instead of one small launch per block.  Output tiles: 128 x 128 on the
 * default exact-split arithmetic (pdae_set_gemm_arith below; two accumulator sets per tile); on the fp32-input kernels
 * 128 x 384 when every K of the group is a multiple of 384 (8 waves per block), 128 x 128 otherwise.  On the exact-split
 * arithmetic a group whose layers share their row count and whose tiles fill the chip more than once runs as whole
 * rounds of ONE TILE PER BLOCK first (consecutive tiles together on an XCD, stored by their blocks: no partials) and the
 * remaining tiles as above.  The workspace queries answer for the arithmetic in force. */
int pdae_rows_wgrad_multi_workspace(int nprob, const int* Ms, const int* Ns, const int* Ks, long long* floats);
/* One weight gradient dW[N,K] = sum_m dY[rowA(m)]^T x(X[rowB(m)]) with the patch embedder's operand forms (Encoder,
 * models/PointCAE_transformer.py:37-51; its conv layers' weight gradients, which autograd computes with cuDNN/cuBLAS in
 * the reference): whole 32-row groups gathered on either operand (rowA(m) = a_groups[m/32]*32 + m%32; lists nullable =
 * compact operand; M % 32 == 0 with a list) and x = relu(X * scale[k] + shift[k]) when scale / shift are given
 * (BatchNorm + ReLU recomputed while X is staged).  db nullable: column sums of dY.  On the grouped weight-gradient
 * kernel: partial tiles in `workspace` (pdae_rows_wgrad_workspace(M, 1, &N, &K) floats), ordered reduction -- no
 * atomics, no memset, bit-identical run to run. */
int pdae_rows_wgrad_listed(int M, int N, int K, const float* dY, const int32_t* a_groups /*nullable*/, const float* X,
                           const int32_t* b_groups /*nullable*/, const float* scale /*nullable*/,
                           const float* shift /*nullable*/, float* dW, float* db /*nullable*/, float* workspace,
                           pdae_stream_t stream);
int pdae_rows_wgrad_multi(int nprob, const int* Ms, const float* const* dY, const float* const* X, float* const* dW,
                          float* const* db, const int* Ns, const int* Ks, float* workspace, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused layers of the patch embedder, Encoder.forward
 * (models/PointCAE_transformer.py:37-51).  Rows are points, 32 consecutive rows
 * form one group (patch); M % 32 == 0.  The reference runs each of these as a
 * cuDNN 1x1 conv followed by separate BatchNorm / ReLU / max / concat passes
 * over (B*G*32, C) tensors.
 *
 *   embed_conv_store_groupmax   (first_conv[3] + the max of :47)
 *       Y[M,N] = X[M,K].W[N,K]^T + bias;  gmax[M/32,N] = max over each group's
 *       32 rows of Y, garg[M/32,N] (u8) = first row inside the group attaining it.
 *   embed_conv_groupbias_stats  (second_conv[0] on concat([global, local]), :48-49,
 *       with the weight split into its global / local halves)
 *       Y[M,N] = X[M,K].W[N,K]^T + gbias[M/32,N]   (gbias = global half + bias,
 *       one row per group);  stats[8][2][N] receives 8 partial (sum, sum of
 *       squares) of Y's columns -- BatchNorm batch statistics (:32) without a
 *       pass over Y.  stats is zero-filled by the call.
 *   embed_bnrelu_conv_groupmax  (second_conv[1..3] + the max of :50)
 *       gmax[M/32,N] = max over groups of (relu(X*scale + shift).W^T + bias),
 *       garg = argmax row; the (M,N) product itself is never written.
 *   bnrelu_linear_backward_weight
 *       dW[N,K] = dY[M,N]^T . relu(X[M,K]*scale + shift)  (activation recomputed
 *       while it is staged); dW is overwritten.
 */
int pdae_embed_conv_store_groupmax(int M, int N, int K, const float* X,
                                   const float* W, const float* bias, float* Y,
                                   float* gmax, unsigned char* garg,
                                   pdae_stream_t stream);
int pdae_embed_conv_groupbias_stats(int M, int N, int K, const float* X,
                                    const float* W, const float* gbias,
                                    float* Y, float* stats,
                                    pdae_stream_t stream);
int pdae_embed_bnrelu_conv_groupmax(int M, int N, int K, const float* X,
                                    const float* scale, const float* shift,
                                    const float* W, const float* bias,
                                    float* gmax, unsigned char* garg,
                                    const int32_t* groups /*nullable*/,
                                    pdae_stream_t stream);
int pdae_bnrelu_linear_backward_weight(int M, int N, int K, const float* dY,
                                       const float* X, const float* scale,
                                       const float* shift, float* dW,
                                       float* dbias /*nullable: column sums of dY*/,
                                       const int32_t* groups /*nullable*/,
                                       pdae_stream_t stream);
/* `groups` (embed_bnrelu_conv_groupmax, bnrelu_linear_backward_weight,
 * bnrelu_backward): a list of group (patch) ids.  The last conv of the embedder
 * comes after the last BatchNorm, and the tokens of masked patches are thrown
 * away by MaskTransformer.forward (:449), so only the VISIBLE patches need that
 * GEMM: with a list, the M = 32*len(groups) rows of the product are gathered
 * from X group-wise (row m <- X[groups[m/32]*32 + m%32]) and the outputs are
 * compact, in list order.  Same numbers as computing all patches and selecting. */
/*   embed_bnrelu_conv_store_groupmax  (first_conv[1..3] + the max of :47)
 *       Y[M,N] = relu(X*scale + shift).W^T + bias, plus its group max / argmax. */
int pdae_embed_bnrelu_conv_store_groupmax(int M, int N, int K, const float* X,
                                          const float* scale, const float* shift,
                                          const float* W, const float* bias,
                                          float* Y, float* gmax,
                                          unsigned char* garg,
                                          pdae_stream_t stream);

/*   embed_conv1_stats   first_conv[0] (:24, Conv1d(3, C, 1)) on rows (R,3):
 *       y[R,C] = x.W^T + bias, evaluated as ((x0*w0 + x1*w1) + x2*w2) + b, and
 *       stats[0][c] += sum y, stats[1][c] += sum y*y in fp64 (caller zeroes
 *       stats before the first call; C/4 must divide 256).
 *   bn_finalize         training-mode nn.BatchNorm1d bookkeeping (:25,:31) in one
 *       launch: batch mean and biased variance from `stats64` ([2][C] fp64 sums)
 *       or, when that is null, from `P` fp32 partial sets `partials` [P][2][C]
 *       (what embed_conv_groupbias_stats leaves); running_mean / running_var
 *       (unbiased, `momentum`; nullable) and num_batches_tracked (nullable)
 *       are updated in place; scale = gamma*invstd, shift = beta - mean*scale,
 *       mean and invstd (C each) are written.                               */
int pdae_embed_conv1_stats(int R, int C, const float* x, const float* W,
                           const float* bias /*nullable*/, float* y,
                           double* stats, pdae_stream_t stream);
/* embed_conv1_backward_weight: the first conv's weight gradient (K = 3), dW[c][k] = sum_r d[r][c] x[r][k], as
 * per-block partials part[blk][3][C] in one pass over d (embed_conv1_backward_weight_parts(R) sets; the caller
 * adds them in block order and transposes to the (C, 3, 1) weight layout). */
int pdae_embed_conv1_backward_weight_parts(int R);
int pdae_embed_conv1_backward_weight(int R, int C, const float* d, const float* x, float* part,
                                     pdae_stream_t stream);
int pdae_bn_finalize(int C, long long rows, const double* stats64 /*nullable*/,
                     const float* partials /*nullable*/, int P,
                     const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean /*nullable*/,
                     float* running_var /*nullable*/,
                     long long* num_batches_tracked /*nullable*/,
                     float* scale, float* shift, float* mean, float* invstd,
                     pdae_stream_t stream);

/* Memory-bound backward passes of the embedder (autograd of torch.max over the
 * 32 points of a group, nn.ReLU and training-mode nn.BatchNorm1d, :26,:32,:47,:50),
 * each a single fused sweep.  G groups of 32 rows, C channels (C % 4 == 0).
 *   group_max_scatter: out[g*32+r][c] = (r == arg[g][c]) ? grad[g][c] : 0
 *   group_scatter_add: dst[g*32+arg[g][c]][c] += grad[g][c]
 *   bnrelu_backward:   t = dA * (X*scale+shift > 0); S[0][c] = sum t (= d beta),
 *                      S[1][c] = sum t*xhat (= d gamma), xhat = (X-mean)*invstd;
 *                      dA <- gamma*invstd*(t - S0/R - xhat*S1/R) in place;
 *                      gsum[g][c] (nullable) = sum of the new dA over the group.
 */
int pdae_group_max_scatter(int G, int C, const float* grad,
                           const unsigned char* arg, float* out,
                           pdae_stream_t stream);
int pdae_group_scatter_add(int G, int C, const float* grad,
                           const unsigned char* arg, float* dst,
                           pdae_stream_t stream);
int pdae_bnrelu_backward(int G, int C, float* dA, const float* X,
                         const float* scale, const float* shift,
                         const float* mean, const float* invstd,
                         const float* gamma, float* S, float* gsum /*nullable*/,
                         int n_listed, const int32_t* groups /*nullable*/,
                         const int32_t* inv_group /*nullable*/,
                         float* dX /*nullable*/, pdae_stream_t stream);
/*   with groups / inv_group / dX: dA is compact (n_listed*32 rows, the listed
 *   groups; every other group's dA is zero), inv_group[g] = position of g in the
 *   list or -1, and the result for ALL G groups is written to dX.            */
/* The data gradient that flows INTO such a BatchNorm + ReLU, with bnrelu_backward's sums out of the same launch
 * (autograd of nn.Conv1d -> nn.BatchNorm1d -> nn.ReLU chains: models/PointCAE_transformer.py:26-35 Encoder,
 * extensions/pointnet2/pointnet2_modules.py SharedMLP; the reference runs a cuDNN/cuBLAS product, then ATen's
 * threshold_backward and batch_norm_backward sweeps over the (M, N) gradient):
 *   rows_gemm_bnrelu_stats   T[M,N] = (X*scale+shift > 0) ? dY[M,K] . W[K,N] : 0 -- the ReLU mask applied while the
 *       product tile is in registers -- and S[0][n] = sum_m T, S[1][n] = sum_m T xhat, xhat = (X-mean)*invstd.  X rows
 *       through `groups` (row m of the product = row groups[m/32]*32 + m%32 of X; M % 32 == 0) when given.  workspace:
 *       rows_gemm_bnrelu_stats_workspace(M, N) floats (one partial row of sums per 128-row band; added in band order
 *       in fp64: no atomics, bit-identical run to run).  On the fp32-input arithmetic, or without a workspace, the
 *       entry runs the plain product and bnrelu_backward's own reduction sweep.
 *   bnrelu_backward_apply / bnrelu_backward_listed_apply   the second halves of bnrelu_backward / _listed for a caller
 *       that holds S already: same arguments, S is read. */
long long pdae_rows_gemm_bnrelu_stats_workspace(int M, int N);
int pdae_rows_gemm_bnrelu_stats(int M, int N, int K, const float* dY, const float* W, const float* X,
                                const int32_t* groups /*nullable*/, const float* scale, const float* shift,
                                const float* mean, const float* invstd, float* T, float* S, float* workspace,
                                pdae_stream_t stream);
int pdae_bnrelu_backward_apply(int G, int C, float* dA, const float* X, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* gamma, const float* S,
                               float* gsum /*nullable*/, int n_listed, const int32_t* groups /*nullable*/,
                               const int32_t* inv_group /*nullable*/, float* dX /*nullable*/, pdae_stream_t stream);
int pdae_bnrelu_backward_listed_apply(int G, int C, float* dA, const float* X, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, const float* gamma, const float* S,
                                      float* gsum, int gsum_by_group, float* uv, int n_listed, const int32_t* groups,
                                      pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Transformer block, Block.forward (models/PointCAE_transformer.py:155-158 with
 * the per-block position add of :174-177) and Attention.forward (:125-137).
 * Activations are rows (B*T, C); T tokens per sample, H heads of D = 64.
 *
 *   attention_forward   qkv (B*T, 3*H*D) as laid out by the qkv Linear
 *                       ([3][H][D] per row) -> o (B*T, H*D) =
 *                       softmax(q k^T * scale) v per (sample, head), and
 *                       lse (B,H,T) = log-sum-exp of each score row (saved for
 *                       the backward).  T <= 128.
 *   attention_backward  -> dqkv (B*T, 3*H*D), fully written.
 *   add_layernorm_forward  s = x (+ pos, nullable; then xsum receives s);
 *                       y = LayerNorm(s) * gamma + beta; mean/rstd (M) saved.
 *   layernorm_backward  dx = LayerNorm'(dy) (+ dres, nullable: the gradient
 *                       arriving over the skip connection); dgamma/dbeta (C)
 *                       are overwritten (or added to, see `accumulate`).
 *   gelu_forward / gelu_backward   exact (erf) GELU, nn.GELU default (:97,:107).
 *   scale_residual      y = res + keep[row/T] * (a + bias): Linear bias, timm
 *                       DropPath (keep[b] is 0 or 1/keep_prob) and the residual
 *                       add in one pass; bias / keep / res nullable.
 *   colsum              out[c] = sum over rows of X[:, c] (bias gradients).
 */
int pdae_attention_forward(int B, int T, int H, int D, float scale,
                           const float* qkv, float* o, float* lse,
                           pdae_stream_t stream);
int pdae_attention_backward(int B, int T, int H, int D, float scale,
                            const float* qkv, const float* o, const float* lse,
                            const float* d_o, float* dqkv, pdae_stream_t stream);
int pdae_add_layernorm_forward(int M, int C, const float* x,
                               const float* pos /*nullable*/, const float* gamma,
                               const float* beta, float eps,
                               float* xsum /*nullable*/, float* y, float* mean,
                               float* rstd, pdae_stream_t stream);
int pdae_layernorm_backward(int M, int C, const float* dy, int dy_slabs, const float* x,
                            const float* mean, const float* rstd,
                            const float* gamma, const float* dres /*nullable*/,
                            float* dx, float* dgamma, float* dbeta,
                            int accumulate, float* dacc /*nullable*/, int dacc_mode,
                            pdae_stream_t stream);
/*   residual_layernorm_forward / _backward: the previous sub-layer's tail folded in
 *       (Block.forward :155-158: x = x + drop_path(branch(...)) followed by the next
 *       norm):  s = res + keep[row/T] * (a + bias) (+ pos);  y = LayerNorm(s);
 *       xsum receives s.  bias / keep / pos nullable.  Backward:
 *       dx = LayerNorm'(dy) + dres (the gradient w.r.t. s = w.r.t. res and pos);
 *       da = keep[row/T] * dx (written only when keep is given; otherwise da == dx);
 *       dbias = column sums of da; dgamma / dbeta as layernorm_backward.
 *       a_slabs / dy_slabs (1..8): `a` / `dy` arrive as that many split-K slabs
 *       [slabs][M][C] of partial products (pdae_rows_gemm with splits > 1) and are
 *       added up in slab order while they are read; 1 = a plain [M][C] matrix.
 *       dacc / dacc_mode (both backward entries): dacc_mode 1 also writes dx to dacc,
 *       2 adds dx to dacc (the position embedding is re-added before EVERY block of a
 *       stack, :174-177, so its gradient is the sum of the blocks' dx: accumulated
 *       here instead of by one elementwise add per block); 0 = off.                  */
int pdae_residual_layernorm_forward(int M, int C, int T, const float* a, int a_slabs,
                                    const float* bias /*nullable*/,
                                    const float* keep /*nullable*/,
                                    const float* res, const float* pos /*nullable*/,
                                    const float* gamma, const float* beta,
                                    float eps, float* xsum, float* y, float* mean,
                                    float* rstd, pdae_stream_t stream);
int pdae_residual_layernorm_backward(int M, int C, int T, const float* dy, int dy_slabs,
                                     const float* x, const float* mean,
                                     const float* rstd, const float* gamma,
                                     const float* dres /*nullable*/,
                                     const float* keep /*nullable*/, float* dx,
                                     float* da /*nullable unless keep*/,
                                     float* dgamma, float* dbeta, float* dbias,
                                     int accumulate, float* dacc /*nullable*/,
                                     int dacc_mode, pdae_stream_t stream);
int pdae_gelu_forward(long long n, const float* z, float* h,
                      pdae_stream_t stream);
int pdae_gelu_backward(long long n, const float* z, const float* dh, float* dz,
                       pdae_stream_t stream);
int pdae_scale_residual(int M, int C, int T, const float* a,
                        const float* bias /*nullable*/,
                        const float* keep /*nullable*/,
                        const float* res /*nullable*/, float* y,
                        pdae_stream_t stream);
/* Two small fused launches in place of chains of framework elementwise kernels inside the step:
 * drop_path_keep: timm 0.4.5 DropPath (models/PointCAE_transformer.py:146-158 `self.drop_path`) for all `sites`
 *   stochastic-depth sites of a stack: r (sites, B) uniforms -> floor(r + keep[s]) / keep[s] (in place allowed).
 * pos_embed_fc1: the first layer of pos_embed / decoder_pos_embed (:329-333, :636-640: Linear(3,128) -> GELU) on the
 *   centre rows `rows` (int64 indices into xyz (.,3); NULL = all rows in order): h = GELU(z), gp = GELU'(z) for
 *   z = xyz . W1^T + b1, and xp (M,4) = the gathered rows zero-padded (the backward's weight-gradient operand). */
int pdae_drop_path_keep(int sites, int B, const float* r, const float* keep, float* out, pdae_stream_t stream);
/* tail_rows_gather / _scatter: TransformerDecoder.forward returns x[:, -return_token_num:] (models/PointCAE_transformer.py:
 *   225-232), so the last block's row-wise second half runs on the last `tail` tokens of every sample only.
 *   gather:  a_t (B tail, C) = rows T - tail .. T - 1 of every sample of a (B T, C); the same for b -> b_t (b nullable).
 *   scatter: da (B T, C) = da_t on those rows, ZERO elsewhere; the same for db_t -> db (db nullable).  C % 4 == 0. */
int pdae_tail_rows_gather(int B, int T, int tail, int C, const float* a, const float* b /*nullable*/, float* a_t,
                          float* b_t /*nullable*/, pdae_stream_t stream);
int pdae_tail_rows_scatter(int B, int T, int tail, int C, const float* da_t, const float* db_t /*nullable*/, float* da,
                           float* db /*nullable*/, pdae_stream_t stream);
int pdae_pos_embed_fc1(int M, int H, const float* xyz, const int64_t* rows, const float* w1, const float* b1, float* h,
                       float* gp, float* xp, pdae_stream_t stream);
int pdae_colsum(int M, int N, const float* X, float* out, int accumulate,
                pdae_stream_t stream);
/* Single launches in place of groups of framework launches inside the graphed step (csrc/glue.hip; every node of the step's
 * graph costs ~5 us whatever it moves):
 *   partials_sum_t    out (C, K) = sum_p part[p] (K, C) in a fixed order, written transposed: the first conv's weight gradient in the
 *                     parameter's layout (models/PointCAE_transformer.py:22 nn.Conv1d(3, 128, 1): weight (128, 3, 1)) from
 *                     embed_conv1_backward_weight's ordered partials.
 *   multi_copy        n independent float copies src[i] -> dst[i] of counts[i] elements in ONE launch per 128 entries (the gather of
 *                     the gradients autograd produced outside the flat gradient buffer of the data-parallel wrapper -- the
 *                     reference's DistributedDataParallel owns that copy inside its reducer, runner_pretrain.py:84-90).  src, dst,
 *                     counts, cols, src_ld are HOST arrays; cols / src_ld (both or neither; NULL = contiguous): entry i reads a 2-D
 *                     source of cols[i] columns with row stride src_ld[i] (<= 65535), dst is always contiguous.  src[i] NULL: dst[i] is zero-filled
 *                     (the parameters that received no gradient).
 *   assemble_tokens   the decoder's input (:700-703): out (B, G, C) = per sample [the Tv visible tokens | G - Tv copies of
 *                     mask_token]; _grad splits dout (B, G, C) into dvis (B, Tv, C) and dmask (B, G - Tv, C) (the mask token's
 *                     gradient is the column sum of dmask).  C % 4 == 0.
 *   embed_split_conv3_weight   second_conv[0]'s weight w (N, 2 K2) (:29 nn.Conv1d(512, 512, 1) on concat([global, local]), :44-46)
 *                     -> wg = w[:, :K2], wl = w[:, K2:], and wlt = wl^T (K2, N) (nullable).
 *   embed_masked_prep / embed_dw3_assemble   element-wise steps of the masked-groups algebra of conv3's backward
 *                     (point_dae_amd/patch_embed.py _masked_by_algebra): xe (Gm, C3) = u + gb[masked] * v and wv (C3, C2) =
 *                     diag(v) wl, with uv (2, C3) = [u; v];  dw3 (C3, 2 C2) = [dwg | dwl + diag(v) wgram + xterm]
 *                     (v NULL: [dwg | dwl], wgram / xterm unread). */
/*   pad2d             out (R2, C2) = in (R, C; row stride ld >= C: a column slice of a wider matrix) in the top-left corner, zeros elsewhere: the zero-padding of a ragged weight or
 *                     of a 3-column activation to the row GEMMs' multiples of 4 (F.pad: a fill and a copy) in one launch. */
/*   max_plus_mean     out (B, C) = max over t + mean over t of x (B, T, C), arg (B, C) = the first t of the maximum (uint8, T <= 255):
 *                     the published variant's global feature (models/PointCAE_transformer.py:1024 `x_vis.max(dim=1)[0] +
 *                     x_vis.mean(1)`); _grad: dx = g (1 / T + [t == arg]). */
int pdae_max_plus_mean(int B, int T, int C, const float* x, float* out, uint8_t* arg, pdae_stream_t stream);
int pdae_max_plus_mean_grad(int B, int T, int C, const float* g, const uint8_t* arg, float* dx, pdae_stream_t stream);
/*   hcat              out (R, sum cols) = up to four row-major pieces side by side, piece q read through row stride ld[q] >= cols[q]
 *                     (src, cols, ld: HOST arrays; src[q] NULL: cols[q] zero columns): the gradient of a conv weight whose column blocks were used as separate
 *                     operands (models/PointCAE_transformer.py:1040-1059 folding1/2[0] on [token | grid or point];
 *                     models/PointCAE_pointnetv2.py:157-167 folding2[0] on [grid | coarse point | feature]) in one launch. */
int pdae_hcat(int n, int R, const float* const* src, const int* cols, const int* ld, float* out, pdae_stream_t stream);
int pdae_pad2d(int R, int C, int ld, int R2, int C2, const float* in, float* out, pdae_stream_t stream);
int pdae_partials_sum_t(int P, int K, int C, const float* part, float* out, pdae_stream_t stream);
int pdae_multi_copy(int n, const float* const* src, float* const* dst, const long long* counts, const int* cols /*nullable*/,
                    const int* src_ld /*nullable*/, pdae_stream_t stream);
int pdae_assemble_tokens(int B, int G, int Tv, int C, const float* vis, const float* token, float* out, pdae_stream_t stream);
int pdae_assemble_tokens_grad(int B, int G, int Tv, int C, const float* dout, float* dvis, float* dmask, pdae_stream_t stream);
int pdae_embed_split_conv3_weight(int N, int K2, const float* w, float* wg, float* wl, float* wlt /*nullable*/,
                                  pdae_stream_t stream);
int pdae_embed_masked_prep(int Gm, int C3, int C2, const float* uv, const float* gb, const int32_t* masked, const float* wl,
                           float* xe, float* wv, pdae_stream_t stream);
int pdae_embed_dw3_assemble(int C3, int C2, const float* dwg, const float* dwl, const float* v /*nullable*/,
                            const float* wgram /*nullable*/, const float* xterm /*nullable*/, float* dw3, pdae_stream_t stream);
/*   scale_colsum        Y = keep[row/T] * X and out[c] (nullable) = column sums
 *                       of Y: autograd of scale_residual w.r.t. its branch and
 *                       bias in one pass (DropPath backward, timm drop.py).   */
int pdae_scale_colsum(int M, int N, int T, const float* X, const float* keep,
                      float* Y, float* out /*nullable*/, int accumulate,
                      pdae_stream_t stream);
/*   bias_gelu_forward   h = GELU(z + bias)            (fc1 epilogue, :104-106)
 *   bias_gelu_backward  dz = dh * GELU'(z + bias), dbias = column sums of dz
 * `accumulate` != 0 (layernorm_backward, colsum, bias_gelu_backward): the small
 * reduction outputs (dgamma/dbeta, out, dbias) are added to instead of being
 * zero-filled first, so a caller can hand in slices of ONE pre-zeroed arena and
 * save a memset node per call.                                              */
int pdae_bias_gelu_forward(int M, int C, const float* z, const float* bias,
                           float* h, pdae_stream_t stream);
int pdae_bias_gelu_backward(int M, int C, const float* z, const float* bias,
                            const float* dh, float* dz, float* dbias,
                            int accumulate, pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused AdamW over a contiguous fp32 range (torch.optim.AdamW arithmetic:
 * decoupled weight decay, bias correction, eps added to sqrt(v_hat)); the
 * reference builds it over two parameter groups, tools/builder.py:41-101.
 * step is the 1-based update count.  One streaming pass: 16 B read + 12 B
 * written per parameter.
 */
int pdae_adamw_step(long long n, float* param, const float* grad,
                    float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step,
                    pdae_stream_t stream);

/* ------------------------------------------------------------------------
 * Measurement aids: box calibration for bench.py (csrc/calib.hip).  They replace nothing of the reference; the training
 * step never calls them.  MI355X boxes differ by +-4 % on the step and by up to 12 % on matrix loops (the clock a device
 * holds under load), so a bench line carries the rates of two kernels of known work next to its headline.
 *   calib_mfma_bf16  `blocks` work-groups of 4 waves (one per SIMD) each issue iters x 32 v_mfma_f32_32x32x16_bf16 on
 *                    pseudo-random register operands: *flops (host, nullable) = the FLOPs of the launch; clk[b][0..1] =
 *                    shader-clock cycles and 100 MHz ticks wave 0 of block b lived (clock = 100 MHz x cycles / ticks).
 *   calib_copy       dst[0..bytes) = src[0..bytes), 16 B per lane, grid-stride: 2 x bytes of HBM traffic.            */
int pdae_calib_mfma_bf16(int blocks, int iters, float* sink, long long* clk, double* flops, pdae_stream_t stream);
int pdae_calib_copy(long long bytes, const void* src, void* dst, pdae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PDAE_H */
