/*
 * pointflow_hip.h  --  C ABI of libpointflow_hip.so (MI355X / gfx950 native PointFlow hot path).
 *
 * Drop-in boundary for the reference's operator layer.  The reference binds
 * its one native op through a pybind module taking at::Tensor by value
 * (reference pointmvsnet/functions/csrc/main.cpp:3-6, gather_knn.h:7-13); everything else on the path
 * is a chain of ATen calls.  This library exposes the same operators -- and the fused forms the
 * MI355X pipeline uses -- as plain `extern "C"` functions over raw device pointers, sizes and a
 * hipStream_t, with no torch types in any signature.  The Python host side
 * (pointmvsnet_amd/_lib.py) binds them with ctypes; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call returns
 *     without synchronising (the reference launches on the default stream,
 *     gather_knn_kernel.cu:138 -- a latent bug that is not reproduced);
 *   - tensors are dense row-major float32 unless stated; indices are int64 as the reference API
 *     mandates (functions/gather_knn.py:10-24, utils/torch_utils.py:16-22);
 *   - return value: PF_OK, a negative PF_ERR_* for argument errors (nothing was launched), or a
 *     positive hipError_t;
 *   - out-of-range neighbour indices never fault: the access is skipped / clamped and a sticky
 *     device status bit is raised, readable with pf_check_status() (the reference uses a device
 *     assert, gather_knn_kernel.cu:85).
 */
#ifndef POINTFLOW_HIP_H_
#define POINTFLOW_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_OK 0
#define PF_ERR_INVALID_ARG (-1)
#define PF_ERR_UNSUPPORTED (-2)

#define PF_STATUS_BAD_INDEX 1u

#define PF_MAX_VIEWS 8
#define PF_GEMM_TILE 64 /* points per GEMM / stats tile */

/* ---- packed camera block consumed by pf_flow_features_f32 (floats, device) -----------------
 *  [0..8]   inverse of the reference-view intrinsic, row-major      (model.py:168-169)
 *  [9..17]  inverse of the reference-view rotation                   (model.py:57,177)
 *  [18..20] reference-view translation                                (model.py:56)
 *  [21..23] world-point mean, [24..26] std                            (model.py:46-48)
 *  [27 + 21*v .. ] view v: intrinsic K (9, row-major) then extrinsic [R|t] (12, row-major 3x4)
 */
#define PF_CAM_KREF_INV 0
#define PF_CAM_RREF_INV 9
#define PF_CAM_TREF 18
#define PF_CAM_MEAN 21
#define PF_CAM_STD 24
#define PF_CAM_VIEWS 27
#define PF_CAM_VIEW_STRIDE 21
#define PF_CAM_FLOATS(V) (PF_CAM_VIEWS + PF_CAM_VIEW_STRIDE * (V))

const char* pf_version(void);
const char* pf_error_string(int code);
/* Number of compute units / LDS bytes of the current device, and its gcnArchName. */
int pf_device_info(int* cu_count, int* lds_bytes_per_block, char* arch_host, int arch_len);
/* Diagnostics: a one-thread kernel that writes the constant-rate device clock (100 MHz ticks) to *slot when it
 * runs on `stream` -- a timestamp that can sit INSIDE a captured hipGraph, where events cannot (PF_TIMELINE=1). */
int pf_debug_timestamp(long long* slot, void* stream);
/* Synchronises `stream`, returns the sticky status bits in *status_host and clears them. */
int pf_check_status(unsigned* status_host, void* stream);

/* ---- row G : gather_knn ------------------------------------------------------------------
 * Replaces dgcnn_ext.gather_knn_forward / gather_knn_backward
 * (reference functions/csrc/gather_knn_kernel.cu:25-47 and :97-148).
 *   forward : out[b,c,n,j] = feature[b,c,index[b,n,j]]      feature (B,C,N), index (B,N,K), out (B,C,N,K)
 *   backward: grad_in[b,c,index[b,n,j]] += grad_out[b,c,n,j]  (grad_in is zeroed by the call) -- float atomics in
 *             arrival order like the reference's kernel when inv_order / inv_start are NULL; with the inverted index
 *             lists of pf_knn_inverse(index, K, B, N, ...) (further down) every grad_in element is one thread's sum
 *             over the slots that name it, in ascending slot order: no atomics, bit-reproducible.
 * float32 and float64 like the reference's AT_DISPATCH_FLOATING_TYPES (:134). */
int pf_gather_knn_forward_f32(const float* feature, const int64_t* index, float* out,
                              int64_t B, int64_t C, int64_t N, int64_t K, void* stream);
int pf_gather_knn_forward_f64(const double* feature, const int64_t* index, double* out,
                              int64_t B, int64_t C, int64_t N, int64_t K, void* stream);
int pf_gather_knn_backward_f32(const float* grad_out, const int64_t* index, float* grad_in,
                               int64_t B, int64_t C, int64_t N, int64_t K, const uint32_t* inv_order,
                               const uint32_t* inv_start, void* stream);
int pf_gather_knn_backward_f64(const double* grad_out, const int64_t* index, double* grad_in,
                               int64_t B, int64_t C, int64_t N, int64_t K, const uint32_t* inv_order,
                               const uint32_t* inv_start, void* stream);

/* ---- row K : lattice kNN -------------------------------------------------------------------
 * Replaces get_knn_3d (reference utils/torch_utils.py:16-61).  xyz is (B,3,D,H,W) addressed through
 * `strides_host` (5 element strides, host array) so the strided sub-lattice views of
 * model.py:251-252 need no copy.  Candidates are the kernel_size^3 window, zero outside the lattice;
 * d2 = (dx*dx + dy*dy) + dz*dz in float32 without contraction; ranking: smaller d2 first, ties by
 * smaller candidate code (the reference's tie order is unspecified).
 *   idx_out  (B, D*H*W, knn) int64 or NULL: n + offsets, one global clamp to [0, DHW-1] (torch_utils.py:55-59)
 *   code_out (B, D*H*W, knn) uint8 or NULL: the window candidate code of each pick (kernel_size<=5);
 *            at least one of the two must be given
 * Limits: kernel_size odd, <= 7; knn <= min(32, kernel_size^3). */
int pf_knn_lattice_f32(const float* xyz, const int64_t* strides_host, int64_t B, int64_t D, int64_t H,
                       int64_t W, int kernel_size, int knn, int64_t* idx_out, uint8_t* code_out,
                       void* stream);

/* ---- row W : the warp ----------------------------------------------------------------------
 * Replaces FeatureFetcher.forward (reference utils/feature_fetcher.py:13-60): p = R X + t,
 * (x/z, y/z, 1) K^T, bilinear sample at pixel index (u-.5, v-.5), zero padding, legacy
 * align_corners=True.  maps (B,V,C,H,W), pts (B,3,N), K (B,V,3,3),
 * E (B,V,3,4) or NULL (identity), out (B,V,C,N).  V <= PF_MAX_VIEWS.
 * backward: gradient w.r.t. the maps only (the grid is built under no_grad, :29); grad_maps is
 * zeroed by the call. */
int pf_fetch_forward_f32(const float* maps, const float* pts, const float* K, const float* E, float* out,
                         int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t N, void* stream);
int pf_fetch_backward_f32(const float* grad_out, const float* pts, const float* K, const float* E,
                          float* grad_maps, int64_t B, int64_t V, int64_t C, int64_t H, int64_t W,
                          int64_t N, void* stream);

/* ---- rows W+V : fetch + variance over views --------------------------------------------------
 * Fuses FeatureFetcher with E[x^2]-E[x]^2 over V (reference model.py:102-111, :187-190); never
 * materialises (B,V,C,N).  ref_override != 0 reproduces model.py:103-106: view 0 contributes the
 * un-warped reference feature maps[b,0,c, n mod (H*W)].  out (B,C,N). */
int pf_fetch_variance_f32(const float* maps, const float* pts, const float* K, const float* E, float* out,
                          int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t N,
                          int ref_override, void* stream);
/* The coarse stage's use of it (reference model.py:79-111): the points are the frustum of the reference view,
 * generated in the kernel instead of read -- point n = d*H*W + y*W + x is
 *   world = rinv[b] (depths[b,d] * kinv[b] (x+0.5, y+0.5, 1)^T - t[b])
 * (kinv, rinv row-major 3x3: inverse coarse intrinsics and inverse rotation of view 0; t its translation),
 * with ref_override semantics.  world != NULL also receives the points (B, 3, D*H*W), the model's
 * "world_points" output.  out (B, C, D*H*W). */
int pf_frustum_variance_f32(const float* maps, const float* kinv, const float* rinv, const float* t,
                            const float* depths, const float* K, const float* E, float* out, float* world,
                            int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t D, void* stream);

/* The same on CHANNEL-LAST maps (B, V, H, W, C), C % 4 == 0 (pf_nchw_to_nhwc_f32 converts; a convolution may
 * also write that layout directly): 16 lanes per point read every bilinear tap as one contiguous C-float run,
 * the (C, D*H*W) volume is written in 256-byte rows through an LDS transpose.  Bit-identical results. */
int pf_frustum_variance_cl_f32(const float* maps_cl, const float* kinv, const float* rinv, const float* t,
                               const float* depths, const float* K, const float* E, float* out, float* world,
                               int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t D, void* stream);
/* in (P, C, S) -> out (P, S, C): planar to channel-last for P images of S pixels. */
int pf_nchw_to_nhwc_f32(const float* in, float* out, int64_t P, int64_t C, int64_t S, void* stream);

/* ---- bilinear resize (align_corners = False), the F.interpolate of model.py:184 -------------
 * in (P, IH, IW) -> out (P, OH, OW). */
int pf_resize_bilinear_f32(const float* in, float* out, int64_t P, int64_t IH, int64_t IW, int64_t OH,
                           int64_t OW, void* stream);

/* The three pyramid levels of one flow iteration (model.py:180-186) in one launch: in_l (V, c_l, h_l, w_l)
 * -> out_l (V, h, w, c_l) CHANNEL-LAST, bilinear align_corners = False (a level already at (h, w) is
 * transposed only).  c_l % 4 == 0 (else PF_ERR_UNSUPPORTED); c_l == 0 skips a level.  in_scale / in_shift (host
 * arrays of three device pointers, or NULL; an entry may be NULL): rows (V, c_l) of the pending BatchNorm + ReLU of
 * level l -- the tower's raw convolution output is normalised texel by texel BEFORE it is interpolated, i.e. the
 * F.interpolate of relu(bn(x)) without the normalised map ever being written (reference nn/conv.py:62-77 + model.py:184). */
int pf_flow_pyramid_f32(const float* in1, int c1, int h1, int w1, const float* in2, int c2, int h2, int w2,
                        const float* in3, int c3, int h3, int w3, int V, int h, int w, float* out1, float* out2,
                        float* out3, const float* const* in_scale, const float* const* in_shift, void* stream);

/* ---- row F (+U, +T ordering) : flow feature assembly ------------------------------------------
 * One launch builds what reference model.py:153-204 builds with ~100 ATen calls, for batch item 0..0
 * (one scene): for the 5 hypotheses depth + i*interval, i=-2..2: un-project the pixel centres of the
 * (h,w) flow grid, project into every view, bilinear-fetch the three (already resized to (h,w)) pyramid
 * levels maps1/2/3 -- CHANNEL-LAST (V,h,w,c_l), c_l % 4 == 0, as pf_flow_pyramid_f32 writes them --
 * variance over views, append (world-mean)/std repeated 8x.
 * depth_in (dh,dw) is nearest-resized to (h,w) on the fly (model.py:153-158).  `interval` is a DEVICE pointer
 * to the hypothesis spacing (one float): scene constants live in device memory so that a captured
 * hipGraph of the whole forward can be replayed on a new scene by refreshing small buffers.
 * Points are written in SUB-GRID-MAJOR order for the test-mode tiling of model.py:231-267
 * (ratio r, G = r*r groups, Ng = 5*(h/r)*(w/r) points each; group g=(y%r)*r+(x%r), local index
 * d*(h/r)*(w/r) + (y/r)*(w/r) + (x/r)); r = 1 gives the plain (5,h,w) lattice.
 *   feature (G*Ng, c1+c2+c3+24) POINT-major (the first EdgeConv GEMM's A operand)    xyz (G, 3, Ng) */
int pf_flow_features_f32(const float* maps1, const float* maps2, const float* maps3, int c1, int c2, int c3,
                         int V, int h, int w, const float* depth_in, int dh, int dw, const float* interval,
                         const float* cam, int ratio, float* feature, float* xyz, void* stream);

/* ---- rows E0/E1/E2/M building blocks ----------------------------------------------------------
 * Points are organised as G groups of Ng points; a "tile" is PF_GEMM_TILE consecutive points of one
 * group; a launch uses T = pf_stat_blocks(G, Ng) blocks per group, each reducing its tiles into one
 * float64 partial (sum, sum of squares) per column -> deterministic BatchNorm statistics. */
int pf_stat_blocks(int G, int Ng);
/* Blocks per group used by pf_pointwise_gemm_f32 (128-point tiles); its partials are (G, Tg, Nc, 2). */
int pf_gemm_blocks(int G, int Ng);

/* BatchNorm (training mode) statistics -> affine.  Reduces partials (G, T, pcols, 2) over the T blocks
 * of `groups_per_stat` consecutive groups, columns [col0, col0+C):  mean = S/count,
 * var = SS/count - mean^2 (biased, used to normalise), scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale; written to scale/shift[s*ld_affine + c].  When running_mean != NULL the
 * running statistics are updated S times in group order exactly like S successive nn.BatchNorm
 * calls (momentum, unbiased variance with n = unbias_n; reference runs BN in train mode at test time,
 * test.py:58).  count = elements per STAT group behind the sums. */
int pf_bn_finalize_f32(const double* partials, int T, int pcols, int col0, int C, double count,
                       double unbias_n, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int G, int groups_per_stat,
                       float* scale, float* shift, int ld_affine, void* stream);

/* The same, for 1..32 independent jobs in one launch (e.g. the central and the difference half of an
 * EdgeConv BatchNorm, reference networks.py:33-36).  Fields = the arguments of pf_bn_finalize_f32; the
 * jobs must write disjoint scale/shift/running-stat ranges. */
typedef struct pf_bn_job {
  const double* partials;
  int32_t T, pcols, col0, C;
  double count, unbias_n;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float momentum, eps;
  int32_t G, groups_per_stat;
  float* scale;
  float* shift;
  int32_t ld_affine;
  /* != 0: scale / shift are rows 0 / 1 of a rows tensor (4, S, ld_affine) = [scale | shift | mean | invstd] -- shift =
   * scale + S * ld_affine -- and pf_bn_finalize_jobs_f32 also writes rows 2 / 3, the batch statistics a BatchNorm
   * BACKWARD needs (the training step).  Consumers that resolve the job themselves (`in_bn`) ignore it.  (The field
   * sits in what was the struct's tail padding: size and offsets of everything else are unchanged.) */
  int32_t rows4;
} pf_bn_job;
int pf_bn_finalize_jobs_f32(const pf_bn_job* jobs, int njobs, void* stream);

/* The finalize FOLDED INTO THE CONSUMER (`in_bn`, host pointer, of pf_pointwise_gemm_f32 / pf_conv2d_wide_f32 /
 * pf_flow_head_f32; csrc/pf_bn_resolve.h): instead of (in_scale, in_shift) rows the consumer gets the pf_bn_job of
 * the pending BatchNorm -- finished statistics rows `partials` (G, T, pcols, 2) of the PRODUCER launch, count,
 * gamma, beta, eps -- and every block computes the (scale, shift) of its statistic group itself while its first
 * loads are in flight (fixed summation order: bit-reproducible).  The job's scale / shift rows and running
 * statistics are NOT written by such a consumer: pf_bn_finalize_jobs_f32 on the same job, any time later and off
 * the critical path, does that.  A consumer that cannot resolve (shape outside its fast path, more than 4096
 * rows behind a statistic) runs the finalize itself, without the running-statistics update, into job.scale /
 * job.shift (which must then be valid) and proceeds with those rows. */

/* Y[m, 0:Nc_store] = act(X[m, 0:K]) * Wt, m over G*Ng points.  Wt is (K, Nc) row-major, Nc in
 * {32, 64, 128}.  X is channel-major (G, K, Ng) when x_point_major == 0 (the reference (B,C,N) layout)
 * or point-major rows of ldx floats.  act = ReLU(x*in_scale[s,k] + in_shift[s,k]) when in_scale != NULL
 * (s = g / groups_per_stat) -- the previous layer's BatchNorm+ReLU fused into the load -- else identity.
 * col_partials (G, pf_gemm_blocks(G,Ng), Nc, 2) float64 or NULL receives per-block column sums of Y. */
int pf_pointwise_gemm_f32(const float* X, int x_point_major, int64_t ldx, const float* Wt, float* Y,
                          int64_t ldy, int G, int Ng, int K, int Nc, int Nc_store, const float* in_scale,
                          const float* in_shift, const pf_bn_job* in_bn, int groups_per_stat, double* col_partials,
                          void* stream);

/* Pass A of EdgeConv: for rows LE = [l (C) | e (C)] (point-major, ldle floats per point) and local
 * neighbour indices idx (G, Ng, k): partial sums over all (point, neighbour) pairs of
 * d = e[idx] - l and d*d per channel -> partials (G, T, C, 2) float64.   C in {32, 64, 128}.
 * Neighbourhood, two forms (both passes): `idx` (G, Ng, k) int64 group-local indices -- the reference's
 * tensor (functions/gather_knn.py:10-24) -- or, when `codes` != NULL (then idx may be NULL and k must be 16),
 * the window codes pf_knn_lattice_f32 wrote, (G, Ng, 16) uint8: neighbour j of point n is
 * clamp(n + (pd-hk)*lat_h*lat_w + (ph-hk)*lat_w + (pw-hk), 0, Ng-1) with code = (pd*lat_ks + ph)*lat_ks + pw,
 * hk = lat_ks/2 -- get_knn_3d's own index arithmetic (utils/torch_utils.py:51-59) done on the fly, so the six
 * gather passes of a PointFlow iteration read 16 instead of 128 index bytes per point.  lat_ks in {3, 5}. */
int pf_edge_stats_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                      double* partials, const uint8_t* codes, int lat_ks, int lat_h, int lat_w, void* stream);


/* Pass B of EdgeConv (reference networks.py:37-43 / :74-79):
 *   concat != 0: Y[m, 0:C]  = relu(l*scale[0:C] + shift[0:C])                       (central half)
 *                Y[m, C:2C] = mean_j relu((e[idx_j]-l)*scale[C:2C] + shift[C:2C])
 *   concat == 0: Y[m, 0:C]  = mean_j relu((e[idx_j]-l)*scale[0:C] + shift[0:C])
 * scale/shift are (S, ld_affine).  Y is point-major with ldy floats per point. */
int pf_edge_apply_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                      const float* scale, const float* shift, int ld_affine, int groups_per_stat,
                      int concat, float* Y, int64_t ldy, const uint8_t* codes, int lat_ks, int lat_h, int lat_w,
                      void* stream);

/* ---- EdgeConv backward (training, BASELINE config 4) ------------------------------------------------
 * Gradient of pf_edge_apply's output w.r.t. the rows LE = [l | e] through the train-mode BatchNorm, without
 * the (N, k, C) edge tensor the reference's autograd keeps (networks.py:18-45; its scatter is
 * functions/csrc/gather_knn_kernel.cu:50-89): both passes recompute d = e[idx] - l.
 * grad_y (G*Ng, ldg) point-major: [G_central C | G_diff C] (concat) or [G_diff C]; scale/shift/mean/invstd
 * rows (S, ld_affine) in the same column order (scale = gamma*invstd, shift = beta - mean*scale of the
 * forward; mean / invstd = its batch statistics).
 *   reduce: partials (G, pf_stat_blocks(G,Ng), cols, 2) float64, cols = 2C | C, per column
 *           (sum g, sum g*xhat) with g = [u > 0] * G / k (central half: [u > 0] * G) -> dbeta, dgamma
 *   apply : grad_le (G*Ng, ldle) = [dl | de]; c1 = dbeta/M, c2 = dgamma/M rows (S, ld_affine), M = elements
 *           behind the statistics of that column (diff: gps*Ng*k; central: gps*Ng).  With inv_order /
 *           inv_start (pf_knn_inverse below) de is gathered over the inverted lists, columns [0, 2C) of every row
 *           are written with plain stores and the result is bit-reproducible; with NULL grad_le is zeroed by
 *           the call and the de rows are accumulated with float atomics (as the reference's scatter).
 * C in {32, 64}. */
int pf_edge_backward_reduce_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                                const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int ld_affine, int groups_per_stat,
                                int concat, double* partials, void* stream);
int pf_edge_backward_apply_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                               const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* c1, const float* c2,
                               int ld_affine, int groups_per_stat, int concat, float* grad_le,
                               const uint32_t* inv_order, const uint32_t* inv_start, void* stream);

/* The same backward in TWO walks over the neighbourhoods instead of three (round 6).  dl = -sum_j dd_j is linear in
 * (c1, c2): dl = -a * ((sum_j g_j - k * c1) - c2 * sum_j xhat_j), so the reduce pass can leave every point's own sums
 * behind and nobody has to walk the forward lists again.
 *   sums  : pf_edge_backward_reduce_f32 + grad_le (G*Ng, ldle): row n = [sum_j g_j (C) | sum_j xhat_j (C)].
 *           grad_acc (G*Ng, ld_acc) or NULL: a second upstream gradient in grad_y's column order (the next layer's data
 *           gradient, model.py:209-216's chain); the pass uses grad_y + grad_acc and stores that sum back into grad_acc,
 *           which the caller hands to `finish` as its grad_y (one element-wise launch less per layer).
 *   finish: (after pf_edge_backward_coeffs_f32) the gather over the inverted lists (inv_order / inv_start are required)
 *           writes de AND turns the row's sums into dl (+ the central half of a concat layer): grad_le = [dl | de], plain
 *           stores, bit-reproducible.  dl differs from pf_edge_backward_apply_f32's in float32 rounding only (the sum is
 *           taken before the coefficients are applied instead of after).
 * plane: points per lattice plane (H*W of the D x H x W lattice the neighbours come from) or 0 = unknown -- a hint for
 * the XCD-aware block order only (each of the 8 L2s serves a band of pixel rows in every plane instead of every eighth
 * tile of the whole lattice); results do not depend on it.  pf_edge_stats_f32 / pf_edge_apply_f32 take the same hint as
 * (lat_ks 0, lat_h, lat_w) when they are given an index tensor instead of window codes.
 * Replaces networks.py:18-45's autograd graph like the three-pass form above. */
int pf_edge_backward_sums_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                              const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                              const float* mean, const float* invstd, int ld_affine, int groups_per_stat, int concat,
                              double* partials, float* grad_le, float* grad_acc, int64_t ld_acc, int plane,
                              void* stream);
int pf_edge_backward_finish_f32(const float* LE, int64_t ldle, int C, int k, int G, int Ng, const float* grad_y,
                                int64_t ldg, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* c1, const float* c2, int ld_affine,
                                int groups_per_stat, int concat, float* grad_le, const uint32_t* inv_order,
                                const uint32_t* inv_start, int plane, void* stream);

/* Between the two passes above: partials (G, pf_stat_blocks(G, Ng), cbn, 2) -> c1, c2 (G / groups_per_stat, cbn) =
 * the statistic set's (sum g', sum g' * xhat) / m in a fixed order (m = groups_per_stat * Ng points for the central
 * half [0, C) of a concat layer, * k pairs else), and dbeta / dgamma (cbn,) = the sums over the sets (NULL: not
 * wanted; accumulate != 0: added to what is there).  Replaces a dozen element-wise ATen launches per layer in the
 * training step (reference: autograd through networks.py:36-45). */
int pf_edge_backward_coeffs_f32(const double* partials, int G, int T, int cbn, int C, int concat, int groups_per_stat,
                                int Ng, int k, float* c1, float* c2, float* dgamma, float* dbeta, int accumulate,
                                void* stream);

/* The inverse of a neighbour index tensor idx (G, Ng, k) (csrc/knn_inverse.hip): order (G*Ng*k) = the pair ids
 * p = (g*Ng + n)*k + j grouped by their target row g*Ng + clamp(idx[p], 0, Ng-1); start (G*Ng + 1): the pairs
 * that gather row m are order[start[m] .. start[m+1]), in ascending p.  With (inv_order, inv_start)
 * pf_edge_backward_apply_f32 computes the de rows as a GATHER over those lists -- plain stores in a fixed summation
 * order, bit-reproducible -- instead of the float atomics of the reference's scatter
 * (functions/csrc/gather_knn_kernel.cu:50-89); NULL keeps the atomics.  One inversion serves every layer that
 * shares idx.  A counting sort in kernel launches only (no memset / memcpy nodes, no library call: capturable in a
 * hipGraph).  workspace: pf_knn_inverse_workspace(G, Ng, k) bytes of device scratch. */
int64_t pf_knn_inverse_workspace(int G, int Ng, int k);
int pf_knn_inverse(const int64_t* idx, int k, int G, int Ng, uint32_t* order, uint32_t* start, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* ---- train-mode BatchNorm for the conv stacks around the path (ImageConv / VolumeConv) ----------
 * x (N, C, S) contiguous (NCHW / NCDHW with S = spatial size).  pf_channel_stats_f32 writes float64
 * partial (sum, sum of squares) per (sample, block, channel): partials (N, pf_norm_blocks(S), C, 2), the
 * layout pf_bn_finalize_f32 reduces (G = N samples; groups_per_stat samples pooled per statistic, e.g.
 * the batch of one reference conv call, nn/conv.py:29-35).  pf_channel_affine_f32 applies
 * y = x*scale[s,c] + shift[s,c] (s = n / samples_per_stat) with optional ReLU; y may alias x. */
int pf_norm_blocks(int64_t S);
int pf_channel_stats_f32(const float* x, int64_t N, int64_t C, int64_t S, double* partials, void* stream);
int pf_channel_affine_f32(const float* x, float* y, const float* scale, const float* shift, int64_t N, int64_t C,
                          int64_t S, int samples_per_stat, int relu, void* stream);
/* Finalize + normalise in one launch: every block reduces the partials (N, T, C, 2) of its own
 * (stat group, channel) in a fixed order, then streams y = act(gamma*(x-mean)*rsqrt(var+eps)+beta);
 * one block per channel applies the running-statistics recurrence over the N/samples_per_stat stat groups
 * in order.  count = elements per stat group (= samples_per_stat * S).  addend != NULL (same shape as x):
 * y = addend + act(...), the additive skip of VolumeConv's decoder (reference networks.py:166) in the same pass. */
int pf_channel_bn_apply_f32(const float* x, float* y, const double* partials, int T, int64_t N, int64_t C, int64_t S,
                            int samples_per_stat, double count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int relu,
                            const float* addend, void* stream);
/* The same for TWO convolution outputs at once: y = relu(bn2(x2)) + relu(bn1(x1)) (train-mode statistics from
 * partials1 / partials2, running statistics of both updated), y may alias x1 or x2 -- the last skip add of
 * VolumeConv's decoder (reference networks.py:166) as one pass over three streams. */
int pf_channel_bn_apply2_f32(const float* x1, const double* partials1, int T1, const float* gamma1, const float* beta1,
                             float* running_mean1, float* running_var1, float momentum1, float eps1, const float* x2,
                             const double* partials2, int T2, const float* gamma2, const float* beta2,
                             float* running_mean2, float* running_var2, float momentum2, float eps2, float* y, int64_t N,
                             int64_t C, int64_t S, int samples_per_stat, double count, void* stream);
/* Statistics + finalize + normalise in ONE launch for small tensors (one 1024-thread block per channel walks
 * the stat groups in order): y = act(BN_train(x)) with y == x allowed, and/or (y == NULL) only the affine
 * rows scale/shift (N/samples_per_stat, ld_affine) for a consumer that applies them itself.  Same
 * running-statistics semantics as pf_bn_finalize_f32. */
int pf_channel_bn_fused_f32(const float* x, float* y, int64_t N, int64_t C, int64_t S, int samples_per_stat,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, int relu, float* scale, float* shift, int ld_affine,
                            void* stream);

/* ---- row R : 3x3x3 convolution of VolumeConv on the f32 matrix cores ---------------------------------
 * Replaces nn.Conv3d(k=3, padding=1, stride 1|2, bias=False) inside the Conv3d blocks of reference
 * networks.py:133-142 (nn/conv.py:108,115-121).  x (N,Cin,Di,Hi,Wi) NCDHW, Cin % 4 == 0, Cout <= 32;
 * wp = weights packed on the host as (Cin/4, 27 taps [kd][kh][kw], 4, 16*ceil(Cout/16)), zero padded:
 * wp[g][tap][k][co] = W[co][4g+k][kd][kh][kw];  y (N,Cout,Do,Ho,Wo).
 * partials (N, pf_conv3d_blocks(...), Cout, 2) float64 or NULL receives the per-block (sum, sum of
 * squares) of y per channel -- the BatchNorm batch statistics for free.
 * in_scale / in_shift (N / samples_per_stat, Cin) or in_bn: the pending BatchNorm + ReLU of x, applied while a
 * channel is staged (zero padding after it) -- rows, or resolved by the launch itself ("the finalize folded into
 * the consumer" above); all NULL: x is taken as it is. */
int pf_conv3d_blocks(int64_t Cin, int64_t Cout, int64_t Di, int64_t Hi, int64_t Wi, int stride);
int pf_conv3d_k3_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                     int64_t Hi, int64_t Wi, int stride, const float* in_scale, const float* in_shift,
                     const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream);
/* The same convolution for stride 1 and Cout <= 8 (VolumeConv's conv0_1, networks.py:136: 64 -> 8 on the full cost
 * volume) without the half-empty 16-wide tile: N = 8 channels x 2 adjacent output rows, which read the same four
 * input rows, so K = 36 taps instead of 27 and 1.5x fewer MFMA cycles.  wp = weights packed as
 * (Cin/4, 36 taps [kd][kh'][kw], 4, 16): wp[g][tap][k][c + 8 s] = W[c][4g+k][kd][kh' - s][kw] (zero where kh' - s
 * is outside [0,2]).  partials (N, pf_conv3d_pair_blocks(...), Cout, 2) float64 or NULL as above. */
int pf_conv3d_pair_blocks(int64_t Cin, int64_t Cout, int64_t D, int64_t H, int64_t W);
int pf_conv3d_k3_pair_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t D,
                          int64_t H, int64_t W, double* partials, void* stream);
/* 3x3x3 / pad 1 / stride 1 conv3d with Cout <= 4 (VolumeConv's 8 -> 1 output layer, networks.py:147);
 * w is the unpacked (Cout, Cin, 3, 3, 3) weight. */
int pf_conv3d_k3_few_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t D,
                         int64_t H, int64_t W, void* stream);

/* ConvTranspose3d 3x3x3, stride 2, padding 1, output_padding 1 (VolumeConv decoder, reference
 * networks.py:141-143 via nn/conv.py:189-216): y (N, Cout, 2D, 2H, 2W) from xa (+ xb when not NULL: the
 * decoder's skip add, networks.py:163-165) (N, Cin, D, H, W) and w (Cin, Cout, 3, 3, 3) in
 * nn.ConvTranspose3d's own layout.  partials != NULL: float64 (sum, sum of squares) per (sample, block,
 * channel), (N, pf_deconv3d_blocks(D, H, W), Cout, 2), the layout pf_channel_bn_apply_f32 consumes.
 * in_scale / in_shift (N / samples_per_stat, Cin) or in_bn: the pending BatchNorm + ReLU of xa (the previous
 * layer's raw output; Cin <= 64), applied to every loaded value BEFORE the skip add -- rows, or resolved by the
 * launch itself; all NULL: xa is taken as it is. */
int pf_deconv3d_blocks(int64_t D, int64_t H, int64_t W);
int pf_deconv3d_k3s2_f32(const float* xa, const float* xb, const float* w, float* y, int64_t N, int64_t Cin,
                         int64_t Cout, int64_t D, int64_t H, int64_t W, const float* in_scale, const float* in_shift,
                         const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream);

/* The bottom of VolumeConv's U-Net (reference networks.py:136-141) on volumes of a few thousand voxels, one launch
 * per layer (csrc/conv3d_bottom.hip): 3x3x3 / pad 1 convolutions 32 -> 64 stride 2 and 64 -> 64 stride 1, and the
 * ConvTranspose3d 64 -> 32 (3x3x3, stride 2, pad 1, output_padding 1: output = twice the input per dimension).
 * x (N, Cin, Di, Hi, Wi) -> y (N, Cout, Do, Ho, Wo); in_scale / in_shift / in_bn / samples_per_stat: the pending
 * BatchNorm + ReLU of the input as in pf_conv2d_wide_f32; partials (N, *_blocks(...), Cout, 2) float64 or NULL:
 * per-block sums of y and y*y.  Weights packed for the f32 matrix cores:
 *   conv   wp (3, 3, 3, Cin/16, 4, 64, 4):  wp[kd][kh][kw][kc][kq][co][j] = w[co][16 kc + 4 kq + j][kd][kh][kw]
 *   deconv wp (27, 4, 4, 32, 4):            wp[tap][kc][kq][co][j] = w[16 kc + 4 kq + j][co][tap]   (Cin-major
 *                                           weight of ConvTranspose3d, tap = (kd*3 + kh)*3 + kw)
 * PF_ERR_UNSUPPORTED for other channel counts (the *_supported functions tell). */
int pf_conv3d_bottom_supported(int64_t Cin, int64_t Cout, int stride);
int pf_conv3d_bottom_blocks(int64_t Di, int64_t Hi, int64_t Wi, int stride);
int pf_conv3d_bottom_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                         int64_t Hi, int64_t Wi, int stride, const float* in_scale, const float* in_shift,
                         const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream);
int pf_deconv3d_bottom_supported(int64_t Cin, int64_t Cout);
int pf_deconv3d_bottom_blocks(int64_t Di, int64_t Hi, int64_t Wi);
int pf_deconv3d_bottom_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                           int64_t Hi, int64_t Wi, const float* in_scale, const float* in_shift, const pf_bn_job* in_bn,
                           int samples_per_stat, double* partials, void* stream);

/* ---- ImageConv: conv2d on the f32 matrix cores ------------------------
 * Replaces the nn.Conv2d of the Conv2d blocks of reference networks.py:89-110 (nn/conv.py:62-77) for the tower's
 * shapes: 3x3 / stride 1 / pad 1 (3->8, 3->16, 8->8, 16->16, 32->32, 64->64) and 5x5 / stride 2 / pad 2 (8->16, 16->32,
 * 32->64), bias-free (csrc/conv2d_wide.hip).  x (N,Cin,Hi,Wi) NCHW holds the RAW output of the previous conv when
 * in_scale / in_shift (N/sps, Cin) or in_bn (see "the finalize folded into the consumer" above) are given:
 * relu(x*scale+shift) -- the previous block's BatchNorm+ReLU -- is applied while staging, so that activation is
 * never written to memory; samples_per_stat consecutive samples share one affine row.  partials
 * (N, pf_conv2d_wide_blocks(...), Cout, 2) float64 or NULL receives the BatchNorm statistics of y (per-block sums
 * of y and y*y).  wp is the weight packed
 * (K, K, Cin/8, 2, Cout, 4): wp[kh][kw][kc][h][co][j] = w[co][8 kc + 4 h + j][kh][kw] for Cout 32 / 64, and
 * (K, K, 4, 16, Cin'/4), Cin' = Cin rounded up to 4: wp[kh][kw][kq][co][j] = w[co][(Cin'/4) kq + j][kh][kw] for
 * Cout 16 and the 3 -> 8 image layer (zero where co >= Cout or the channel does not exist), and for 8 -> 8 the PAIRED-ROWS
 * form (K + 1, K, 4, 16, Cin'/4): wp[kh'][kw][kq][co + 8 s][j] = w[co][(Cin'/4) kq + j][kh' - s][kw], zero where kh' - s
 * is outside [0, K): the 16 matrix columns are 8 channels x two adjacent output rows.
 * out_channel_last != 0 (Cout 32 / 64 only): y is written (N, Ho, Wo, Cout) -- the coarse tower's last layer
 * feeds pf_frustum_variance_cl_f32 directly, without the pf_nchw_to_nhwc_f32 pass.
 * PF_ERR_UNSUPPORTED for any other shape (pf_conv2d_wide_supported tells). */
int pf_conv2d_wide_supported(int64_t Cin, int64_t Cout, int kernel_size, int stride);
int pf_conv2d_wide_blocks(int64_t Cout, int64_t Hi, int64_t Wi, int stride);
int pf_conv2d_wide_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Hi,
                       int64_t Wi, int kernel_size, int stride, const float* in_scale, const float* in_shift,
                       const pf_bn_job* in_bn, int samples_per_stat, double* partials, int out_channel_last,
                       void* stream);
/* The same launch over `sets` (1 or 2) PARAMETER SETS -- the model's two towers (coarse: model.py:71-77, flow:
 * model.py:140-148) have identical shapes and run on the same images, so each of their eleven layers is ONE launch:
 * samples [s*N/sets, (s+1)*N/sets) convolve with the packed weights at wp + s*wp_set_stride (floats, a multiple
 * of 4) and take their pending BatchNorm from in_bn[s] (an array of `sets` jobs; the (in_scale, in_shift) rows are
 * indexed by the global statistic group n / samples_per_stat as before).  x_layout says where sample
 * n = s*N/sets + i reads its input: 0 = sample n of x (N samples); 1 = sample i (x holds N/sets samples that
 * every set reads); 2 = sample i*sets + s (x is (N/sets, sets, Cin, Hi, Wi): what ONE convolution with the sets'
 * output channels stacked wrote -- the towers' first layer, 3 -> 8 + 8 on the same views, is a single
 * pf_conv2d_wide_f32 call with Cout = 16 and fills every column of the matrix tile).  Bit s of out_channel_last selects
 * the (Ho, Wo, Cout) layout for the samples of set s (each sample's region of y has the same size either way).
 * Per sample the arithmetic is that of pf_conv2d_wide_f32: results are bit-identical to `sets` separate calls. */
int pf_conv2d_wide_sets_f32(const float* x, int x_layout, const float* wp, int64_t wp_set_stride, int sets, float* y,
                            int64_t N, int64_t Cin, int64_t Cout, int64_t Hi, int64_t Wi, int kernel_size, int stride,
                            const float* in_scale, const float* in_shift, const pf_bn_job* in_bn, int samples_per_stat,
                            double* partials, int out_channel_last, void* stream);

/* ---- rows M (last layer) + H + T : flow head ---------------------------------------------------
 * Z (G*Ng, ldz) holds the pre-BN output of the 64->16 MLP layer.  Per pixel of the (h,w) grid:
 * a = relu(Z*scale+shift); flow_d = sum_c w_out[c]*a_c for the 5 hypotheses; p = softmax(-flow);
 * depth_out = nearest(depth_in) + sum_d p_d*(d-2)*interval  (reference model.py:218-227), written
 * back in image order (undoing the sub-grid-major point order, model.py:256-266).
 * flow_prob (5,h,w), depth_out (h,w). */
int pf_flow_head_f32(const float* Z, int64_t ldz, const float* scale, const float* shift, int ld_affine,
                     const pf_bn_job* in_bn, const float* w_out, const float* depth_in, int dh, int dw,
                     const float* interval, int h, int w, int ratio, float* flow_prob, float* depth_out,
                     void* stream);

/* ---- row S : soft-argmin + probability map -----------------------------------------------------
 * cost (B, D, HW) filtered cost volume; depth = sum_k linspace(start,end,D)[k] * softmax(-cost)[k]
 * (reference model.py:117-124); prob = p[floor(i)] + p[ceil(i)], i = (depth-start)/interval clamped
 * to [0, D-1] (functions/functions.py:141-175).  params (B,3) = (depth_start, depth_end, interval). */
int pf_softargmin_prob_f32(const float* cost, const float* params, float* depth, float* prob, int64_t B,
                           int64_t D, int64_t HW, void* stream);

/* ---- the step after the path: evaluation output + fusion pre-step -------------
 * Device-side halves of reference utils/eval_file_logger.py:12-79 and tools/depthfusion.py:153-170, writing into a
 * staging buffer in PFM row order (flip_rows != 0: bottom row first, the np.flipud of utils/io.py:124) so that one
 * asynchronous D2H copy per depth map feeds the file writer.
 *   pf_eval_pack_map_f32    src (h,w) -> dst (h,w) [row-flipped]
 *   pf_eval_flow_prob_f32   prob (5,h,w) -> confidence p[floor(i)] + p[min(floor(i)+1,4)], i = sum_d p_d (d-2) + 2
 *                           evaluated in float64 like NumPy does (eval_file_logger.py:48-62)
 *   pf_eval_prob_filter_f32 depth (h,w) with depth := 0 where flow_conf (h,w) < flow_threshold or
 *                           init_conf (ih,iw; nearest-resized, cv2.INTER_NEAREST's index rule) < init_threshold */
int pf_eval_pack_map_f32(const float* src, float* dst, int h, int w, int flip_rows, void* stream);
int pf_eval_flow_prob_f32(const float* prob, float* dst, int h, int w, int flip_rows, void* stream);
int pf_eval_prob_filter_f32(const float* depth, const float* flow_conf, const float* init_conf, int h, int w, int ih,
                            int iw, float flow_threshold, float init_threshold, float* dst, int flip_rows,
                            void* stream);


/* ==== Row Z : the training step (BASELINE config 4; reference train.py:72-82) ======================================
 * Hand-written backward for the convolution -> BatchNorm(batch statistics) -> ReLU blocks of ImageConv, VolumeConv
 * and the flow MLP (reference nn/conv.py:24-35,62-77,108-121,197-210; networks.py:84-167; model.py:40-43), replacing
 * ATen's convolution_backward / batch_norm backward (MIOpen / CK solvers) in the reference's loss.backward().
 * Everything below sums in a fixed order: gradients are bit-reproducible run to run.
 *
 * Forward finalize that keeps what the backward needs: like pf_bn_finalize_f32, but the result is ONE tensor
 * rows (4, S, ld_rows) = [scale | shift | mean | invstd], S = G / groups_per_stat; rows[0], rows[1] are the (S, ld)
 * in_scale / in_shift rows the forward kernels take.  The C channels of this call are channels [ch0, ch0 + C) of
 * gamma / beta / the running statistics and land in columns [col_out, col_out + C) of every row (EdgeConv's
 * BatchNorm covers [central | difference] halves with separate statistics, reference networks.py:33-36).  Running
 * statistics updated as by pf_bn_finalize_f32. */
int pf_bn_train_rows_f32(const double* partials, int T, int pcols, int col0, int C, double count, double unbias_n,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int G, int groups_per_stat, float* rows, int ld_rows, int col_out,
                         int ch0, void* stream);
/* BatchNorm(+ReLU) backward on planar tensors (N, C, S): g = dL/dz for z = act(y * scale + shift), y the raw
 * convolution output, rows as above (statistic group s = n / samples_per_stat).
 *   reduce : partials (N, pf_norm_blocks(S), C, 2) float64 = per-block (sum g', sum g' * xhat),
 *            g' = [y * scale + shift > 0] * g when relu != 0, else g;  xhat = (y - mean) * invstd
 *   coeffs : partials (G, T, pcols, 2) -> coef (2, S, C) = [scale * dbeta_s / count | scale * invstd * dgamma_s / count]
 *            and dgamma[c] / dbeta[c] = sum over the S statistic groups (accumulate != 0: added to what is there)
 *   apply  : dy = scale * g' - coef0 - coef1 * (y - mean)           (dy may alias g) */
int pf_bn_bwd_reduce_f32(const float* g, const float* y, const float* rows, int64_t N, int64_t C, int64_t S,
                         int samples_per_stat, int relu, double* partials, void* stream);
int pf_bn_bwd_coeffs_f32(const double* partials, int T, int pcols, int col0, int C, double count, int G,
                         int groups_per_stat, const float* rows, float* coef, float* dgamma, float* dbeta,
                         int accumulate, void* stream);
int pf_bn_bwd_apply_f32(const float* g, const float* y, const float* rows, const float* coef, float* dy, int64_t N,
                        int64_t C, int64_t S, int samples_per_stat, int relu, void* stream);
/* coeffs + apply in one launch (the planar path of the training step): every block re-adds its statistic group's
 * partial rows (pf_norm_blocks(S) * samples_per_stat of them, the same order and bits as pf_bn_bwd_coeffs_f32), block
 * (0, c, 0) writes dgamma[c] / dbeta[c] (NULL: not wanted).  count = samples_per_stat * S. */
int pf_bn_bwd_apply_fused_f32(const float* g, const float* y, const float* rows, const double* partials, int T,
                              double count, float* dy, int64_t N, int64_t C, int64_t S, int samples_per_stat, int relu,
                              float* dgamma, float* dbeta, int accumulate, void* stream);
/* The whole BatchNorm(+ReLU) backward of planar tensors in ONE launch where a plane fits one block's registers (round 6):
 * samples_per_stat == 1 (every sample its own statistic group: the towers' per-view statistics, VolumeConv's one sample),
 * S % 4 == 0, S <= 32 768 (pf_bn_bwd_plane_supported), 16-byte aligned tensors.  rows (4, N, C) as pf_bn_train_rows_f32
 * writes them; dy (N, C, S); dgamma / dbeta (C) = the sums over the N samples in sample order (added when accumulate;
 * NULL: not wanted).  One block per channel loads a plane's (g, y) once, reduces, and writes dy from the registers:
 * 12 bytes per element and one launch where pf_bn_bwd_reduce_f32 + pf_bn_bwd_apply_fused_f32 are 20 and two. */
int pf_bn_bwd_plane_supported(int64_t S, int samples_per_stat);
int pf_bn_bwd_plane_f32(const float* g, const float* y, const float* rows, int64_t N, int64_t C, int64_t S, int relu,
                        float* dy, float* dgamma, float* dbeta, int accumulate, void* stream);
/* The same on point-major rows (G groups of Ng rows, ld floats per row; C in {16, 32, 64, 128} for reduce,
 * C % 4 == 0 for apply / affine): the flow MLP's BatchNorm1d.  reduce partials: (G, pf_rows_bn_blocks(G, Ng), C, 2).
 * pf_rows_affine_f32: z = act(y * scale + shift), the normalised activation handed on. */
int pf_rows_bn_blocks(int G, int Ng);
int pf_rows_bn_bwd_reduce_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* rows, int C, int G,
                              int Ng, int groups_per_stat, int relu, double* partials, void* stream);
int pf_rows_bn_bwd_apply_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* rows,
                             const float* coef, float* dy, int64_t ldo, int C, int G, int Ng, int groups_per_stat,
                             int relu, void* stream);
int pf_rows_affine_f32(const float* y, int64_t ldy, const float* rows, float* z, int64_t ldz, int C, int G, int Ng,
                       int groups_per_stat, int relu, void* stream);

/* Weight gradient of a convolution on the f32 matrix cores (csrc/conv_wgrad.hip), one formulation for nn.Conv2d /
 * nn.Conv3d / nn.ConvTranspose3d / the 1x1 convolutions over points:
 *     dw[cg][cx][kd][kh][kw] = sum_{n, o} gr[n, cg, o] * x[n, cx, o * stride + k - pad]           (zero outside x)
 * gr (N, Cg, Do, Ho, Wo) lives on the COARSE grid, x (N, Cx, Di, Hi, Wi) on the FINE one (2-D: Do = Di = KD = 1):
 *   convolution            gr = dL/dy, x = the layer input        -> dw in (Cout, Cin, k...) order;
 *   transposed convolution gr = the layer input, x = dL/dy (stride 2, pad 1) -> dw in (Cin, Cout, k...) order.
 * x_scale / x_shift (N / x_samples_per_stat, Cx) or NULL: x holds a RAW convolution output whose BatchNorm + ReLU is
 * pending; relu(x * scale + shift) is applied while x is staged (zero padding after it).
 * workspace: pf_conv_wgrad_workspace(...) bytes of device scratch (per-split partial gradients, added in split
 * order: no atomics).  accumulate != 0: dw += instead of dw =.  Limits: kernel extents <= 7, taps/16-channel block
 * <= 28 column tiles (27 taps x 16 channels, 25 x 16, ...); PF_ERR_UNSUPPORTED otherwise. */
int64_t pf_conv_wgrad_workspace(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di,
                                int64_t Hi, int64_t Wi, int KD, int KH, int KW, int stride);
int pf_conv_wgrad_f32(const float* gr, const float* x, float* dw, int64_t N, int64_t Cg, int64_t Cx, int64_t Do,
                      int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi, int64_t Wi, int KD, int KH, int KW, int stride,
                      int pd, int ph, int pw, const float* x_scale, const float* x_shift, int x_samples_per_stat,
                      void* workspace, int64_t workspace_bytes, int accumulate, void* stream);
/* The launch plan the two weight-gradient entry points use for a shape, for tests that pin it (the plan is a pure
 * function of the shape and of the device's occupancy for the chosen instantiation): plan12 = {row tiles per block MT,
 * tile TD, tile TH, channel sub-blocks per block, channels per sub-block, accumulator tiles per wave, position splits,
 * channel blocks, row blocks, LDS bytes, position tiles per block, stride}. */
int pf_conv_wgrad_plan(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi,
                       int64_t Wi, int KD, int KH, int KW, int stride, int* plan12);
int pf_rows_wgrad_plan(int64_t P, int Cg, int Cx, int* plan12);
/* The same for a 1x1 convolution over point-major rows: dw[cg][cx] = sum_p gr[p, cg] * act(x[p, cx]);
 * gr (P, ldg), x (P, ldx), Cg % 4 == Cx % 4 == 0; x_scale / x_shift rows (P / x_rows_per_stat, Cx) or NULL. */
int64_t pf_rows_wgrad_workspace(int64_t P, int Cg, int Cx);
int pf_rows_wgrad_f32(const float* gr, int64_t ldg, const float* x, int64_t ldx, float* dw, int64_t P, int Cg, int Cx,
                      const float* x_scale, const float* x_shift, int64_t x_rows_per_stat, void* workspace,
                      int64_t workspace_bytes, int accumulate, void* stream);
/* Several layers' pf_conv_wgrad_f32 (partials only, as with dw == NULL) in as few launches as their kernel instantiations
 * allow (round 6): a training node queues the weight gradients of its layers -- nothing in the step waits for them -- and
 * issues them together once its data-gradient chain is done; layers that share an instantiation (VolumeConv's 96-384-block
 * layers below 24x32x40 beside conv1_0; the 25 600-point PointFlow iteration's 1x1 layers beside the 102 400-point one's)
 * ride in ONE grid.  Each item's arguments mean what pf_conv_wgrad_f32's mean; the
 * partials land in item.workspace (>= pf_conv_wgrad_workspace bytes) in the layout described below and are summed by
 * pf_wgrad_reduce_batch_f32.  n <= 64. */
typedef struct pf_wgrad_item {
  const float* gr;
  const float* x;
  int64_t N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi;
  int KD, KH, KW, stride, pd, ph, pw;
  int x_samples_per_stat;
  const float* x_scale;
  const float* x_shift;
  void* workspace;
  int64_t workspace_bytes;
  /* rows_P > 0: the item is a pf_rows_wgrad_f32 call instead -- gr (rows_P, ldg), x (rows_P, ldx) point-major rows, Cg / Cx
   * columns of them, x_rows_per_stat rows behind one (x_scale, x_shift) row; the grid / kernel / pad fields are unused
   * (stride must be 1). */
  int64_t rows_P, ldg, ldx, x_rows_per_stat;
} pf_wgrad_item;
int pf_conv_wgrad_batch_f32(const pf_wgrad_item* items, int n, void* stream);
/* dw == NULL in the two calls above: the split partials only.  The workspace then holds (splits, Cg, taps, Cx), splits =
 * workspace bytes / (4 * Cg * Cx * taps) -- the channel index fastest: what the kernel's lanes store as 64-byte runs
 * (round 6; it was (splits, Cg, Cx, taps)).  pf_wgrad_reduce_batch_f32 adds the partials of n layers in ONE launch, each
 * in the same fixed order as the single call (bit-identical results), and puts the sums into nn.ConvNd's order:
 * rows[i] = the Cg and taps[i] the KD * KH * KW (1 for pf_rows_wgrad_f32) of the call that wrote parts[i];
 *   swapped[i] == 0:  dws[i][(a * B + b) * T + t]         (+)= sum_k parts[i][k][(a * T + t) * B + b],  B = elems / (rows * T);
 *   swapped[i] != 0:  dws[i][(b * A + a) * T + T - 1 - t] (+)= the same sum,  A = rows[i].
 * SWAPPED: for a stride-1 'same' convolution the two operands of pf_conv_wgrad_f32 may change places -- gr' = x,
 * x' = dL/dy -- which gives dW'[cx][cg][k'] = dW[cg][cx][K-1-k'] per axis: the rows of the MFMA tile are then the layer's
 * INPUT channels and the patch staged with its halo is the (narrow) gradient tensor.  VolumeConv's conv0_1 (64 -> 8: 8 of
 * 16 MFMA rows used, a 64-channel halo patch per tile) and conv6_2 (8 -> 1: 1 of 16 rows) take it (reference
 * networks.py:134,147 under train.py:80; round 6: 185 -> 115 us and 38 -> 27 us at config 4). */
int pf_wgrad_reduce_batch_f32(const float* const* parts, float* const* dws, const int64_t* elems, const int* splits,
                              const int* rows, const int* taps, const int* swapped, int n, int accumulate, void* stream);


/* Data gradient of ImageConv's 5x5 / stride 2 / pad 2 convolutions (reference networks.py:93,98,103), i.e.
 * ConvTranspose2d(5, stride 2, pad 2, output_padding 1): dy (N, Cout, Ho, Wo) -> dx (N, Cin, 2 Ho, 2 Wo);
 * wp = the convolution's weight W (Cout, Cin, 5, 5) packed (Cout/4, 25 taps [kh][kw], 4, 16*ceil(Cin/16)):
 * wp[g][tap][k][ci] = W[4g + k][ci][kh][kw], zero padded.  Cout % 4 == 0, Cin <= 32.  (csrc/conv_dgrad.hip) */
int pf_deconv2d_k5s2_supported(int64_t Cout, int64_t Cin);
int pf_deconv2d_k5s2_f32(const float* dy, const float* wp, float* dx, int64_t N, int64_t Cout, int64_t Cin, int64_t Ho,
                         int64_t Wo, void* stream);
/* 3x3x3 / pad 1 / stride 1 conv3d of a ONE-channel volume x (N, 1, D, H, W) with w (Cout <= 8, 27): the data
 * gradient of VolumeConv's 8 -> 1 output layer (reference networks.py:147) when w holds the flipped kernels. */
int pf_conv3d_k3_c1_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cout, int64_t D, int64_t H,
                        int64_t W, void* stream);


/* ---- Row Z : backward of the fused warp + variance stages without atomics (csrc/warp_bwd.hip) -------------------
 * Replaces, in the reference's loss.backward(), grid_sample's backward (a float-atomic scatter-add,
 * utils/feature_fetcher.py:55) and the element-wise chain of the variance (model.py:103-111, :187-190) for the coarse
 * cost volume and for the flow feature assembly.  One scene; maps channel-last (V, H, W, C) as the forward kernels
 * read them; points n = d * H * W + y * W + x.  Gradients never flow into the sampling positions
 * (utils/feature_fetcher.py:29).
 *
 * pf_sort_pairs_by_key: counting sort of `pairs` pair ids by keys[p] < nkeys (larger keys are dropped): start
 * (nkeys + 1), order = the ids grouped by key, ascending inside a group.  Kernel launches only (capturable). */
int64_t pf_sort_pairs_workspace(int64_t pairs, int64_t nkeys);
int pf_sort_pairs_by_key(const uint32_t* keys, int64_t pairs, int64_t nkeys, uint32_t* order, uint32_t* start,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* taps: for every (view v, point n), pair id p = v * N + n: keys[p] = v * (H+1) * (W+1) + (yi+1) * (W+1) + (xi+1) for
 * the north-west tap (yi, xi) of its bilinear footprint (0xffffffff when no tap is inside the map) and
 * fxy[p] = (fx, fy), the fractional offsets.  flow: the points of pf_flow_features_f32 (depth (H, W) at the flow
 * resolution, ratio 1); frustum: the points of pf_frustum_variance_f32 (skip_view0 != 0: view 0 gets no keys, it
 * contributes its un-warped map, model.py:103-106). */
int pf_warp_taps_flow_f32(const float* depth, const float* interval, const float* cam, int V, int H, int W,
                          uint32_t* keys, float* fxy, void* stream);
int pf_warp_taps_frustum_f32(const float* kinv, const float* rinv, const float* t, const float* depths, const float* K,
                             const float* E, int V, int H, int W, int D, int skip_view0, uint32_t* keys, float* fxy,
                             void* stream);
/* gval (V, N, ctot) = (2 / V) * dvar[n, c] * (f_v[n, c] - mean_v f), ctot = c1 + c2 + c3 (levels concatenated, c_l % 4
 * == 0, c2 / c3 may be 0); dvar rows (N, ldv) point-major, or (ctot, N) channel-major with ldv = -1 (a cost volume's
 * gradient as it is). */
int pf_variance_grad_f32(const float* maps1, int c1, const float* maps2, int c2, const float* maps3, int c3, int V, int H,
                         int W, int64_t N, const uint32_t* keys, const float* fxy, const float* dvar, int64_t ldv,
                         int ref_override, float* gval, void* stream);
/* dmaps[v] (H, W, ctot) for v in [v0, V): every texel adds weight * gval over the sorted lists of the four cells whose
 * pairs have it as a tap -- plain stores, fixed order. */
int pf_warp_gather_f32(const float* gval, const float* fxy, const uint32_t* order, const uint32_t* start, int V, int v0,
                       int H, int W, int ctot, float* dmaps, void* stream);
/* ddepth (H, W): gradient w.r.t. the prior depth through the xyz features (24 columns from c0 of the point's
 * feature row; reference model.py:178-194). */
int pf_flow_depth_grad_f32(const float* dfeat, int64_t ld, int c0, const float* cam, int H, int W, float* ddepth,
                           void* stream);
/* Adjoint of the bilinear resize (align_corners = False) of pf_flow_pyramid_f32 for one level: dres (V, OH, OW, ld)
 * channel-last, columns [c0, c0 + C) -> dlevel (V, C, IH, IW). */
int pf_resize_bilinear_backward_f32(const float* dres, int ld, int c0, int C, int V, int OH, int OW, int IH, int IW,
                                    float* dlevel, void* stream);


/* Weight packing for a whole training step in ONE launch (csrc/norm_bwd.hip): `table` is a device array of `npacks`
 * descriptors (pf_pack_desc_bytes() bytes each; layout in pointmvsnet_amd/train_packs.py), each an affine gather
 *     dst[d_0 .. d_{n-1}] = src[sum_k a_k * sstride_k],  a_k = off_k + sum_i M[k][i] * d_i,  0 unless 0 <= a_k < lim_k
 * -- the MFMA operand layouts, zero paddings, flips and transpositions the forward and backward kernels of the step
 * read their weights in.  max_total: the largest destination element count in the table. */
int pf_pack_desc_bytes(void);
int pf_pack_gather_f32(const void* table, int npacks, long long max_total, void* stream);

/* ---- Row Z: the small differentiable heads of the training step (csrc/train_heads.hip) --------------------------
 * One or two launches each where autograd makes 15-25 element-wise ATen launches; float64 fixed-order reductions.
 *
 * pf_softargmin_backward_f32: backward of row S's depth (reference model.py:117-124): gcost[b,k,i] =
 *   -gdepth[b,i] * p_k * (z_k - depth[b,i]) with p = softmax(-cost) recomputed as pf_softargmin_prob_f32 rounds it.
 * pf_flow_head_train_f32: act (5*hw, ld >= 16) rows (row = d*hw + pixel) -> logit = act . w16 (channel order),
 *   prob (5, hw) = softmax(-logit) over d, offset (hw) = sum_d prob_d * (d - 2) * interval[0] (model.py:40-43,218-227).
 * pf_flow_head_backward_f32: gact (5*hw, 16) = d offset / d act * goffset, gw16 (16) (+)= its weight gradient;
 *   workspace: pf_flow_head_backward_workspace(hw) bytes.
 * pf_masked_mae_f32: loss[0] = weight * sum_b [ sum_{gt != 0} |pred - gt| / interval[b] / (count_b + 1e-7) ] with gt
 *   (B, H, W) read at the nearest-resized positions of pred (B, h, w) (networks.py:170-181, model.py:308-339);
 *   coef (B) = weight / (interval_b * (count_b + 1e-7)) for the backward: gpred = gloss[0] * coef_b * sign(pred - gt)
 *   on valid pixels, 0 elsewhere. */
int pf_softargmin_backward_f32(const float* cost, const float* params, const float* depth, const float* gdepth,
                               float* gcost, int64_t B, int64_t D, int64_t HW, void* stream);
int pf_flow_head_train_f32(const float* act, int64_t ld, const float* w16, const float* interval, int64_t hw,
                           float* offset, float* prob, void* stream);
int64_t pf_flow_head_backward_workspace(int64_t hw);
int pf_flow_head_backward_f32(const float* act, int64_t ld, const float* w16, const float* interval, const float* prob,
                              const float* goffset, int64_t hw, float* gact, float* gw16, int accumulate,
                              void* workspace, int64_t workspace_bytes, void* stream);
int pf_masked_mae_f32(const float* pred, const float* gt, const float* interval, int B, int h, int w, int H, int W,
                      float weight, float* loss, float* coef, void* stream);
int pf_masked_mae_backward_f32(const float* pred, const float* gt, const float* coef, const float* gloss, int B, int h,
                               int w, int H, int W, float* gpred, void* stream);
/* torch.optim.RMSprop's update (reference solver.py:17-52; no momentum, not centered) on flat float32 buffers of n
 * elements: g' = g + wd[i] * p (wd NULL: no weight decay); sq = alpha * sq + (1 - alpha) * g'^2;
 * p -= lr * g' / (sqrt(sq) + eps).  alpha is taken as the float nearest to the caller's double: pass (float)alpha. */
int pf_rmsprop_f32(float* p, const float* g, float* sq, const float* wd, int64_t n, float lr, float alpha, float eps,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTFLOW_HIP_H_ */
