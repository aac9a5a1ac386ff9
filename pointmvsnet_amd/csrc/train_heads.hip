// Row Z (training step, BASELINE config 4): the small differentiable heads around the convolution stacks, forward and
// backward, one or two launches each instead of the 15-25 element-wise ATen launches autograd makes of them:
//
//   soft-argmin backward   depth = sum_k z_k softmax(-cost)_k (reference model.py:117-124); the forward is row S's kernel
//                          (pf_softargmin_prob_f32, csrc/fetch.hip);  d depth / d cost_k = -p_k (z_k - depth).
//   flow head (training)   16 -> 1 convolution, softmax over the 5 hypotheses, expected offset (reference
//                          model.py:40-43, 218-227) and its backward (gradient rows of the MLP output + the 16 weights).
//   masked MAE loss        reference networks.py:170-181 (MAELoss) over the nearest-resized ground truth
//                          (model.py:308-339): sum_b [ sum valid |pred - gt| / interval_b / (count_b + 1e-7) ] * weight.
//
//   RMSprop                reference solver.py:17-52 on ONE flat parameter buffer: one launch for PyTorch's five
//                          multi-tensor launches over 115 tensors (the step's only work outside its hipGraph).
//
// All reductions are float64 in a fixed order (no atomics): the step stays bit-reproducible.  Every kernel here moves a
// few hundred KB: they are launch-latency items, written to be ONE dependency-chain link each.
#include "pf_common.h"

namespace {

__device__ __forceinline__ float linspace_at(float start, float end, float step, int k, int D) {
  // ATen linspace: start + step*k below the midpoint, end - step*(D-1-k) above it (as csrc/fetch.hip)
  return (k < D / 2) ? (start + step * (float)k) : (end - step * (float)(D - 1 - k));
}

// one thread per pixel: softmax(-cost) recomputed exactly as the forward kernel rounds it (max, sum of expf, quotient).
// DT = the number of depth planes at compile time (48: BASELINE configs 2 and 4; 96: config 3) or 0 (any D, below).
// With D a run-time bound the three passes were 3 x 48 loads one round trip after the other on 20 blocks of 256 pixels
// (40 us for 1 MB at config 4); with D known the column sits in registers after ONE batch of loads, and 64-pixel blocks
// put the 5 120 pixels on 80 CUs (round 6).  Same arithmetic in the same order.
template <int DT>
__global__ __launch_bounds__(64) void softargmin_bwd_fixed_kernel(const float* __restrict__ cost,
                                                                  const float* __restrict__ params,
                                                                  const float* __restrict__ depth,
                                                                  const float* __restrict__ gdepth,
                                                                  float* __restrict__ gcost, int64_t HW) {
  constexpr int D = DT;
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (i >= HW) return;
  const float start = params[b * 3 + 0], end = params[b * 3 + 1];
  const float step = (D > 1) ? (end - start) / (float)(D - 1) : 0.0f;
  const float* c = cost + b * D * HW + i;
  float* gc = gcost + b * D * HW + i;
  float cv[D];
#pragma unroll
  for (int k = 0; k < D; ++k) cv[k] = -c[(int64_t)k * HW];
  float mx = -__builtin_huge_valf();
#pragma unroll
  for (int k = 0; k < D; ++k) mx = fmaxf(mx, cv[k]);
  constexpr int dq = (D + 3) >> 2;
  float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < D; ++k) {
    cv[k] = expf(cv[k] - mx);
    part[k / dq] += cv[k];
  }
  const float den = ((part[0] + part[1]) + part[2]) + part[3];
  const float dep = depth[b * HW + i], g = gdepth[b * HW + i];
#pragma unroll
  for (int k = 0; k < D; ++k) gc[(int64_t)k * HW] = -g * (cv[k] / den) * (linspace_at(start, end, step, k, D) - dep);
}

__global__ __launch_bounds__(256) void softargmin_bwd_kernel(const float* __restrict__ cost,
                                                             const float* __restrict__ params,
                                                             const float* __restrict__ depth,
                                                             const float* __restrict__ gdepth,
                                                             float* __restrict__ gcost, int D, int64_t HW) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (i >= HW) return;
  const float start = params[b * 3 + 0], end = params[b * 3 + 1];
  const float step = (D > 1) ? (end - start) / (float)(D - 1) : 0.0f;
  const float* c = cost + b * D * HW + i;
  float* gc = gcost + b * D * HW + i;
  float mx = -__builtin_huge_valf();
  for (int k = 0; k < D; ++k) mx = fmaxf(mx, -c[(int64_t)k * HW]);
  // the forward sums four depth slices apart and adds them in slice order
  const int dq = (D + 3) >> 2;
  float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int q = 0; q < 4; ++q)
    for (int k = q * dq; k < min(D, (q + 1) * dq); ++k) part[q] += expf(-c[(int64_t)k * HW] - mx);
  const float den = ((part[0] + part[1]) + part[2]) + part[3];
  const float dep = depth[b * HW + i], g = gdepth[b * HW + i];
  for (int k = 0; k < D; ++k) {
    const float p = expf(-c[(int64_t)k * HW] - mx) / den;
    gc[(int64_t)k * HW] = -g * p * (linspace_at(start, end, step, k, D) - dep);
  }
}

// ---- flow head -----------------------------------------------------------------------------------------------
// act (5 * hw, ld) point-major rows, hypothesis-major (row = d * hw + pixel); logit_d = sum_c act[row][c] * w[c] in
// channel order; p = softmax(-logit) over d; offset = sum_d p_d * (d - 2) * interval.
__global__ __launch_bounds__(256) void flow_head_train_kernel(const float* __restrict__ act, int64_t ld,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ interval, int64_t hw,
                                                              float* __restrict__ offset, float* __restrict__ prob) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  float wv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) wv[c] = w[c];
  float lg[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const float4* row = reinterpret_cast<const float4*>(act + ((int64_t)d * hw + p) * ld);
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = row[q];
      s += v.x * wv[4 * q];
      s += v.y * wv[4 * q + 1];
      s += v.z * wv[4 * q + 2];
      s += v.w * wv[4 * q + 3];
    }
    lg[d] = -s;
  }
  float mx = lg[0];
#pragma unroll
  for (int d = 1; d < 5; ++d) mx = fmaxf(mx, lg[d]);
  float e[5], den = 0.0f;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    e[d] = expf(lg[d] - mx);
    den += e[d];
  }
  const float iv = interval[0];
  float off = 0.0f;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const float pd = e[d] / den;
    prob[(int64_t)d * hw + p] = pd;
    off += pd * ((float)(d - 2) * iv);
  }
  offset[p] = off;
}

// gradient rows of act and per-block float64 partials of the 16 weight gradients: partials (blocks, 16)
__global__ __launch_bounds__(256) void flow_head_bwd_kernel(const float* __restrict__ act, int64_t ld,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ interval,
                                                            const float* __restrict__ prob,
                                                            const float* __restrict__ goffset, int64_t hw,
                                                            float* __restrict__ gact, double* __restrict__ partials) {
  __shared__ double red[4][16];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float wv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) wv[c] = w[c];
  double gw[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) gw[c] = 0.0;
  if (p < hw) {
    const float iv = interval[0];
    float pd[5], off = 0.0f;
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      pd[d] = prob[(int64_t)d * hw + p];
      off += pd[d] * ((float)(d - 2) * iv);
    }
    const float g = goffset[p];
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      // offset = sum p_d len_d with p = softmax(-logit):  d offset / d logit_d = -p_d (len_d - offset)
      const float gl = -g * pd[d] * ((float)(d - 2) * iv - off);
      const float4* row = reinterpret_cast<const float4*>(act + ((int64_t)d * hw + p) * ld);
      float4* grow = reinterpret_cast<float4*>(gact + ((int64_t)d * hw + p) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = row[q];
        grow[q] = make_float4(gl * wv[4 * q], gl * wv[4 * q + 1], gl * wv[4 * q + 2], gl * wv[4 * q + 3]);
        gw[4 * q] += (double)(gl * v.x);
        gw[4 * q + 1] += (double)(gl * v.y);
        gw[4 * q + 2] += (double)(gl * v.z);
        gw[4 * q + 3] += (double)(gl * v.w);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    double v = gw[c];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16)
    partials[(int64_t)blockIdx.x * 16 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// 16 weights x 16 slices: slice sl adds blocks sl, sl + 16, ... and the slices are added in slice order (fixed order).
// (Round 6: it was 16 threads adding `blocks` dependent loads each -- 24 us for the 80 partial rows of a 20 480-pixel map.)
__global__ __launch_bounds__(256) void flow_head_wsum_kernel(const double* __restrict__ partials, int blocks,
                                                             float* __restrict__ gw, int accumulate) {
  __shared__ double red[16][16];
  const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
  double s = 0.0;
#pragma unroll 4
  for (int b = sl; b < blocks; b += 16) s += partials[(int64_t)b * 16 + c];
  red[sl][c] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) t += red[i][c];
    gw[c] = (accumulate ? gw[c] : 0.0f) + (float)t;
  }
}

// ---- masked MAE ----------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  // ATen upsample_nearest: min(floor(dst * (float)in / out), in - 1)
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

// ONE block: per sample the masked error sum and the count (float64, fixed order), then the loss and the backward's
// per-sample coefficient  coef_b = weight / (interval_b * (count_b + 1e-7)).
__global__ __launch_bounds__(1024) void masked_mae_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ interval, int B, int h, int w,
                                                          int H, int W, float weight, float* __restrict__ loss,
                                                          float* __restrict__ coef) {
  __shared__ double red[2][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  double total = 0.0;
  for (int b = 0; b < B; ++b) {
    double e = 0.0, c = 0.0;
    // four pixels per trip, both loads unconditional: the loop was 2 dependent round trips per pixel on ONE block
    // (19 us for a 20 480-pixel map); a thread still adds its pixels in ascending order
    for (int i0 = threadIdx.x; i0 < h * w; i0 += 4096) {
      float gv[4], pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        const int ii = i < h * w ? i : 0;
        const int y = ii / w, x = ii - y * w;
        gv[u] = gt[((int64_t)b * H + nearest_src(y, sy, H)) * W + nearest_src(x, sx, W)];
        pv[u] = pred[(int64_t)b * h * w + ii];
        if (i >= h * w) gv[u] = 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (gv[u] != 0.0f) {
          c += 1.0;
          e += (double)fabsf(pv[u] - gv[u]);
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      e += __shfl_down(e, o, 64);
      c += __shfl_down(c, o, 64);
    }
    __syncthreads();                       // the previous sample's read of red[][] is done
    if (lane == 0) {
      red[0][wave] = e;
      red[1][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double es = 0.0, cs = 0.0;
      for (int q = 0; q < 16; ++q) {
        es += red[0][q];
        cs += red[1][q];
      }
      const double cnt = cs + 1e-7;
      total += es / (double)interval[b] / cnt;
      coef[b] = (float)((double)weight / ((double)interval[b] * cnt));
    }
  }
  if (threadIdx.x == 0) loss[0] = (float)(total * (double)weight);
}

__global__ __launch_bounds__(256) void masked_mae_bwd_kernel(const float* __restrict__ pred,
                                                             const float* __restrict__ gt,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ gloss, int h, int w, int H,
                                                             int W, float* __restrict__ gpred) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= h * w) return;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  const int y = i / w, x = i - y * w;
  const float g = gt[((int64_t)b * H + nearest_src(y, sy, H)) * W + nearest_src(x, sx, W)];
  const float d = pred[(int64_t)b * h * w + i] - g;
  const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
  gpred[(int64_t)b * h * w + i] = g != 0.0f ? gloss[0] * coef[b] * sgn : 0.0f;
}

// ---- RMSprop on the flat parameter / gradient / square-average buffers ------------------------------------------------
// torch.optim.RMSprop (reference solver.py:17-52; the foreach form PyTorch runs on a GPU: five multi-tensor launches):
//   g' = g + wd * p;  sq = sq * alpha + (1 - alpha) * (g' * g');  p = p + (-lr) * (g' / (sqrt(sq) + eps)).
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ sq, const float* __restrict__ wd, int64_t n,
                                                      float neg_lr, float alpha, float one_minus_alpha, float eps) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 pv = *reinterpret_cast<float4*>(p + i), sv = *reinterpret_cast<float4*>(sq + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, sa[4] = {sv.x, sv.y, sv.z, sv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = wd != nullptr ? ga[j] + wd[i + j] * pa[j] : ga[j];
      sa[j] = sa[j] * alpha + one_minus_alpha * (gg * gg);
      pa[j] = pa[j] + neg_lr * (gg / (sqrtf(sa[j]) + eps));
    }
    *reinterpret_cast<float4*>(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(sq + i) = make_float4(sa[0], sa[1], sa[2], sa[3]);
  } else {
    for (int64_t e = i; e < n; ++e) {
      const float gg = wd != nullptr ? g[e] + wd[e] * p[e] : g[e];
      const float s = sq[e] * alpha + one_minus_alpha * (gg * gg);
      sq[e] = s;
      p[e] = p[e] + neg_lr * (gg / (sqrtf(s) + eps));
    }
  }
}

}  // namespace

extern "C" {

int pf_rmsprop_f32(float* p, const float* g, float* sq, const float* wd, int64_t n, float lr, float alpha, float eps,
                   void* stream) {
  PF_REQUIRE(n >= 0 && lr >= 0.0f && alpha >= 0.0f && eps >= 0.0f);
  if (n == 0) return PF_OK;
  PF_REQUIRE(p && g && sq && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)sq)) & 15) == 0);
  const float one_minus_alpha = (float)(1.0 - (double)alpha);
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)pf_cdiv(pf_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, p, g,
                     sq, wd, n, -lr, alpha, one_minus_alpha, eps);
  return pf_launch_status();
}

int pf_softargmin_backward_f32(const float* cost, const float* params, const float* depth, const float* gdepth,
                               float* gcost, int64_t B, int64_t D, int64_t HW, void* stream) {
  PF_REQUIRE(B >= 0 && D >= 1 && HW >= 0 && B <= 65535 && D <= INT32_MAX);
  if (B == 0 || HW == 0) return PF_OK;
  PF_REQUIRE(cost && params && depth && gdepth && gcost);
  if (D == 48 || D == 96) {
    dim3 grid64((unsigned)pf_cdiv(HW, 64), (unsigned)B);
    if (D == 48)
      hipLaunchKernelGGL(softargmin_bwd_fixed_kernel<48>, grid64, dim3(64), 0, (hipStream_t)stream, cost, params, depth,
                         gdepth, gcost, HW);
    else
      hipLaunchKernelGGL(softargmin_bwd_fixed_kernel<96>, grid64, dim3(64), 0, (hipStream_t)stream, cost, params, depth,
                         gdepth, gcost, HW);
    return pf_launch_status();
  }
  dim3 grid((unsigned)pf_cdiv(HW, 256), (unsigned)B);
  hipLaunchKernelGGL(softargmin_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, cost, params, depth, gdepth,
                     gcost, (int)D, HW);
  return pf_launch_status();
}

int pf_flow_head_train_f32(const float* act, int64_t ld, const float* w16, const float* interval, int64_t hw,
                           float* offset, float* prob, void* stream) {
  PF_REQUIRE(hw >= 0 && ld >= 16 && (ld & 3) == 0);
  if (hw == 0) return PF_OK;
  PF_REQUIRE(act && w16 && interval && offset && prob && (((uintptr_t)act) & 15) == 0);
  hipLaunchKernelGGL(flow_head_train_kernel, dim3((unsigned)pf_cdiv(hw, 256)), dim3(256), 0, (hipStream_t)stream, act,
                     ld, w16, interval, hw, offset, prob);
  return pf_launch_status();
}

int64_t pf_flow_head_backward_workspace(int64_t hw) { return hw < 0 ? -1 : 8 * 16 * pf_cdiv(hw > 0 ? hw : 1, 256); }

int pf_flow_head_backward_f32(const float* act, int64_t ld, const float* w16, const float* interval, const float* prob,
                              const float* goffset, int64_t hw, float* gact, float* gw16, int accumulate,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  PF_REQUIRE(hw >= 1 && ld >= 16 && (ld & 3) == 0 && act && w16 && interval && prob && goffset && gact && gw16);
  PF_REQUIRE(((((uintptr_t)act) | ((uintptr_t)gact)) & 15) == 0);
  const int64_t blocks = pf_cdiv(hw, 256);
  PF_REQUIRE(workspace != nullptr && workspace_bytes >= 8 * 16 * blocks && blocks <= INT32_MAX);
  double* partials = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(flow_head_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, act, ld, w16,
                     interval, prob, goffset, hw, gact, partials);
  int rc = pf_launch_status();
  if (rc != PF_OK) return rc;
  hipLaunchKernelGGL(flow_head_wsum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)blocks, gw16,
                     accumulate);
  return pf_launch_status();
}

int pf_masked_mae_f32(const float* pred, const float* gt, const float* interval, int B, int h, int w, int H, int W,
                      float weight, float* loss, float* coef, void* stream) {
  PF_REQUIRE(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && (int64_t)h * w <= INT32_MAX);
  PF_REQUIRE(pred && gt && interval && loss && coef);
  hipLaunchKernelGGL(masked_mae_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, gt, interval, B, h, w, H, W,
                     weight, loss, coef);
  return pf_launch_status();
}

int pf_masked_mae_backward_f32(const float* pred, const float* gt, const float* coef, const float* gloss, int B, int h,
                               int w, int H, int W, float* gpred, void* stream) {
  PF_REQUIRE(B >= 1 && B <= 65535 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && (int64_t)h * w <= INT32_MAX);
  PF_REQUIRE(pred && gt && coef && gloss && gpred);
  dim3 grid((unsigned)pf_cdiv((int64_t)h * w, 256), (unsigned)B);
  hipLaunchKernelGGL(masked_mae_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, pred, gt, coef, gloss, h, w, H, W,
                     gpred);
  return pf_launch_status();
}

}  // extern "C"
