// The step after the path: what reference utils/eval_file_logger.py:12-79 and
// tools/depthfusion.py:153-170 compute per depth map before files are written, on the device, so that the
// evaluation loop hands ONE staging buffer to an asynchronous D2H copy instead of stalling on .cpu().numpy()
// of every map (eval_file_logger.py:33-35,50,66).
//
//   eval_pack_map      a (h,w) map -> PFM row order (bottom row first: write_pfm's np.flipud, utils/io.py:124)
//   eval_flow_prob     (5,h,w) hypothesis probabilities -> the scalar confidence map of eval_file_logger.py:48-62:
//                      i = sum_d p_d*(d-2) + 2 in float64 (NumPy promotes: the offsets are a float64 array),
//                      conf = p[floor(i)] + p[min(floor(i)+1, 4)] in float32; PFM row order
//   eval_prob_filter   depth with pixels of low flow / initial confidence zeroed (depthfusion.py:166-167); the
//                      initial-confidence map may be smaller (nearest resize, cv2.INTER_NEAREST's index rule)
#include "pf_common.h"

namespace {

__global__ __launch_bounds__(256) void eval_pack_map_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int h, int w, int flip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  dst[(flip ? h - 1 - y : y) * w + x] = src[i];
}

__device__ __forceinline__ float flow_confidence(const float* __restrict__ prob, int i, int hw) {
  float p[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) p[d] = prob[(int64_t)d * hw + i];
  // np.sum(out_flow_prob_map * interval_list, axis=-1): float64 products, add.reduce = a0 + (((a1+a2)+a3)+a4)
  const double a0 = (double)p[0] * -2.0, a1 = (double)p[1] * -1.0, a2 = (double)p[2] * 0.0;
  const double a3 = (double)p[3] * 1.0, a4 = (double)p[4] * 2.0;
  const double idx = (a0 + (((a1 + a2) + a3) + a4)) + 2.0;
  int fl = (int)floor(idx);
  int ce = fl + 1;
  ce = ce < 0 ? 0 : (ce > 4 ? 4 : ce);                    // np.clip(pred_ceil, 0, 4)
  fl = fl < 0 ? fl + 5 : fl;                              // NumPy fancy indexing wraps a negative index
  fl = fl < 0 ? 0 : (fl > 4 ? 4 : fl);                    // (out of range would raise in NumPy; clamp, never fault)
  return p[fl] + p[ce];
}

__global__ __launch_bounds__(256) void eval_flow_prob_kernel(const float* __restrict__ prob, float* __restrict__ dst,
                                                             int h, int w, int flip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  dst[(flip ? h - 1 - y : y) * w + x] = flow_confidence(prob, i, h * w);
}

__global__ __launch_bounds__(256) void eval_prob_filter_kernel(const float* __restrict__ depth,
                                                               const float* __restrict__ flow_conf,
                                                               const float* __restrict__ init_conf, int h, int w,
                                                               int ih, int iw, float flow_thr, float init_thr,
                                                               float* __restrict__ dst, int flip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  // cv2.resize(..., INTER_NEAREST): src = min(floor(dst * src_size / dst_size), src_size - 1)
  int sy = y, sx = x;
  if (ih != h || iw != w) {
    sy = (int)floor((double)y * ((double)ih / (double)h));
    sx = (int)floor((double)x * ((double)iw / (double)w));
    sy = sy > ih - 1 ? ih - 1 : sy;
    sx = sx > iw - 1 ? iw - 1 : sx;
  }
  float d = depth[i];
  if (flow_conf[i] < flow_thr) d = 0.0f;
  if (init_conf[sy * iw + sx] < init_thr) d = 0.0f;
  dst[(flip ? h - 1 - y : y) * w + x] = d;
}

}  // namespace

extern "C" {

int pf_eval_pack_map_f32(const float* src, float* dst, int h, int w, int flip_rows, void* stream) {
  PF_REQUIRE(h >= 0 && w >= 0 && (int64_t)h * w <= INT32_MAX);
  if (h == 0 || w == 0) return PF_OK;
  PF_REQUIRE(src && dst && src != dst);
  hipLaunchKernelGGL(eval_pack_map_kernel, dim3((unsigned)pf_cdiv((int64_t)h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, h, w, flip_rows);
  return pf_launch_status();
}

int pf_eval_flow_prob_f32(const float* prob, float* dst, int h, int w, int flip_rows, void* stream) {
  PF_REQUIRE(h >= 0 && w >= 0 && (int64_t)h * w <= INT32_MAX);
  if (h == 0 || w == 0) return PF_OK;
  PF_REQUIRE(prob && dst);
  hipLaunchKernelGGL(eval_flow_prob_kernel, dim3((unsigned)pf_cdiv((int64_t)h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, prob, dst, h, w, flip_rows);
  return pf_launch_status();
}

int pf_eval_prob_filter_f32(const float* depth, const float* flow_conf, const float* init_conf, int h, int w, int ih,
                            int iw, float flow_threshold, float init_threshold, float* dst, int flip_rows,
                            void* stream) {
  PF_REQUIRE(h >= 0 && w >= 0 && ih >= 1 && iw >= 1 && (int64_t)h * w <= INT32_MAX);
  if (h == 0 || w == 0) return PF_OK;
  PF_REQUIRE(depth && flow_conf && init_conf && dst && dst != depth);
  hipLaunchKernelGGL(eval_prob_filter_kernel, dim3((unsigned)pf_cdiv((int64_t)h * w, 256)), dim3(256), 0,
                     (hipStream_t)stream, depth, flow_conf, init_conf, h, w, ih, iw, flow_threshold, init_threshold, dst,
                     flip_rows);
  return pf_launch_status();
}

}  // extern "C"
