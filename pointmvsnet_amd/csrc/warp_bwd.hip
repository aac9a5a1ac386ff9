// Row Z (training step): the BACKWARD of the fused warp + variance stages -- the coarse cost volume (reference
// model.py:79-111) and the flow feature assembly of a PointFlow iteration (model.py:153-204) -- without float atomics.
//
// The reference's autograd reaches grid_sample's backward (a scatter-add with float atomics in arrival order,
// utils/feature_fetcher.py:55) once per (hypothesis, pyramid level) -- 31 launches per step here until round 3
// (pf_fetch_backward_f32: 4.3 ms per step, and the reason the step's gradient was not bit-reproducible) -- plus ~40
// element-wise ATen passes over (V, C, N) tensors for the variance.  Here, for a stage with cost
//     var[c][n] = mean_v f_v[c][n]^2 - (mean_v f_v[c][n])^2,   f_v = bilinear fetch of view v's map at point n,
// the gradient w.r.t. the maps is computed in four kernels, gradients never flowing into the sampling positions (the
// reference builds the grid under no_grad, utils/feature_fetcher.py:29):
//   taps     one thread per (view, point): project, keep the fractional offsets and the KEY of the pair = the texel
//            cell (yi, xi) of its north-west tap, yi in [-1, H-1], xi in [-1, W-1] (no key when no tap is inside);
//   sort     pf_sort_pairs_by_key (csrc/knn_inverse.hip): the pairs grouped by (view, cell), ascending point index
//            inside a group -- a counting sort in kernel launches only;
//   vgrad    point-parallel: re-fetch f_v, gval[v][n][c] = (2 / V) * dvar[n][c] * (f_v - mean_v f) -- point-major rows;
//   gather   texel-parallel: texel (y, x) of view v is the nw / ne / sw / se tap of the pairs in cells (y, x),
//            (y, x-1), (y-1, x), (y-1, x-1); it adds weight * gval over those four lists, in list order, with plain
//            stores: bit-reproducible, and the output needs no zero-fill.
// The flow stage adds: the gradient w.r.t. the prior depth map through the xyz features (model.py:178-194: world is
// linear in depth), and the adjoint of the bilinear resize of the three pyramid levels (model.py:184) as a gather.
// All maps here are channel-last (V, H, W, C) as the forward kernels of csrc/fetch.hip read them.
// Bound: L2 -> L1 row traffic (every (view, point) row of gval is read by four texels), like the forward warp.
#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WbPoints {
  int mode;                 // 0: flow -- pixel grid (H, W) x 5 hypotheses around `depth`; 1: frustum of D depth planes
  int H, W, D;
  const float* depth;       // flow: (H, W) prior depth at the flow resolution
  const float* interval;    // flow: device pointer to the hypothesis spacing
  const float* cam;         // flow: packed camera block (PF_CAM_*)
  const float* kinv;        // frustum: 9, 9, 3 floats and D depths (one scene)
  const float* rinv;
  const float* t;
  const float* depths;
  const float* K;           // frustum: (V, 9) intrinsics, (V, 12) extrinsics
  const float* E;
};

// world point n (n = d * H * W + y * W + x), the arithmetic of the forward kernels (fetch.hip: flow_features_hyp_kernel's
// `world` and fetch_variance_kernel<.., FRUSTUM>) so that the taps are the forward's taps bit for bit
__device__ __forceinline__ void wb_world(const WbPoints& P, int64_t n, float& X, float& Y, float& Z) {
  const int hw = P.H * P.W;
  const int d = (int)(n / hw);
  const int pix = (int)(n - (int64_t)d * hw);
  const int y = pix / P.W, x = pix - y * P.W;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float* Ki = P.mode == 0 ? P.cam + PF_CAM_KREF_INV : P.kinv;
  const float* Ri = P.mode == 0 ? P.cam + PF_CAM_RREF_INV : P.rinv;
  const float* t0 = P.mode == 0 ? P.cam + PF_CAM_TREF : P.t;
  const float u0 = fmaf(Ki[2], 1.0f, fmaf(Ki[1], py, Ki[0] * px));
  const float u1 = fmaf(Ki[5], 1.0f, fmaf(Ki[4], py, Ki[3] * px));
  const float u2 = fmaf(Ki[8], 1.0f, fmaf(Ki[7], py, Ki[6] * px));
  const float depth = P.mode == 0 ? P.depth[pix] + P.interval[0] * (float)(d - 2) : P.depths[d];
  const float q0 = u0 * depth - t0[0], q1 = u1 * depth - t0[1], q2 = u2 * depth - t0[2];
  X = fmaf(Ri[2], q2, fmaf(Ri[1], q1, Ri[0] * q0));
  Y = fmaf(Ri[5], q2, fmaf(Ri[4], q1, Ri[3] * q0));
  Z = fmaf(Ri[8], q2, fmaf(Ri[7], q1, Ri[6] * q0));
}

constexpr uint32_t kNoKey = 0xffffffffu;

// grid = (ceil(N / 256), V): keys (V, N), fxy (V, N, 2)
__global__ __launch_bounds__(256) void warp_taps_kernel(WbPoints P, int64_t N, int skip_view0, uint32_t* __restrict__ keys,
                                                        float2* __restrict__ fxy) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int v = blockIdx.y;
  if (n >= N) return;
  const int64_t p = (int64_t)v * N + n;
  if (skip_view0 && v == 0) {
    keys[p] = kNoKey;
    fxy[p] = make_float2(0.0f, 0.0f);
    return;
  }
  float X, Y, Z;
  wb_world(P, n, X, Y, Z);
  const float* Kv = P.mode == 0 ? P.cam + PF_CAM_VIEWS + v * PF_CAM_VIEW_STRIDE : P.K + v * 9;
  const float* Ev = P.mode == 0 ? Kv + 9 : P.E + v * 12;
  PfTaps t;
  pf_project_taps(X, Y, Z, Kv, Ev, P.H, P.W, t);
  const bool any = t.ok[0] || t.ok[1] || t.ok[2] || t.ok[3];
  const uint32_t cells = (uint32_t)(P.H + 1) * (uint32_t)(P.W + 1);
  keys[p] = any ? (uint32_t)v * cells + (uint32_t)(t.yi + 1) * (uint32_t)(P.W + 1) + (uint32_t)(t.xi + 1) : kNoKey;
  fxy[p] = make_float2(t.fx, t.fy);
}

// the taps of a pair from its key and fractional offsets: offsets (texel index y * W + x, 0 when outside) and weights,
// exactly pf_project_taps' (pf_common.h)
__device__ __forceinline__ void wb_decode(uint32_t key, float2 f, int v, int H, int W, int off[4], float wgt[4]) {
  if (key == kNoKey) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      off[k] = 0;
      wgt[k] = 0.0f;
    }
    return;
  }
  const uint32_t cell = key - (uint32_t)v * (uint32_t)(H + 1) * (uint32_t)(W + 1);
  const int yi = (int)(cell / (uint32_t)(W + 1)) - 1, xi = (int)(cell % (uint32_t)(W + 1)) - 1;
  const bool vx0 = xi >= 0, vx1 = xi + 1 <= W - 1, vy0 = yi >= 0, vy1 = yi + 1 <= H - 1;
  const float wx1 = f.x, wy1 = f.y, wx0 = 1.0f - f.x, wy0 = 1.0f - f.y;
  const bool ok[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
  off[0] = ok[0] ? yi * W + xi : 0;
  off[1] = ok[1] ? yi * W + xi + 1 : 0;
  off[2] = ok[2] ? (yi + 1) * W + xi : 0;
  off[3] = ok[3] ? (yi + 1) * W + xi + 1 : 0;
  wgt[0] = ok[0] ? wy0 * wx0 : 0.0f;
  wgt[1] = ok[1] ? wy0 * wx1 : 0.0f;
  wgt[2] = ok[2] ? wy1 * wx0 : 0.0f;
  wgt[3] = ok[3] ? wy1 * wx1 : 0.0f;
}

struct WbLevels {
  const float* maps[3];     // channel-last (V, H, W, c_l)
  int c[3];
  int ctot;                 // c[0] + c[1] + c[2]
};

// thread = (point n, channel quad q of the ctot concatenated channels); dvar rows (N, ldv) point-major (the first ctot
// columns); gval (V, N, ctot).  ref_override: view 0 contributes its un-warped map (reference model.py:103-106).
template <int V>
__global__ __launch_bounds__(256) void variance_grad_kernel(WbLevels L, int H, int W, int64_t N,
                                                            const uint32_t* __restrict__ keys,
                                                            const float2* __restrict__ fxy,
                                                            const float* __restrict__ dvar, int64_t ldv,
                                                            int ref_override, float* __restrict__ gval) {
  const int Q = L.ctot >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = i / Q;
  const int q = (int)(i - n * Q);
  if (n >= N) return;
  int c = 4 * q, level = 0;
  if (c >= L.c[0]) {
    c -= L.c[0];
    level = 1;
    if (c >= L.c[1]) {
      c -= L.c[1];
      level = 2;
    }
  }
  const int cl = L.c[level];
  const float* maps = L.maps[level];
  const int64_t hw = (int64_t)H * W;
  f32x4 f[V];
  f32x4 s = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float* mv = maps + (int64_t)v * hw * cl + c;
    if (v == 0 && ref_override) {
      f[v] = *reinterpret_cast<const f32x4*>(mv + (n % hw) * cl);
    } else {
      int off[4];
      float wg[4];
      wb_decode(keys[(int64_t)v * N + n], fxy[(int64_t)v * N + n], v, H, W, off, wg);
      const f32x4 a = *reinterpret_cast<const f32x4*>(mv + (int64_t)off[0] * cl);
      const f32x4 b = *reinterpret_cast<const f32x4*>(mv + (int64_t)off[1] * cl);
      const f32x4 cc = *reinterpret_cast<const f32x4*>(mv + (int64_t)off[2] * cl);
      const f32x4 dd = *reinterpret_cast<const f32x4*>(mv + (int64_t)off[3] * cl);
#pragma unroll
      for (int j = 0; j < 4; ++j) f[v][j] = fmaf(dd[j], wg[3], fmaf(cc[j], wg[2], fmaf(b[j], wg[1], a[j] * wg[0])));
    }
    s += f[v];
  }
  f32x4 g;
  if (ldv > 0) {
    g = *reinterpret_cast<const f32x4*>(dvar + n * ldv + 4 * q);
  } else {                          // channel-major (ctot, N): the cost volume's own layout, no transposed copy
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = dvar[(int64_t)(4 * q + j) * N + n];
  }
  const float inv_v = 1.0f / (float)V;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (2.0f * inv_v) * g[j] * (f[v][j] - s[j] * inv_v);
    *reinterpret_cast<f32x4*>(gval + ((int64_t)v * N + n) * L.ctot + 4 * q) = o;
  }
}

// thread = (texel, channel quad); grid.y = views v0 .. V-1.  dmaps (V, H, W, ctot) channel-last.
__global__ __launch_bounds__(256) void warp_gather_kernel(const float* __restrict__ gval, const float2* __restrict__ fxy,
                                                          const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ start, int H, int W, int ctot,
                                                          int v0, float* __restrict__ dmaps) {
  const int Q = ctot >> 2;
  const int v = v0 + blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t texel = i / Q;
  const int q = (int)(i - texel * Q);
  if (texel >= (int64_t)H * W) return;
  const int y = (int)(texel / W), x = (int)(texel - (int64_t)y * W);
  const uint32_t cells = (uint32_t)(H + 1) * (uint32_t)(W + 1);
  f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {                 // this texel as the nw / ne / sw / se tap of a pair
    const int yi = y - (k >> 1), xi = x - (k & 1);
    const uint32_t key = (uint32_t)v * cells + (uint32_t)(yi + 1) * (uint32_t)(W + 1) + (uint32_t)(xi + 1);
    const uint32_t t0 = start[key], t1 = start[key + 1];
    // 8 pairs per trip: their ids in one batch of loads, then their offsets and rows in another -- two dependent round
    // trips per 8 pairs (round 4: two per PAIR; a texel of the coarse volume sums ~4 x 48 pairs): 133 -> 74 us there.
    // Same summation order.  (Measured and not kept: a lane per cell list, the four sums joined by shuffles -- 98 us; the
    // first 4 pairs of all four lists fetched together -- the flow stage 85 -> 117 us at 142 VGPRs.)
    for (uint32_t e = t0; e < t1; e += 8) {
      uint32_t p[8];
      float2 f[8];
      f32x4 g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = order[e + u < t1 ? e + u : t1 - 1];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f[u] = fxy[p[u]];
        g[u] = *reinterpret_cast<const f32x4*>(gval + (int64_t)p[u] * ctot + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (e + u < t1) {
          const float wy = (k >> 1) ? f[u].y : 1.0f - f[u].y;
          const float wx = (k & 1) ? f[u].x : 1.0f - f[u].x;
          const float wgt = wy * wx;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(g[u][j], wgt, acc[j]);
        }
      }
    }
  }
  *reinterpret_cast<f32x4*>(dmaps + ((int64_t)v * H * W + texel) * ctot + 4 * q) = acc;
}

// d(prior depth)[y][x] = sum over the 5 hypotheses of  sum_axis dxyz[axis] / std[axis] * (Rinv Kinv (x+.5, y+.5, 1))[axis],
// dxyz[axis] = the 8 repeats of the xyz feature (columns c0 + 3 r + axis of the point's feature row; model.py:193-194)
__global__ __launch_bounds__(256) void flow_depth_grad_kernel(const float* __restrict__ dfeat, int64_t ld, int c0,
                                                              const float* __restrict__ cam, int H, int W,
                                                              float* __restrict__ ddepth) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int y = pix / W, x = pix - y * W;
  const float* Ki = cam + PF_CAM_KREF_INV;
  const float* Ri = cam + PF_CAM_RREF_INV;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float u0 = fmaf(Ki[2], 1.0f, fmaf(Ki[1], py, Ki[0] * px));
  const float u1 = fmaf(Ki[5], 1.0f, fmaf(Ki[4], py, Ki[3] * px));
  const float u2 = fmaf(Ki[8], 1.0f, fmaf(Ki[7], py, Ki[6] * px));
  float dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
    dir[a] = fmaf(Ri[3 * a + 2], u2, fmaf(Ri[3 * a + 1], u1, Ri[3 * a] * u0)) / cam[PF_CAM_STD + a];
  float acc = 0.0f;
  for (int d = 0; d < 5; ++d) {
    const float* row = dfeat + ((int64_t)d * H * W + pix) * ld + c0;
    float g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 24; ++j) g[j % 3] += row[j];
    acc += fmaf(g[2], dir[2], fmaf(g[1], dir[1], g[0] * dir[0]));
  }
  ddepth[pix] = acc;
}

// bilinear resize, align_corners = False (fetch.hip: resize_axis)
__device__ __forceinline__ void wb_resize_axis(int o, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}

// Adjoint of pf_flow_pyramid_f32 for one level: dres (V, OH, OW, ld) channel-last, columns [c0, c0 + C) -> dlevel
// (V, C, IH, IW) planar.  A source texel gathers the output pixels that interpolate from it, in (oy, ox) order.
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dres, int ld, int c0, int C, int OH,
                                                         int OW, int IH, int IW, float* __restrict__ dlevel) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, v = blockIdx.z;
  if (i >= IH * IW) return;
  const int iy = i / IW, ix = i - iy * IW;
  const float* src = dres + (int64_t)v * OH * OW * ld + c0 + c;
  float acc = 0.0f;
  if (IH == OH && IW == OW) {
    acc = src[(int64_t)i * ld];
  } else {
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo;
    ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi;
    ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      wb_resize_axis(oy, sy, IH, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.0f) + (y1 == iy ? ly1 : 0.0f);
      if (wy == 0.0f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        wb_resize_axis(ox, sx, IW, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.0f) + (x1 == ix ? lx1 : 0.0f);
        if (wx != 0.0f) acc = fmaf(src[((int64_t)oy * OW + ox) * ld], wy * wx, acc);
      }
    }
  }
  dlevel[(((int64_t)v * C + c) * IH + iy) * IW + ix] = acc;
}

// The same adjoint with thread = (source texel, channel QUAD): the channel-last rows of dres are read as 16-byte pieces,
// the Q lanes of a texel one contiguous run (the kernel above reads ONE float per lane out of every 4 * ld-byte row and
// repeats the window arithmetic per channel: 64 us for the 64-channel level of the 102 400-point iteration).  Same sums
// in the same (oy, ox) order: the same bits.  Needs C, c0 and ld multiples of 4.
__global__ __launch_bounds__(256) void resize_bwd_quad_kernel(const float* __restrict__ dres, int ld, int c0, int C, int OH,
                                                              int OW, int IH, int IW, float* __restrict__ dlevel) {
  const int Q = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t texel = i / Q;
  const int q = (int)(i - texel * Q);
  const int v = blockIdx.y;
  if (texel >= (int64_t)IH * IW) return;
  const int iy = (int)(texel / IW), ix = (int)(texel - (int64_t)iy * IW);
  const float* src = dres + (int64_t)v * OH * OW * ld + c0 + 4 * q;
  f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  if (IH == OH && IW == OW) {
    acc = *reinterpret_cast<const f32x4*>(src + texel * ld);
  } else {
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sy - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sx - 0.5f) + 1;
    oy_lo = oy_lo < 0 ? 0 : oy_lo;
    ox_lo = ox_lo < 0 ? 0 : ox_lo;
    oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi;
    ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      wb_resize_axis(oy, sy, IH, y0, y1, ly0, ly1);
      const float wy = (y0 == iy ? ly0 : 0.0f) + (y1 == iy ? ly1 : 0.0f);
      if (wy == 0.0f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        wb_resize_axis(ox, sx, IW, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.0f) + (x1 == ix ? lx1 : 0.0f);
        if (wx != 0.0f) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(src + ((int64_t)oy * OW + ox) * ld);
          const float w = wy * wx;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(g[j], w, acc[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) dlevel[(((int64_t)v * C + 4 * q + j) * IH + iy) * IW + ix] = acc[j];
}

template <int V>
void launch_vgrad(const WbLevels& L, int H, int W, int64_t N, const uint32_t* keys, const float2* fxy, const float* dvar,
                  int64_t ldv, int ref_override, float* gval, hipStream_t s) {
  const int64_t items = N * (L.ctot >> 2);
  hipLaunchKernelGGL((variance_grad_kernel<V>), dim3((unsigned)pf_cdiv(items, 256)), dim3(256), 0, s, L, H, W, N, keys,
                     fxy, dvar, ldv, ref_override, gval);
}

}  // namespace

extern "C" {

int pf_warp_taps_flow_f32(const float* depth, const float* interval, const float* cam, int V, int H, int W,
                          uint32_t* keys, float* fxy, void* stream) {
  PF_REQUIRE(V >= 1 && V <= PF_MAX_VIEWS && H >= 1 && W >= 1 && depth && interval && cam && keys && fxy);
  PF_REQUIRE((int64_t)V * (H + 1) * (W + 1) < ((int64_t)1 << 32) - 2);
  WbPoints P = {};
  P.mode = 0;
  P.H = H;
  P.W = W;
  P.D = 5;
  P.depth = depth;
  P.interval = interval;
  P.cam = cam;
  const int64_t N = (int64_t)5 * H * W;
  hipLaunchKernelGGL(warp_taps_kernel, dim3((unsigned)pf_cdiv(N, 256), (unsigned)V), dim3(256), 0, (hipStream_t)stream, P,
                     N, 0, keys, reinterpret_cast<float2*>(fxy));
  return pf_launch_status();
}

int pf_warp_taps_frustum_f32(const float* kinv, const float* rinv, const float* t, const float* depths, const float* K,
                             const float* E, int V, int H, int W, int D, int skip_view0, uint32_t* keys, float* fxy,
                             void* stream) {
  PF_REQUIRE(V >= 1 && V <= PF_MAX_VIEWS && H >= 1 && W >= 1 && D >= 1 && kinv && rinv && t && depths && K && E && keys && fxy);
  PF_REQUIRE((int64_t)V * (H + 1) * (W + 1) < ((int64_t)1 << 32) - 2);
  WbPoints P = {};
  P.mode = 1;
  P.H = H;
  P.W = W;
  P.D = D;
  P.kinv = kinv;
  P.rinv = rinv;
  P.t = t;
  P.depths = depths;
  P.K = K;
  P.E = E;
  const int64_t N = (int64_t)D * H * W;
  hipLaunchKernelGGL(warp_taps_kernel, dim3((unsigned)pf_cdiv(N, 256), (unsigned)V), dim3(256), 0, (hipStream_t)stream, P,
                     N, skip_view0, keys, reinterpret_cast<float2*>(fxy));
  return pf_launch_status();
}

int pf_variance_grad_f32(const float* maps1, int c1, const float* maps2, int c2, const float* maps3, int c3, int V, int H,
                         int W, int64_t N, const uint32_t* keys, const float* fxy, const float* dvar, int64_t ldv,
                         int ref_override, float* gval, void* stream) {
  PF_REQUIRE(V >= 1 && V <= PF_MAX_VIEWS && H >= 1 && W >= 1 && N >= 1 && c1 >= 4 && c2 >= 0 && c3 >= 0);
  if ((c1 & 3) || (c2 & 3) || (c3 & 3) || (ldv > 0 && (ldv & 3))) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(maps1 && (c2 == 0 || maps2) && (c3 == 0 || maps3) && keys && fxy && dvar && gval);
  PF_REQUIRE(ldv == -1 || ldv >= c1 + c2 + c3);                    // -1: dvar is channel-major (ctot, N)
  PF_REQUIRE(!ref_override || (N % ((int64_t)H * W)) == 0);
  WbLevels L;
  L.maps[0] = maps1;
  L.maps[1] = maps2;
  L.maps[2] = maps3;
  L.c[0] = c1;
  L.c[1] = c2;
  L.c[2] = c3;
  L.ctot = c1 + c2 + c3;
  hipStream_t s = (hipStream_t)stream;
  const float2* f2 = reinterpret_cast<const float2*>(fxy);
  switch (V) {
    case 1: launch_vgrad<1>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 2: launch_vgrad<2>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 3: launch_vgrad<3>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 4: launch_vgrad<4>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 5: launch_vgrad<5>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 6: launch_vgrad<6>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    case 7: launch_vgrad<7>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
    default: launch_vgrad<8>(L, H, W, N, keys, f2, dvar, ldv, ref_override, gval, s); break;
  }
  return pf_launch_status();
}

int pf_warp_gather_f32(const float* gval, const float* fxy, const uint32_t* order, const uint32_t* start, int V, int v0,
                       int H, int W, int ctot, float* dmaps, void* stream) {
  PF_REQUIRE(V >= 1 && v0 >= 0 && v0 <= V && H >= 1 && W >= 1 && ctot >= 4 && gval && fxy && order && start && dmaps);
  if (ctot & 3) return PF_ERR_UNSUPPORTED;
  if (v0 == V) return PF_OK;
  const int64_t items = (int64_t)H * W * (ctot >> 2);
  hipLaunchKernelGGL(warp_gather_kernel, dim3((unsigned)pf_cdiv(items, 256), (unsigned)(V - v0)), dim3(256), 0,
                     (hipStream_t)stream, gval, reinterpret_cast<const float2*>(fxy), order, start, H, W, ctot, v0, dmaps);
  return pf_launch_status();
}

int pf_flow_depth_grad_f32(const float* dfeat, int64_t ld, int c0, const float* cam, int H, int W, float* ddepth,
                           void* stream) {
  PF_REQUIRE(H >= 1 && W >= 1 && c0 >= 0 && ld >= c0 + 24 && dfeat && cam && ddepth);
  hipLaunchKernelGGL(flow_depth_grad_kernel, dim3((unsigned)pf_cdiv((int64_t)H * W, 256)), dim3(256), 0,
                     (hipStream_t)stream, dfeat, ld, c0, cam, H, W, ddepth);
  return pf_launch_status();
}

int pf_resize_bilinear_backward_f32(const float* dres, int ld, int c0, int C, int V, int OH, int OW, int IH, int IW,
                                    float* dlevel, void* stream) {
  PF_REQUIRE(V >= 1 && C >= 1 && c0 >= 0 && ld >= c0 + C && OH >= 1 && OW >= 1 && IH >= 1 && IW >= 1 && dres && dlevel);
  PF_REQUIRE(C <= 65535 && V <= 65535);
  if ((C & 3) == 0 && (c0 & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)dres & 15) == 0)
    hipLaunchKernelGGL(resize_bwd_quad_kernel, dim3((unsigned)pf_cdiv((int64_t)IH * IW * (C >> 2), 256), (unsigned)V),
                       dim3(256), 0, (hipStream_t)stream, dres, ld, c0, C, OH, OW, IH, IW, dlevel);
  else
    hipLaunchKernelGGL(resize_bwd_kernel, dim3((unsigned)pf_cdiv((int64_t)IH * IW, 256), (unsigned)C, (unsigned)V),
                       dim3(256), 0, (hipStream_t)stream, dres, ld, c0, C, OH, OW, IH, IW, dlevel);
  return pf_launch_status();
}

}  // extern "C"
