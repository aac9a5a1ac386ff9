// Rows W, V, F (+U): the multi-view feature warp and its fused forms.
//
// Mapping: one lane == one world point, the wave walks the channels.  Consecutive lanes are consecutive
// pixels of the reference grid, whose projections into a source view are neighbouring texels, so every
// per-channel tap load of a wave falls in one or two 256-byte row segments of the NCHW map (coalesced
// HBM/L2 reads without re-laying the maps out), and every per-channel store is one contiguous 256-byte
// segment of the (C,N) output.  Projection / tap arithmetic lives in pf_common.h (pf_project_taps).
//
// Algorithmic HBM bytes:
//   fetch_variance : V*C*H*W*4 (maps, read once) + 3*N*4 (points) + C*N*4 (cost volume written once)
//   flow_features  : V*(c1+c2+c3)*h*w*4 + 136*N*4 + 3*N*4 + h*w*4
// The reference moves V*C*N*4 for the fetched features alone and then makes four more passes over it.
#include <type_traits>

#include <stdlib.h>

#include "pf_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// W: (B,V,C,N) fetch, forward and backward (reference utils/feature_fetcher.py:13-60)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fetch_fwd_kernel(const float* __restrict__ maps,
                                                        const float* __restrict__ pts,
                                                        const float* __restrict__ Kmat,
                                                        const float* __restrict__ Emat, float* __restrict__ out,
                                                        int V, int C, int H, int W, int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  const int64_t b = blockIdx.z;
  if (n >= N) return;
  const float* pb = pts + b * 3 * N;
  PfTaps t;
  pf_project_taps(pb[n], pb[N + n], pb[2 * N + n], Kmat + (b * V + v) * 9,
                  Emat ? Emat + (b * V + v) * 12 : nullptr, H, W, t);
  const int64_t HW = (int64_t)H * W;
  const float* m = maps + (b * V + v) * C * HW;
  float* o = out + (b * V + v) * C * N + n;
  for (int c = 0; c < C; ++c) o[(int64_t)c * N] = pf_sample(m + (int64_t)c * HW, t);
}

__global__ __launch_bounds__(256) void fetch_bwd_kernel(const float* __restrict__ gout,
                                                        const float* __restrict__ pts,
                                                        const float* __restrict__ Kmat,
                                                        const float* __restrict__ Emat,
                                                        float* __restrict__ gmaps, int V, int C, int H, int W,
                                                        int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  const int64_t b = blockIdx.z;
  if (n >= N) return;
  const float* pb = pts + b * 3 * N;
  PfTaps t;
  pf_project_taps(pb[n], pb[N + n], pb[2 * N + n], Kmat + (b * V + v) * 9,
                  Emat ? Emat + (b * V + v) * 12 : nullptr, H, W, t);
  const int64_t HW = (int64_t)H * W;
  float* gm = gmaps + (b * V + v) * C * HW;
  const float* g = gout + (b * V + v) * C * N + n;
  for (int c = 0; c < C; ++c) {
    const float go = g[(int64_t)c * N];
    float* plane = gm + (int64_t)c * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t.ok[k]) unsafeAtomicAdd(plane + t.off[k], go * t.wgt[k]);
  }
}

// ------------------------------------------------------------------------------------------------
// W+V: fetch + variance over views (reference model.py:102-111 coarse, :187-190 flow)
// ------------------------------------------------------------------------------------------------
// The three coarse fetch+variance kernels share these two expressions (their outputs are bit-identical to each
// other, tests/test_gpu_ops.py): the bilinear sample as one fmaf chain over the taps in (nw, ne, sw, se) order and the
// variance E[x^2] - E[x]^2 with the means taken by a multiplication with 1/V.  Round 3: through round 2 they were
// separate multiplies / adds and two IEEE divisions per channel -- 22 instead of 3 instructions per variance, a third
// of the warp kernel's arithmetic; against the reference's own composition the fused result moves by float32
// rounding either way (tolerance in tests/test_gpu_ops.py::test_fetch_variance_vs_oracle).
__device__ __forceinline__ float pf_bilerp(float a, float b, float c, float d, float w0, float w1, float w2, float w3) {
  return fmaf(d, w3, fmaf(c, w2, fmaf(b, w1, a * w0)));
}
__device__ __forceinline__ float pf_variance(float s, float s2, float inv_v) {
  const float m1 = s * inv_v;
  return fmaf(-m1, m1, s2 * inv_v);
}

// Frustum of the reference view (reference model.py:79-100): point n = d*H*W + y*W + x is the un-projection
// of pixel centre (x+0.5, y+0.5) at depth hypothesis d,  world = Rinv (depth * Kinv (x+0.5, y+0.5, 1)^T - t).
struct Frustum {
  const float* kinv;     // (B, 9)
  const float* rinv;     // (B, 9)
  const float* t;        // (B, 3)
  const float* depths;   // (B, D)
  float* world;          // (B, 3, N) out, or nullptr
  int D;
};

template <int V, bool FRUSTUM>
__global__ __launch_bounds__(256) void fetch_variance_kernel(const float* __restrict__ maps,
                                                             const float* __restrict__ pts, Frustum fr,
                                                             const float* __restrict__ Kmat,
                                                             const float* __restrict__ Emat,
                                                             float* __restrict__ out, int C, int H, int W,
                                                             int64_t N, int ref_override) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (n >= N) return;
  float X, Y, Z;
  if (FRUSTUM) {
    const int hw = H * W;
    const int d = (int)(n / hw);
    const int pix = (int)(n - (int64_t)d * hw);
    const int py = pix / W, px = pix - py * W;
    const float gx = (float)px + 0.5f, gy = (float)py + 0.5f;
    const float* ki = fr.kinv + b * 9;
    const float* ri = fr.rinv + b * 9;
    const float* tt = fr.t + b * 3;
    const float depth = fr.depths[b * fr.D + d];
    // the dot products as fmaf chains in index order (the reference's are library matmuls of K = 3)
    const float u0 = fmaf(ki[2], 1.0f, fmaf(ki[1], gy, ki[0] * gx));
    const float u1 = fmaf(ki[5], 1.0f, fmaf(ki[4], gy, ki[3] * gx));
    const float u2 = fmaf(ki[8], 1.0f, fmaf(ki[7], gy, ki[6] * gx));
    const float c0 = u0 * depth - tt[0], c1 = u1 * depth - tt[1], c2 = u2 * depth - tt[2];
    X = fmaf(ri[2], c2, fmaf(ri[1], c1, ri[0] * c0));
    Y = fmaf(ri[5], c2, fmaf(ri[4], c1, ri[3] * c0));
    Z = fmaf(ri[8], c2, fmaf(ri[7], c1, ri[6] * c0));
    if (fr.world != nullptr) {
      float* wb = fr.world + b * 3 * N;
      wb[n] = X;
      wb[N + n] = Y;
      wb[2 * N + n] = Z;
    }
  } else {
    const float* pb = pts + b * 3 * N;
    X = pb[n];
    Y = pb[N + n];
    Z = pb[2 * N + n];
  }
  PfTaps t[V];
#pragma unroll
  for (int v = 0; v < V; ++v)
    pf_project_taps(X, Y, Z, Kmat + (b * V + v) * 9, Emat ? Emat + (b * V + v) * 12 : nullptr, H, W, t[v]);
  const int64_t HW = (int64_t)H * W;
  const int ref_off = (int)(n % HW);
  const float* mb = maps + b * V * C * HW;
  float* o = out + b * C * N + n;
#pragma unroll 2
  for (int c = 0; c < C; ++c) {
    float s = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const float* plane = mb + ((int64_t)v * C + c) * HW;
      const float f = (v == 0 && ref_override)
                          ? plane[ref_off]
                          : pf_bilerp(plane[t[v].off[0]], plane[t[v].off[1]], plane[t[v].off[2]], plane[t[v].off[3]],
                                      t[v].wgt[0], t[v].wgt[1], t[v].wgt[2], t[v].wgt[3]);
      if (v == 0) {
        s = f;
        s2 = f * f;
      } else {
        s = s + f;
        s2 = fmaf(f, f, s2);
      }
    }
    o[(int64_t)c * N] = pf_variance(s, s2, 1.0f / (float)V);
  }
}

// ------------------------------------------------------------------------------------------------
// W+V for the coarse cost volume, channel-last maps (reference model.py:79-111)
// ------------------------------------------------------------------------------------------------
// fetch_variance_kernel above walks the channels of NCHW maps with one lane per point: 4 scattered 4-byte
// loads per (channel, view), 64 channels in sequence -- latency-bound at 16 % of the HBM roof on the write of
// the cost volume (53 us for 70 MB at BASELINE config 2).  Here the maps are channel-last (B,V,H,W,C) and 16
// lanes share a point, lane q owning channels [4q, 4q+4) of a 64-channel pass: a bilinear tap is one 256-byte
// contiguous read per point, all V*4 taps of a point are in flight together, and there is no loop over
// channels.  A block owns 64 consecutive points (consecutive x of one cost-volume row, so their taps are
// neighbouring texels that stay in L1 between the four 16-point passes); the 64 x C variances are transposed
// through LDS so that every channel row leaves as a 256-byte segment of the (C, D*H*W) volume.
// Arithmetic per channel is that of fetch_variance_kernel (pf_sample's expression, sums in view order):
// the two kernels are bit-identical (tests/test_gpu_ops.py).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int64_t S) {
  __shared__ float tile[64][65];
  const int64_t p = blockIdx.z;
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r;
    const int64_t s = s0 + tx;
    tile[r][tx] = (c < C && s < S) ? in[(p * C + c) * S + s] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t s = s0 + r;
    const int c = c0 + tx;
    if (s < S && c < C) out[(p * S + s) * C + c] = tile[tx][r];
  }
}

template <int V>
__global__ __launch_bounds__(256) void frustum_variance_cl_kernel(const float* __restrict__ maps, Frustum fr,
                                                                  const float* __restrict__ Kmat,
                                                                  const float* __restrict__ Emat,
                                                                  float* __restrict__ out, int C, int H, int W,
                                                                  int64_t N) {
  constexpr int PTS = 64;
  __shared__ float tile[64 + 3][PTS + 1];
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int64_t b = blockIdx.y;
  const int hw = H * W;
  // (XCD x owns a band of pixel rows at ALL depths: the depth planes of a pixel sample neighbouring texels -- pf_common.h)
  unsigned ptile = blockIdx.x;
  if ((PF_XCD & PF_XCD_FRUSTUM) != 0 && hw % PTS == 0 && (int64_t)gridDim.x * PTS == N && (gridDim.x & 7u) == 0u)
    ptile = pf_xcd_band(ptile, (unsigned)(hw / PTS));
  const int64_t n0 = (int64_t)ptile * PTS;
  const float* ki = fr.kinv + b * 9;
  const float* ri = fr.rinv + b * 9;
  const float* tt = fr.t + b * 3;
  const float* mb = maps + b * V * (int64_t)hw * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
#pragma unroll 1
    for (int pass = 0; pass < PTS / 16; ++pass) {
      const int p = pass * 16 + pl;
      int64_t n = n0 + p;
      n = n < N ? n : N - 1;                               // lanes past the end shadow the last point
      const int d = (int)(n / hw);
      const int pix = (int)(n - (int64_t)d * hw);
      const int py = pix / W, px = pix - py * W;
      const float gx = (float)px + 0.5f, gy = (float)py + 0.5f;
      const float depth = fr.depths[b * fr.D + d];
      const float u0 = fmaf(ki[2], 1.0f, fmaf(ki[1], gy, ki[0] * gx));
      const float u1 = fmaf(ki[5], 1.0f, fmaf(ki[4], gy, ki[3] * gx));
      const float u2 = fmaf(ki[8], 1.0f, fmaf(ki[7], gy, ki[6] * gx));
      const float q0 = u0 * depth - tt[0], q1 = u1 * depth - tt[1], q2 = u2 * depth - tt[2];
      const float X = fmaf(ri[2], q2, fmaf(ri[1], q1, ri[0] * q0));
      const float Y = fmaf(ri[5], q2, fmaf(ri[4], q1, ri[3] * q0));
      const float Z = fmaf(ri[8], q2, fmaf(ri[7], q1, ri[6] * q0));
      if (c0 == 0 && q < 3) tile[64 + q][p] = q == 0 ? X : (q == 1 ? Y : Z);
      // the V - 1 projections of a point: lane q of its 16 lanes computes ONE of them (view 1 + q mod (V - 1)) and the
      // 16 lanes exchange taps and weights by shuffles -- round 2 had every lane compute all V - 1 (~150 instructions
      // each, two IEEE divisions: two thirds of the kernel's arithmetic at V = 3, 85 % at V = 7)
      PfTaps t[V];
      if (V > 1) {
        const int vq = 1 + (q % (V > 1 ? V - 1 : 1));
        PfTaps mine;
        pf_project_taps(X, Y, Z, Kmat + (b * V + vq) * 9, Emat ? Emat + (b * V + vq) * 12 : nullptr, H, W, mine);
#pragma unroll
        for (int v = 1; v < V; ++v) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            t[v].off[k] = __shfl(mine.off[k], v - 1, 16);
            t[v].wgt[k] = __shfl(mine.wgt[k], v - 1, 16);
          }
        }
      }
      const int c = c0 + 4 * q;
      if (c < C) {
        // view 0 contributes its un-warped feature (model.py:103-106)
        const float4 r0 = *reinterpret_cast<const float4*>(mb + (int64_t)pix * C + c);
        float s[4] = {r0.x, r0.y, r0.z, r0.w};
        float s2[4] = {r0.x * r0.x, r0.y * r0.y, r0.z * r0.z, r0.w * r0.w};
#pragma unroll
        for (int v = 1; v < V; ++v) {
          const float* mv = mb + (int64_t)v * hw * C + c;
          const float4 a = *reinterpret_cast<const float4*>(mv + (int64_t)t[v].off[0] * C);
          const float4 bb = *reinterpret_cast<const float4*>(mv + (int64_t)t[v].off[1] * C);
          const float4 cc = *reinterpret_cast<const float4*>(mv + (int64_t)t[v].off[2] * C);
          const float4 dd = *reinterpret_cast<const float4*>(mv + (int64_t)t[v].off[3] * C);
          const float w0 = t[v].wgt[0], w1 = t[v].wgt[1], w2 = t[v].wgt[2], w3 = t[v].wgt[3];
          const float f[4] = {pf_bilerp(a.x, bb.x, cc.x, dd.x, w0, w1, w2, w3), pf_bilerp(a.y, bb.y, cc.y, dd.y, w0, w1, w2, w3),
                              pf_bilerp(a.z, bb.z, cc.z, dd.z, w0, w1, w2, w3), pf_bilerp(a.w, bb.w, cc.w, dd.w, w0, w1, w2, w3)};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s[i] = s[i] + f[i];
            s2[i] = fmaf(f[i], f[i], s2[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[4 * q + i][p] = pf_variance(s[i], s2[i], 1.0f / (float)V);
      }
    }
    __syncthreads();
    // rows of the tile -> 256-byte segments of the (C, N) volume; plus the three rows of world points once
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bool inb = n0 + tx < N;
    const int rows = min(64, C - c0);
    for (int r = ty; r < rows; r += 4)
      if (inb) out[(b * C + c0 + r) * N + n0 + tx] = tile[r][tx];
    if (c0 == 0 && fr.world != nullptr && ty < 3 && inb) fr.world[(b * 3 + ty) * N + n0 + tx] = tile[64 + ty][tx];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (F.interpolate at reference model.py:184)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_axis(int o, float scale, int in_size, int& i0, int& i1, float& l0,
                                            float& l1) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in,
                                                              float* __restrict__ out, int IH, int IW, int OH,
                                                              int OW, float sy, float sx) {
  const int64_t p = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= OH * OW) return;
  const int oy = i / OW, ox = i - oy * OW;
  const float* src = in + p * IH * IW;
  float v;
  if (IH == OH && IW == OW) {
    v = src[i];
  } else {
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    resize_axis(oy, sy, IH, y0, y1, ly0, ly1);
    resize_axis(ox, sx, IW, x0, x1, lx0, lx1);
    const float a = src[y0 * IW + x0], b = src[y0 * IW + x1];
    const float c = src[y1 * IW + x0], d = src[y1 * IW + x1];
    v = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c + lx1 * d);
  }
  out[p * OH * OW + i] = v;
}

// The three pyramid levels of one flow iteration in ONE launch, written CHANNEL-LAST (V, OH, OW, C): the
// consumer (flow_features_kernel) reads every bilinear tap of a point as 16-byte loads of 4 consecutive
// channels instead of one scattered 4-byte load per (channel, view, tap) -- 4x fewer texture-address
// operations in the kernel that was bound by them.  A level already at (OH, OW) is copied (transposed).
struct PyramidLevel {
  const float* in;
  float* out;
  const float* scale;   // (V, C) rows of a pending BatchNorm + ReLU of `in`, or nullptr: in is taken as is
  const float* shift;
  int C, IH, IW;
};
struct Pyramid {
  PyramidLevel l[3];
};

__global__ __launch_bounds__(256) void pyramid_resize_kernel(Pyramid py, int V, int OH, int OW, int cq_max) {
  const PyramidLevel L = py.l[blockIdx.z];
  const int v = blockIdx.y / cq_max, cq = blockIdx.y - v * cq_max;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * cq >= L.C || i >= OH * OW) return;
  const int oy = i / OW, ox = i - oy * OW;
  const int plane = L.IH * L.IW;
  const float* src = L.in + ((int64_t)v * L.C + 4 * cq) * plane;
  float r[4];
  // the tower's last BatchNorm + ReLU of this level, applied to every texel BEFORE it is interpolated (round 3: the
  // tower used to write the normalised maps with a pass of its own; this kernel reads the raw convolution output)
  float sc[4] = {1.0f, 1.0f, 1.0f, 1.0f}, sh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool aff = L.scale != nullptr;
  if (aff) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sc[c] = L.scale[(int64_t)v * L.C + 4 * cq + c];
      sh[c] = L.shift[(int64_t)v * L.C + 4 * cq + c];
    }
  }
  auto act = [&](float x, int c) { return aff ? fmaxf(fmaf(x, sc[c], sh[c]), 0.0f) : x; };
  if (L.IH == OH && L.IW == OW) {
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = act(src[(int64_t)c * plane + i], c);
  } else {
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    resize_axis(oy, (float)L.IH / (float)OH, L.IH, y0, y1, ly0, ly1);
    resize_axis(ox, (float)L.IW / (float)OW, L.IW, x0, x1, lx0, lx1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* pc = src + (int64_t)c * plane;
      const float a = act(pc[y0 * L.IW + x0], c), b = act(pc[y0 * L.IW + x1], c);
      const float cc = act(pc[y1 * L.IW + x0], c), d = act(pc[y1 * L.IW + x1], c);
      r[c] = ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * cc + lx1 * d);
    }
  }
  *reinterpret_cast<float4*>(L.out + ((int64_t)v * OH * OW + i) * L.C + 4 * cq) = make_float4(r[0], r[1], r[2], r[3]);
}

// ------------------------------------------------------------------------------------------------
// F: flow feature assembly (reference model.py:153-204), sub-grid-major output (model.py:236-255)
// ------------------------------------------------------------------------------------------------
// Eight lanes share one point: lane q samples channels [32 p + 4 q, +4) of a level (p = pass), so one tap of
// one point is read as 8 x 16 B = one full 128-byte line of the channel-last map.  (One lane per point with
// a loop over channel quads -- the previous layout -- touches ~400 lines per wave and iteration and
// comes back to each of them 8 times, long after a 32 KB L1 has dropped it.)  The per-point projection
// setup is replicated on the 8 lanes (a few hundred VALU instructions against ~50 line fetches).
// Output rows are POINT-major: feature (G*Ng, ctot), a point's ctot floats contiguous (the first
// EdgeConv GEMM reads them as its A operand); xyz stays planar (G, 3, Ng) for the lattice kNN.
constexpr int kFeatLanes = 8;
constexpr int kFeatPY = 4, kFeatPX = 8;           // 32 points per 256-thread block
static_assert(kFeatPY * kFeatPX * kFeatLanes == 256, "one point per 8 lanes");

// The five hypotheses of a pixel inside ONE block.  Round 1's kernel gave every hypothesis plane its own block; SQ / PMC
// counters put it (and the coarse warp) at ~10 TB/s of L2 -> L1 tap traffic (550 MB of 128-byte tap lines per launch
// at 4 x 25 600 points), not at HBM; it was removed in round 3 (59 vs 43 us at flow-2, outputs bit-identical).
// The five hypotheses of a pixel are 0.1-0.3 texels apart in every source view, i.e. they sample the SAME four
// texels almost always.  Here a block owns a 4 x 8 pixel patch with all five hypotheses and walks
// level -> 32-channel chunk -> view -> hypothesis: within one (level, view) the five hypotheses re-read lines
// that are in L1 already (6 x 10 texels x 64-256 B per view and level), which cuts the L2 traffic ~5x.  The
// 15 (view, hypothesis) projections of a point are computed once by its 8 lanes (two each) and parked in LDS.
// Per (hypothesis, channel) the views are accumulated in ascending order.
template <int V, bool ROLLV>
__global__ __launch_bounds__(256) void flow_features_hyp_kernel(const float* __restrict__ maps1,
                                                                const float* __restrict__ maps2,
                                                                const float* __restrict__ maps3, int c1, int c2,
                                                                int c3, int h, int w,
                                                                const float* __restrict__ depth_in, int dh, int dw,
                                                                const float* __restrict__ interval_p,
                                                                const float* __restrict__ cam, int ratio,
                                                                float* __restrict__ feature, float* __restrict__ xyz) {
  constexpr int PTS = kFeatPY * kFeatPX;                    // 32 points (pixels) per block
  constexpr int kUnrollV = ROLLV ? 1 : V;                   // view loop: rolled (fewer registers) or unrolled
  __shared__ int tap_off[PTS][V * 5][4];
  __shared__ float tap_wgt[PTS][V * 5][4];
  __shared__ float nxyz[PTS][5][4];
  const int hs = h / ratio, ws = w / ratio;
  const int64_t Ng = (int64_t)5 * hs * ws;
  const int q = threadIdx.x & (kFeatLanes - 1);
  const int pl = threadIdx.x / kFeatLanes;
  const int tiles_x = (w + kFeatPX - 1) / kFeatPX;
  // (XCD x owns a band of patch rows: the bilinear footprints of neighbouring patches overlap -- pf_common.h)
  const unsigned blk = (PF_XCD & PF_XCD_FETCH) ? pf_xcd_chunk(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bx = (int)blk % tiles_x, by = (int)blk / tiles_x;
  const int y_raw = by * kFeatPY + pl / kFeatPX, x_raw = bx * kFeatPX + pl % kFeatPX;
  const bool live = y_raw < h && x_raw < w;
  const int y = live ? y_raw : h - 1, x = live ? x_raw : w - 1;
  const int g = (y % ratio) * ratio + (x % ratio);
  const int64_t loc0 = (int64_t)(y / ratio) * ws + x / ratio;     // + d*hs*ws per hypothesis

  const float scy = (float)dh / (float)h, scx = (float)dw / (float)w;
  int sy = (int)floorf((float)y * scy);
  int sx = (int)floorf((float)x * scx);
  sy = sy > dh - 1 ? dh - 1 : sy;
  sx = sx > dw - 1 ? dw - 1 : sx;
  const float depth0 = depth_in[sy * dw + sx];
  const float interval = interval_p[0];
  const float* Ki = cam + PF_CAM_KREF_INV;
  const float* Ri = cam + PF_CAM_RREF_INV;
  const float* t0 = cam + PF_CAM_TREF;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;
  const float u0 = fmaf(Ki[2], 1.0f, fmaf(Ki[1], py, Ki[0] * px));
  const float u1 = fmaf(Ki[5], 1.0f, fmaf(Ki[4], py, Ki[3] * px));
  const float u2 = fmaf(Ki[8], 1.0f, fmaf(Ki[7], py, Ki[6] * px));
  auto world = [&](int d, float& X, float& Y, float& Z) {
    const float depth = depth0 + interval * (float)(d - 2);
    const float q0 = u0 * depth - t0[0], q1 = u1 * depth - t0[1], q2 = u2 * depth - t0[2];
    X = fmaf(Ri[2], q2, fmaf(Ri[1], q1, Ri[0] * q0));
    Y = fmaf(Ri[5], q2, fmaf(Ri[4], q1, Ri[3] * q0));
    Z = fmaf(Ri[8], q2, fmaf(Ri[7], q1, Ri[6] * q0));
  };
  // phase 0: the V*5 (view, hypothesis) projections of this point, spread over its 8 lanes
  for (int p = q; p < V * 5; p += kFeatLanes) {
    const int v = p / 5, d = p - v * 5;
    float X, Y, Z;
    world(d, X, Y, Z);
    PfTaps t;
    const float* cv = cam + PF_CAM_VIEWS + v * PF_CAM_VIEW_STRIDE;
    pf_project_taps(X, Y, Z, cv, cv + 9, h, w, t);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tap_off[pl][p][k] = t.off[k];
      tap_wgt[pl][p][k] = t.wgt[k];
    }
  }
  if (q < 5) {
    float X, Y, Z;
    world(q, X, Y, Z);
    const float nx = (X - cam[PF_CAM_MEAN + 0]) / cam[PF_CAM_STD + 0];
    const float ny = (Y - cam[PF_CAM_MEAN + 1]) / cam[PF_CAM_STD + 1];
    const float nz = (Z - cam[PF_CAM_MEAN + 2]) / cam[PF_CAM_STD + 2];
    nxyz[pl][q][0] = nx;
    nxyz[pl][q][1] = ny;
    nxyz[pl][q][2] = nz;
    if (live) {
      float* xo = xyz + (int64_t)g * 3 * Ng + (int64_t)q * hs * ws + loc0;
      xo[0] = nx;
      xo[Ng] = ny;
      xo[2 * Ng] = nz;
    }
  }
  __syncthreads();

  const int ctot = c1 + c2 + c3 + 24;
  float* frow0 = feature + ((int64_t)g * Ng + loc0) * ctot;       // + d*hs*ws*ctot per hypothesis
  const int64_t dstride = (int64_t)hs * ws * ctot;
  const int64_t hw = (int64_t)h * w;
  int ch = 0;
#pragma unroll 1
  for (int level = 0; level < 3; ++level) {
    const float* maps = level == 0 ? maps1 : (level == 1 ? maps2 : maps3);
    const int cl = level == 0 ? c1 : (level == 1 ? c2 : c3);
#pragma unroll 1
    for (int c0 = 0; c0 < cl; c0 += 4 * kFeatLanes) {
      const int c = c0 + 4 * q;
      if (c < cl) {
        float s[5][4], s2[5][4];
        // (a zero start instead of "first view assigns" keeps every bit: 0 + f == f, and the sign of a zero sum
        // cannot reach m2 - m1 * m1)
#pragma unroll
        for (int d = 0; d < 5; ++d)
#pragma unroll
          for (int i = 0; i < 4; ++i) s[d][i] = s2[d][i] = 0.0f;
#pragma unroll kUnrollV
        for (int v = 0; v < V; ++v) {
          const float* mv = maps + (int64_t)v * hw * cl + c;
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            const int4 o = *reinterpret_cast<const int4*>(&tap_off[pl][v * 5 + d][0]);
            const float4 wg = *reinterpret_cast<const float4*>(&tap_wgt[pl][v * 5 + d][0]);
            const float4 a = *reinterpret_cast<const float4*>(mv + (int64_t)o.x * cl);
            const float4 b = *reinterpret_cast<const float4*>(mv + (int64_t)o.y * cl);
            const float4 cc = *reinterpret_cast<const float4*>(mv + (int64_t)o.z * cl);
            const float4 dd = *reinterpret_cast<const float4*>(mv + (int64_t)o.w * cl);
            const float f[4] = {pf_bilerp(a.x, b.x, cc.x, dd.x, wg.x, wg.y, wg.z, wg.w),
                                pf_bilerp(a.y, b.y, cc.y, dd.y, wg.x, wg.y, wg.z, wg.w),
                                pf_bilerp(a.z, b.z, cc.z, dd.z, wg.x, wg.y, wg.z, wg.w),
                                pf_bilerp(a.w, b.w, cc.w, dd.w, wg.x, wg.y, wg.z, wg.w)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              s[d][i] = s[d][i] + f[i];
              s2[d][i] = fmaf(f[i], f[i], s2[d][i]);
            }
          }
        }
        if (live) {
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pf_variance(s[d][i], s2[d][i], 1.0f / (float)V);
            *reinterpret_cast<float4*>(frow0 + d * dstride + ch + c) = make_float4(o[0], o[1], o[2], o[3]);
          }
        }
      }
    }
    ch += cl;
  }
  // xyz.repeat(1, 8, 1): channel j of the 24 holds axis j % 3 (model.py:193-194); lanes 0..5 write 4 each
  if (live && q < 6) {
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = nxyz[pl][d][(4 * q + i) % 3];
      *reinterpret_cast<float4*>(frow0 + d * dstride + ch + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// S: soft-argmin + probability map (reference model.py:117-130, functions/functions.py:141-175)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_at(float start, float end, float step, int k, int D) {
  // ATen linspace: start + step*k below the midpoint, end - step*(D-1-k) above it
  return (k < D / 2) ? (start + step * (float)k) : (end - step * (float)(D - 1 - k));
}

// 64 pixels x 4 depth slices per block: the three passes over D (max, sum of exp, expectation) are split
// over the 4 waves and combined through LDS in a fixed order (5120 pixels alone would be 20 blocks).
__global__ __launch_bounds__(256) void softargmin_prob_kernel(const float* __restrict__ cost,
                                                              const float* __restrict__ params,
                                                              float* __restrict__ depth,
                                                              float* __restrict__ prob, int D, int64_t HW) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  const bool ok = i < HW;
  const float start = params[b * 3 + 0], end = params[b * 3 + 1], interval = params[b * 3 + 2];
  const float step = (D > 1) ? (end - start) / (float)(D - 1) : 0.0f;
  const float* c = cost + b * D * HW + (ok ? i : 0);
  const int dq = (D + 3) >> 2;
  const int k0 = q * dq, k1 = min(D, k0 + dq);

  float mx = -__builtin_huge_valf();
  for (int k = k0; k < k1; ++k) mx = fmaxf(mx, -c[(int64_t)k * HW]);
  sh[q][lane] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0][lane], sh[1][lane]), fmaxf(sh[2][lane], sh[3][lane]));
  __syncthreads();

  float den = 0.0f;
  for (int k = k0; k < k1; ++k) den += expf(-c[(int64_t)k * HW] - mx);
  sh[q][lane] = den;
  __syncthreads();
  den = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
  __syncthreads();

  float acc = 0.0f;
  for (int k = k0; k < k1; ++k) acc += linspace_at(start, end, step, k, D) * (expf(-c[(int64_t)k * HW] - mx) / den);
  sh[q][lane] = acc;
  __syncthreads();
  if (q != 0 || !ok) return;
  acc = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
  depth[b * HW + i] = acc;
  const float fi = (acc - start) / interval;
  float lo = floorf(fi), hi = ceilf(fi);
  lo = fminf(fmaxf(lo, 0.0f), (float)(D - 1));
  hi = fminf(fmaxf(hi, 0.0f), (float)(D - 1));
  const float plo = expf(-c[(int64_t)(int)lo * HW] - mx) / den;
  const float phi = expf(-c[(int64_t)(int)hi * HW] - mx) / den;
  prob[b * HW + i] = plo + phi;
}

template <typename F>
int dispatch_views(int V, F&& f) {
  switch (V) {
    case 1: return f(std::integral_constant<int, 1>());
    case 2: return f(std::integral_constant<int, 2>());
    case 3: return f(std::integral_constant<int, 3>());
    case 4: return f(std::integral_constant<int, 4>());
    case 5: return f(std::integral_constant<int, 5>());
    case 6: return f(std::integral_constant<int, 6>());
    case 7: return f(std::integral_constant<int, 7>());
    case 8: return f(std::integral_constant<int, 8>());
    default: return PF_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int pf_fetch_forward_f32(const float* maps, const float* pts, const float* K, const float* E, float* out,
                         int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t N, void* stream) {
  PF_REQUIRE(B >= 0 && V >= 0 && C >= 0 && H >= 1 && W >= 1 && N >= 0);
  PF_REQUIRE(B <= 65535 && V <= 65535 && H * W <= INT32_MAX && C <= INT32_MAX);
  if (B == 0 || V == 0 || C == 0 || N == 0) return PF_OK;
  PF_REQUIRE(maps && pts && K && out);
  dim3 grid((unsigned)pf_cdiv(N, 256), (unsigned)V, (unsigned)B);
  hipLaunchKernelGGL(fetch_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, maps, pts, K, E, out, (int)V,
                     (int)C, (int)H, (int)W, N);
  return pf_launch_status();
}

int pf_fetch_backward_f32(const float* grad_out, const float* pts, const float* K, const float* E,
                          float* grad_maps, int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t N,
                          void* stream) {
  PF_REQUIRE(B >= 0 && V >= 0 && C >= 0 && H >= 1 && W >= 1 && N >= 0);
  PF_REQUIRE(B <= 65535 && V <= 65535 && H * W <= INT32_MAX && C <= INT32_MAX);
  if (B == 0 || V == 0 || C == 0) return PF_OK;
  PF_REQUIRE(grad_maps != nullptr);
  {
    const int zrc = pf_zero_async(grad_maps, sizeof(float) * (size_t)(B * V * C * H * W), (hipStream_t)stream);
    if (zrc != PF_OK) return zrc;
  }
  if (N == 0) return PF_OK;
  PF_REQUIRE(grad_out && pts && K);
  dim3 grid((unsigned)pf_cdiv(N, 256), (unsigned)V, (unsigned)B);
  hipLaunchKernelGGL(fetch_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, pts, K, E, grad_maps,
                     (int)V, (int)C, (int)H, (int)W, N);
  return pf_launch_status();
}

int pf_fetch_variance_f32(const float* maps, const float* pts, const float* K, const float* E, float* out,
                          int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t N, int ref_override,
                          void* stream) {
  PF_REQUIRE(B >= 0 && V >= 1 && C >= 0 && H >= 1 && W >= 1 && N >= 0);
  PF_REQUIRE(B <= 65535 && H * W <= INT32_MAX && C <= INT32_MAX);
  if (V > PF_MAX_VIEWS) return PF_ERR_UNSUPPORTED;
  if (B == 0 || C == 0 || N == 0) return PF_OK;
  PF_REQUIRE(maps && pts && K && out);
  dim3 grid((unsigned)pf_cdiv(N, 256), (unsigned)B);
  return dispatch_views((int)V, [&](auto vtag) {
    constexpr int VV = decltype(vtag)::value;
    hipLaunchKernelGGL((fetch_variance_kernel<VV, false>), grid, dim3(256), 0, (hipStream_t)stream, maps, pts,
                       Frustum{}, K, E, out, (int)C, (int)H, (int)W, N, ref_override);
    return pf_launch_status();
  });
}

int pf_frustum_variance_f32(const float* maps, const float* kinv, const float* rinv, const float* t,
                            const float* depths, const float* K, const float* E, float* out, float* world,
                            int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t D, void* stream) {
  PF_REQUIRE(B >= 0 && V >= 1 && C >= 0 && H >= 1 && W >= 1 && D >= 0);
  PF_REQUIRE(B <= 65535 && H * W <= INT32_MAX && C <= INT32_MAX && D <= INT32_MAX);
  if (V > PF_MAX_VIEWS) return PF_ERR_UNSUPPORTED;
  const int64_t N = D * H * W;
  if (B == 0 || N == 0) return PF_OK;
  PF_REQUIRE(maps && kinv && rinv && t && depths && K && out);
  Frustum fr;
  fr.kinv = kinv;
  fr.rinv = rinv;
  fr.t = t;
  fr.depths = depths;
  fr.world = world;
  fr.D = (int)D;
  dim3 grid((unsigned)pf_cdiv(N, 256), (unsigned)B);
  return dispatch_views((int)V, [&](auto vtag) {
    constexpr int VV = decltype(vtag)::value;
    hipLaunchKernelGGL((fetch_variance_kernel<VV, true>), grid, dim3(256), 0, (hipStream_t)stream, maps, nullptr, fr,
                       K, E, out, (int)C, (int)H, (int)W, N, 1);
    return pf_launch_status();
  });
}

int pf_nchw_to_nhwc_f32(const float* in, float* out, int64_t P, int64_t C, int64_t S, void* stream) {
  PF_REQUIRE(P >= 0 && C >= 1 && S >= 1 && P <= 65535 && C <= INT32_MAX && pf_cdiv(C, 64) <= 65535);
  if (P == 0) return PF_OK;
  PF_REQUIRE(in && out);
  dim3 grid((unsigned)pf_cdiv(S, 64), (unsigned)pf_cdiv(C, 64), (unsigned)P);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)C, S);
  return pf_launch_status();
}

int pf_frustum_variance_cl_f32(const float* maps_cl, const float* kinv, const float* rinv, const float* t,
                               const float* depths, const float* K, const float* E, float* out, float* world,
                               int64_t B, int64_t V, int64_t C, int64_t H, int64_t W, int64_t D, void* stream) {
  PF_REQUIRE(B >= 0 && V >= 1 && C >= 0 && H >= 1 && W >= 1 && D >= 0);
  PF_REQUIRE(B <= 65535 && H * W <= INT32_MAX && C <= INT32_MAX && D <= INT32_MAX);
  if (V > PF_MAX_VIEWS || (C % 4) != 0) return PF_ERR_UNSUPPORTED;
  const int64_t N = D * H * W;
  if (B == 0 || N == 0 || C == 0) return PF_OK;
  PF_REQUIRE(maps_cl && kinv && rinv && t && depths && K && out);
  PF_REQUIRE(H * W * C <= (int64_t)INT32_MAX * 4);
  Frustum fr;
  fr.kinv = kinv;
  fr.rinv = rinv;
  fr.t = t;
  fr.depths = depths;
  fr.world = world;
  fr.D = (int)D;
  dim3 grid((unsigned)pf_cdiv(N, 64), (unsigned)B);
  return dispatch_views((int)V, [&](auto vtag) {
    constexpr int VV = decltype(vtag)::value;
    hipLaunchKernelGGL((frustum_variance_cl_kernel<VV>), grid, dim3(256), 0, (hipStream_t)stream, maps_cl, fr, K, E,
                       out, (int)C, (int)H, (int)W, N);
    return pf_launch_status();
  });
}

int pf_resize_bilinear_f32(const float* in, float* out, int64_t P, int64_t IH, int64_t IW, int64_t OH,
                           int64_t OW, void* stream) {
  PF_REQUIRE(P >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1);
  PF_REQUIRE(P <= 65535 && IH * IW <= INT32_MAX && OH * OW <= INT32_MAX);
  if (P == 0) return PF_OK;
  PF_REQUIRE(in && out);
  dim3 grid((unsigned)pf_cdiv(OH * OW, 256), (unsigned)P);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  hipLaunchKernelGGL(resize_bilinear_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, (int)IH, (int)IW,
                     (int)OH, (int)OW, sy, sx);
  return pf_launch_status();
}

int pf_flow_pyramid_f32(const float* in1, int c1, int h1, int w1, const float* in2, int c2, int h2, int w2,
                        const float* in3, int c3, int h3, int w3, int V, int h, int w, float* out1, float* out2,
                        float* out3, const float* const* in_scale, const float* const* in_shift, void* stream) {
  PF_REQUIRE(V >= 1 && h >= 1 && w >= 1 && c1 >= 0 && c2 >= 0 && c3 >= 0);
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  const float* sc[3] = {nullptr, nullptr, nullptr};
  const float* sh[3] = {nullptr, nullptr, nullptr};
  for (int l = 0; in_scale != nullptr && l < 3; ++l) {
    PF_REQUIRE((in_scale[l] == nullptr) == (in_shift[l] == nullptr));
    sc[l] = in_scale[l];
    sh[l] = in_shift[l];
  }
  PF_REQUIRE(h1 >= 1 && w1 >= 1 && h2 >= 1 && w2 >= 1 && h3 >= 1 && w3 >= 1);
  if ((c1 % 4) != 0 || (c2 % 4) != 0 || (c3 % 4) != 0) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE((int64_t)h * w <= INT32_MAX && (int64_t)h1 * w1 <= INT32_MAX && (int64_t)h2 * w2 <= INT32_MAX &&
             (int64_t)h3 * w3 <= INT32_MAX);
  int cmax = c1 > c2 ? c1 : c2;
  cmax = cmax > c3 ? cmax : c3;
  if (cmax == 0) return PF_OK;
  PF_REQUIRE((c1 == 0 || (in1 && out1)) && (c2 == 0 || (in2 && out2)) && (c3 == 0 || (in3 && out3)));
  PF_REQUIRE((int64_t)V * (cmax / 4) <= 65535);
  Pyramid py;
  py.l[0] = PyramidLevel{in1, out1, sc[0], sh[0], c1, h1, w1};
  py.l[1] = PyramidLevel{in2, out2, sc[1], sh[1], c2, h2, w2};
  py.l[2] = PyramidLevel{in3, out3, sc[2], sh[2], c3, h3, w3};
  dim3 grid((unsigned)pf_cdiv((int64_t)h * w, 256), (unsigned)(V * (cmax / 4)), 3);
  hipLaunchKernelGGL(pyramid_resize_kernel, grid, dim3(256), 0, (hipStream_t)stream, py, V, h, w, cmax / 4);
  return pf_launch_status();
}

int pf_flow_features_f32(const float* maps1, const float* maps2, const float* maps3, int c1, int c2, int c3,
                         int V, int h, int w, const float* depth_in, int dh, int dw, const float* interval,
                         const float* cam, int ratio, float* feature, float* xyz, void* stream) {
  PF_REQUIRE(c1 >= 0 && c2 >= 0 && c3 >= 0 && V >= 1 && h >= 1 && w >= 1 && dh >= 1 && dw >= 1 && ratio >= 1);
  if ((c1 % 4) != 0 || (c2 % 4) != 0 || (c3 % 4) != 0) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(h % ratio == 0 && w % ratio == 0);
  PF_REQUIRE(ratio * ratio <= 65535);
  if (V > PF_MAX_VIEWS) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(maps1 && maps2 && maps3 && depth_in && interval && cam && feature && xyz);
  const int64_t Ng = (int64_t)5 * (h / ratio) * (w / ratio);
  (void)Ng;
  PF_REQUIRE(pf_cdiv(h, kFeatPY) * pf_cdiv(w, kFeatPX) * 5 <= INT32_MAX);
  dim3 gridh((unsigned)(pf_cdiv(h, kFeatPY) * pf_cdiv(w, kFeatPX)));
  return dispatch_views(V, [&](auto vtag) {
    constexpr int VV = decltype(vtag)::value;
    hipLaunchKernelGGL((flow_features_hyp_kernel<VV, false>), gridh, dim3(256), 0, (hipStream_t)stream, maps1, maps2,
                       maps3, c1, c2, c3, h, w, depth_in, dh, dw, interval, cam, ratio, feature, xyz);
    return pf_launch_status();
  });
}

int pf_softargmin_prob_f32(const float* cost, const float* params, float* depth, float* prob, int64_t B,
                           int64_t D, int64_t HW, void* stream) {
  PF_REQUIRE(B >= 0 && D >= 1 && HW >= 0 && B <= 65535 && D <= INT32_MAX);
  if (B == 0 || HW == 0) return PF_OK;
  PF_REQUIRE(cost && params && depth && prob);
  dim3 grid((unsigned)pf_cdiv(HW, 64), (unsigned)B);
  hipLaunchKernelGGL(softargmin_prob_kernel, grid, dim3(256), 0, (hipStream_t)stream, cost, params, depth, prob,
                     (int)D, HW);
  return pf_launch_status();
}

}  // extern "C"
