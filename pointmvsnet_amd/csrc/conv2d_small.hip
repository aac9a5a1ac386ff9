// ImageConv, the few-channel layers: 3 -> 8 and 8 -> 8 at full resolution,
// 8 -> 16 with a 5x5 / stride-2 kernel.  These are HBM-bound (72..200 multiply-adds per input float, 32-64 B
// moved per pixel) and took 62-86 us each in the library's Winograd kernel plus two BatchNorm passes
// (profiles/r01c_last_step_dispatches.txt lines 41-49); an MFMA tile is half empty at 8 output channels and
// instruction-bound (conv2d.hip).  So: plain float32 FMAs.
//
//   * one lane = one output column, PPT output rows; all C_out accumulators in registers;
//   * the block's input patch is staged in LDS four channels at a time (the previous layer's
//     BatchNorm+ReLU applied on the way in, zero outside the image); the group's weights are wave-uniform
//     and come through the scalar cache ([ch][kh][kw][C_out], s_load into SGPRs), reused for the PPT rows;
//   * stores are 128-byte row segments per channel straight from registers; per-channel sum / sum of
//     squares are carried in registers across the block's tiles and reduced once (float64 partials).
#include "pf_common.h"

namespace {

struct SmallGeom {
  int Cin, Hi, Wi, Ho, Wo, tiles_h, tiles_w, sps;
};

template <int COUT, int KS, int STRIDE, int PPT>
__global__ __launch_bounds__(256) void conv2d_small_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           float* __restrict__ y, SmallGeom g,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           double* __restrict__ partials) {
  constexpr int TR = 8 * PPT, TC = 32;                 // output tile rows x cols
  constexpr int IH = (TR - 1) * STRIDE + KS, IW = (TC - 1) * STRIDE + KS;
  constexpr int IWP = IW | 1;                          // odd row stride
  constexpr int PAD = KS / 2;
  constexpr int WSZ = 4 * KS * KS * COUT;
  __shared__ __attribute__((aligned(16))) float xs[4 * IH * IWP];
  __shared__ double red[4][2 * COUT];
  const int tid = threadIdx.x;
  const int col = tid & 31, row = tid >> 5;            // 32 columns x 8 rows of lanes
  const int n = blockIdx.y;
  const int64_t plane_i = (int64_t)g.Hi * g.Wi, plane_o = (int64_t)g.Ho * g.Wo;
  const float* xb = x + (int64_t)n * g.Cin * plane_i;
  float* yb = y + (int64_t)n * COUT * plane_o;
  const int cgroups = (g.Cin + 3) >> 2;
  const float* sc = in_scale ? in_scale + (int64_t)(n / g.sps) * g.Cin : nullptr;
  const float* sh = in_scale ? in_shift + (int64_t)(n / g.sps) * g.Cin : nullptr;

  float ssum[COUT], ssq[COUT];     // a lane adds only its own few dozen pixels: float is ample here
#pragma unroll
  for (int c = 0; c < COUT; ++c) ssum[c] = ssq[c] = 0.0f;

  const int total = g.tiles_h * g.tiles_w;
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const int tw = item % g.tiles_w, th = item / g.tiles_w;
    const int oh0 = th * TR, ow0 = tw * TC;
    const int ih0 = oh0 * STRIDE - PAD, iw0 = ow0 * STRIDE - PAD;
    float acc[PPT][COUT];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[p][c] = 0.0f;

    // Staging issues U global loads back to back before their LDS stores: a one-load-one-store loop waits
    // out a full memory latency per element (10-37 of them per group; measured 42 us vs a 12 us HBM floor).
    constexpr int U = KS == 3 ? 5 : 8;
    for (int cg = 0; cg < cgroups; ++cg) {
      __syncthreads();                        // everyone is done reading the previous group from LDS
      for (int e0 = tid; e0 < 4 * IH * IW; e0 += 256 * U) {
        float v[U];
        int lo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + 256 * u;
          const int ch = e / (IH * IW);
          const int rem = e - ch * (IH * IW);
          const int yy = rem / IW, xx = rem - yy * IW;
          const int c = cg * 4 + ch, ih = ih0 + yy, iw = iw0 + xx;
          const bool ok = e < 4 * IH * IW && c < g.Cin && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
          float t = ok ? xb[(int64_t)c * plane_i + (int64_t)ih * g.Wi + iw] : 0.0f;
          if (sc != nullptr && ok) t = fmaxf(fmaf(t, sc[c], sh[c]), 0.0f);    // previous BatchNorm + ReLU
          v[u] = t;
          lo[u] = e < 4 * IH * IW ? (ch * IH + yy) * IWP + xx : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (lo[u] >= 0) xs[lo[u]] = v[u];
      }
      __syncthreads();
      // Weights are wave-uniform: read them through the scalar cache into SGPRs (s_load) instead of LDS
      // broadcasts -- with them in LDS the kernel was LDS-bandwidth bound (12 LDS cycles per 16 FMAs x 4 SIMDs).
      const float* __restrict__ wg = wp + (int64_t)cg * WSZ;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            float w[COUT];
#pragma unroll
            for (int c = 0; c < COUT; ++c) w[c] = wg[((ch * KS + kh) * KS + kw) * COUT + c];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
              const float v = xs[(ch * IH + (row + 8 * p) * STRIDE + kh) * IWP + col * STRIDE + kw];
#pragma unroll
              for (int c = 0; c < COUT; ++c) acc[p][c] = fmaf(v, w[c], acc[p][c]);
            }
          }
        }
      }
    }

    const int ow = ow0 + col;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int oh = oh0 + row + 8 * p;
      if (oh < g.Ho && ow < g.Wo) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const float v = acc[p][c];
          yb[(int64_t)c * plane_o + (int64_t)oh * g.Wo + ow] = v;
          ssum[c] += v;
          ssq[c] += v * v;
        }
      }
    }
  }

  if (partials != nullptr) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      double s = (double)ssum[c], q = (double)ssq[c];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        q += __shfl_xor(q, off);
      }
      if (lane == 0) {
        red[wave][2 * c] = s;
        red[wave][2 * c + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      partials[(((int64_t)n * gridDim.x + blockIdx.x) * COUT + (tid >> 1)) * 2 + (tid & 1)] = v;
    }
  }
}

int small_blocks(int64_t Ho, int64_t Wo, int tr, int64_t N) {
  const int64_t total = ((Ho + tr - 1) / tr) * ((Wo + 31) / 32);
  int64_t cap = 2048 / (N < 1 ? 1 : N);
  cap = cap < 64 ? 64 : cap;
  return (int)(total < cap ? total : cap);
}

template <int COUT, int KS, int STRIDE, int PPT>
int launch_small(const float* x, const float* wp, float* y, SmallGeom g, int64_t N, const float* in_scale,
                 const float* in_shift, double* partials, hipStream_t s) {
  constexpr int TR = 8 * PPT;
  g.tiles_h = (g.Ho + TR - 1) / TR;
  g.tiles_w = (g.Wo + 31) / 32;
  dim3 grid((unsigned)small_blocks(g.Ho, g.Wo, TR, N), (unsigned)N);
  hipLaunchKernelGGL((conv2d_small_kernel<COUT, KS, STRIDE, PPT>), grid, dim3(256), 0, s, x, wp, y, g, in_scale,
                     in_shift, partials);
  return pf_launch_status();
}

}  // namespace

extern "C" {

int pf_conv2d_small_blocks(int64_t N, int64_t Hi, int64_t Wi, int kernel_size, int stride) {
  if (Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int64_t Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  return small_blocks(Ho, Wo, 16, N);
}

int pf_conv2d_small_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Hi,
                        int64_t Wi, int kernel_size, int stride, const float* in_scale, const float* in_shift,
                        int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  const bool k3s1 = kernel_size == 3 && stride == 1, k5s2 = kernel_size == 5 && stride == 2;
  if (!(k3s1 || k5s2) || (Cout != 8 && Cout != 16) || Cin > 16) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cin * Hi * Wi <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  SmallGeom g;
  g.Cin = (int)Cin;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Ho = (int)((Hi - 1) / stride + 1);
  g.Wo = (int)((Wi - 1) / stride + 1);
  g.tiles_h = g.tiles_w = 0;
  g.sps = samples_per_stat;
  hipStream_t s = (hipStream_t)stream;
  if (k3s1) {
    if (Cout == 8) return launch_small<8, 3, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
    return launch_small<16, 3, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
  }
  if (Cout == 8) return launch_small<8, 5, 2, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
  return launch_small<16, 5, 2, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
}

}  // extern "C"
