// ImageConv, the few-channel layers: 3 -> 8 and 8 -> 8 at full resolution,
// 8 -> 16 with a 5x5 / stride-2 kernel.  These are HBM-bound (72..200 multiply-adds per input float, 32-64 B
// moved per pixel) and took 62-86 us each in the library's Winograd kernel plus two BatchNorm passes
// (profiles/r01c_last_step_dispatches.txt lines 41-49); an MFMA tile is half empty at 8 output channels and
// instruction-bound (conv2d.hip).  So: plain float32 FMAs.
//
//   * one lane = one output column (30 per tile), PPT output rows; all C_out accumulators in registers;
//   * the block's input patch is staged in LDS four channels at a time (the previous layer's
//     BatchNorm+ReLU applied on the way in, zero outside the image); the group's weights are wave-uniform
//     and come through the scalar cache ([ch][kh][kw][C_out], s_load into SGPRs), reused for the PPT rows;
//   * stores are 128-byte row segments per channel straight from registers; per-channel sum / sum of
//     squares are carried in registers across the block's tiles and reduced once (float64 partials).
#include "pf_common.h"
#include "pf_bn_tail.h"

namespace {

struct SmallGeom {
  int Cin, Hi, Wi, Ho, Wo, tiles_h, tiles_w, sps;
};

constexpr int TC = 30;   // output columns per tile: the staged patch is then 32 (3x3/s1) or 63 (5x5/s2) floats wide

template <int COUT, int KS, int STRIDE, int PPT>
__global__ __launch_bounds__(256) void conv2d_small_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           float* __restrict__ y, SmallGeom g,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           double* __restrict__ partials, PfTail tail) {
  constexpr int TR = 8 * PPT;                          // output tile rows
  constexpr int IH = (TR - 1) * STRIDE + KS, IW = (TC - 1) * STRIDE + KS;
  constexpr int IWP = IW | 1;                          // odd row stride
  constexpr int PAD = KS / 2;
  constexpr int WSZ = 4 * KS * KS * COUT;
  constexpr int ROWS = 4 * IH;
  constexpr int NXR = (ROWS + 7) / 8;                  // staged rows per lane-row (8 rows of 32 lanes per pass)
  constexpr int XPASS = (IW + 31) / 32;
  constexpr int XS = 4 * IH * IWP;
  constexpr bool DB = 2 * XS * 4 <= 40 * 1024;         // double-buffer the patch when it is small enough
  __shared__ __attribute__((aligned(16))) float xs[(DB ? 2 : 1) * XS];
  __shared__ double red[4][2 * COUT];
  const int tid = threadIdx.x;
  const int col = tid & 31, row = tid >> 5;            // 32 columns x 8 rows of lanes
  const int ccol = col < TC ? col : 0;                 // lanes 30, 31 only help staging
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

  // Persistent block, flat sequence of (tile, channel group) steps; the NEXT step's patch is loaded into
  // registers while the current one is computed, across tile boundaries too.  (SQ counters on the first
  // version -- one tile per block, load / sync / compute / sync -- showed the VALU busy only 50 % of the
  // time and 45 % of the wave cycles in s_waitcnt / barriers: profiles/r01d_conv2d_small_sq_counters_*.)
  const int total = g.tiles_h * g.tiles_w;
  float rx[NXR * XPASS];
  auto load_step = [&](int item, int cg) {
    const int tw = item % g.tiles_w, th = item / g.tiles_w;
    const int ih0 = th * TR * STRIDE - PAD, iw0 = tw * TC * STRIDE - PAD;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int srow = r * 8 + row;
      const int ch = srow / IH, yy = srow - ch * IH;
      const int c = cg * 4 + ch, ih = ih0 + yy;
      const bool rok = srow < ROWS && c < g.Cin && ih >= 0 && ih < g.Hi;
      const float* src = xb + (int64_t)(rok ? c : 0) * plane_i + (int64_t)(rok ? ih : 0) * g.Wi;
      float a = 1.0f, b = 0.0f;
      if (sc != nullptr && rok) {
        a = sc[c];
        b = sh[c];
      }
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int cc = col + 32 * p, iw = iw0 + cc;
        const bool ok = rok && cc < IW && iw >= 0 && iw < g.Wi;
        float v = ok ? src[iw] : 0.0f;
        if (sc != nullptr && ok) v = fmaxf(fmaf(v, a, b), 0.0f);           // previous BatchNorm + ReLU
        rx[r * XPASS + p] = v;
      }
    }
  };
  auto store_step = [&](int buf) {
    float* dst = xs + buf * XS;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int srow = r * 8 + row;
      if (srow < ROWS) {
        const int ch = srow / IH, yy = srow - ch * IH;
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
          const int cc = col + 32 * p;
          if (cc < IW) dst[(ch * IH + yy) * IWP + cc] = rx[r * XPASS + p];
        }
      }
    }
  };

  float acc[PPT][COUT];
  int item = blockIdx.x, cg = 0, buf = 0;
  if (item < total) {
    load_step(item, 0);
    store_step(0);
  }
  __syncthreads();
  while (item < total) {
    int n_item = item, n_cg = cg + 1;
    if (n_cg == cgroups) {
      n_cg = 0;
      n_item = item + gridDim.x;
    }
    const bool has_next = n_item < total;
    if (has_next) load_step(n_item, n_cg);               // in flight during the FMAs below
    if (cg == 0) {
#pragma unroll
      for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[p][c] = 0.0f;
    }
    {
      // weights are wave-uniform: scalar cache -> SGPRs (in LDS the kernel was LDS-bandwidth bound)
      const float* __restrict__ wg = wp + (int64_t)cg * WSZ;
      const float* xsb = xs + buf * XS;
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
              const float v = xsb[(ch * IH + (row + 8 * p) * STRIDE + kh) * IWP + ccol * STRIDE + kw];
#pragma unroll
              for (int c = 0; c < COUT; ++c) acc[p][c] = fmaf(v, w[c], acc[p][c]);
            }
          }
        }
      }
    }
    if (cg == cgroups - 1) {
      const int tw = item % g.tiles_w, th = item / g.tiles_w;
      const int ow = tw * TC + col;
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        const int oh = th * TR + row + 8 * p;
        if (col < TC && oh < g.Ho && ow < g.Wo) {
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
    if (!DB) __syncthreads();                             // single buffer: everyone is done reading it
    if (has_next) store_step(DB ? (buf ^ 1) : 0);
    __syncthreads();
    item = n_item;
    cg = n_cg;
    if (DB) buf ^= 1;
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
      double* o = partials + (((int64_t)n * gridDim.x + blockIdx.x) * COUT + (tid >> 1)) * 2 + (tid & 1);
      if (tail.njobs > 0) pf_row_store(o, v);
      else *o = v;
    }
    // this layer's BatchNorm finalize by the last block (pf_bn_tail.h); the patch buffer is free by now
    static_assert(sizeof(xs) >= sizeof(double) * kTailSmemDoubles, "the BatchNorm tail borrows the patch buffer");
    if (tail.njobs > 0) pf_bn_tail<256>(tail, n, blockIdx.x, reinterpret_cast<double*>(xs));
  }
}

// Persistent blocks (about four per CU over the whole batch) so that every block streams several tiles.
int small_blocks(int64_t Ho, int64_t Wo, int tr, int64_t N) {
  const int64_t total = ((Ho + tr - 1) / tr) * ((Wo + TC - 1) / TC);
  int64_t cap = 1024 / (N < 1 ? 1 : N);
  cap = cap < 64 ? 64 : cap;
  return (int)(total < cap ? total : cap);
}

template <int COUT, int KS, int STRIDE, int PPT>
int launch_small(const float* x, const float* wp, float* y, SmallGeom g, int64_t N, const float* in_scale,
                 const float* in_shift, double* partials, const pf_bn_job* jobs, int njobs, unsigned* tickets,
                 hipStream_t s) {
  constexpr int TR = 8 * PPT;
  g.tiles_h = (g.Ho + TR - 1) / TR;
  g.tiles_w = (g.Wo + TC - 1) / TC;
  dim3 grid((unsigned)small_blocks(g.Ho, g.Wo, TR, N), (unsigned)N);
  PfTail tail;
  {
    const int rc = pf_tail_setup(tail, jobs, njobs, partials, (int)N, (int)grid.x, COUT, tickets);
    if (rc != PF_OK) return rc;
  }
  hipLaunchKernelGGL((conv2d_small_kernel<COUT, KS, STRIDE, PPT>), grid, dim3(256), 0, s, x, wp, y, g, in_scale,
                     in_shift, partials, tail);
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
                        int samples_per_stat, double* partials, const pf_bn_job* bn_jobs, int n_bn_jobs,
                        unsigned* tickets, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE(n_bn_jobs >= 0 && (n_bn_jobs == 0 || partials != nullptr));
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
    if (Cout == 8) return launch_small<8, 3, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, bn_jobs, n_bn_jobs, tickets, s);
    return launch_small<16, 3, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, bn_jobs, n_bn_jobs, tickets, s);
  }
  if (Cout == 8) return launch_small<8, 5, 2, 2>(x, wp, y, g, N, in_scale, in_shift, partials, bn_jobs, n_bn_jobs, tickets, s);
  return launch_small<16, 5, 2, 2>(x, wp, y, g, N, in_scale, in_shift, partials, bn_jobs, n_bn_jobs, tickets, s);
}

}  // extern "C"
