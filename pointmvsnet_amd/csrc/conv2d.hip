// ImageConv ("the step before the path"): KxK / pad K/2 / stride 1|2 conv2d
// (K = 3 or 5) as an implicit GEMM on the f32 matrix cores, with the previous layer's BatchNorm+ReLU
// applied while the input is staged and this layer's BatchNorm batch statistics produced in the
// epilogue.  Same structure as conv3d.hip (read that header first); what is specific here:
//
//   * the feature towers run 2 x V times per depth map (reference model.py:71-77, :132-141) and are, after
//     the PointFlow kernels, 40 % of the step: the library path costs one Winograd kernel (~40 us at these
//     small channel counts), one statistics pass and one normalise pass per layer.  Here a layer is ONE
//     launch plus the tiny finalize: the raw (pre-BN) convolution output is written once and read once;
//   * per-view BatchNorm: the V views of a scene are batched along N, `samples_per_stat` consecutive
//     samples share one statistic (== one reference module call), in_scale/in_shift are (N/sps, Cin);
//   * a 256-thread block owns (4*TR) x 16 output pixels; wave w owns TR rows; K is walked in groups of
//     4 input channels (C_in = 3 is zero-padded to 4), double buffered through LDS.
// Bound: fp32 MFMA for the wide layers, HBM (4*(C_in+C_out) B/pixel) for the 3->8 / 8->8 full-resolution ones.
#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr size_t kMaxLds2d = 96 * 1024;

struct Conv2Geom {
  int Cin, Cout, Hi, Wi, Ho, Wo, tiles_h, tiles_w, sps;
};

template <int STRIDE, int KS, int TR>
struct Stage2 {
  static constexpr int IH = (4 * TR - 1) * STRIDE + KS;
  static constexpr int IW = 15 * STRIDE + KS;
  static constexpr int IWP = IW + 1;
  static constexpr int RAW = IH * IWP;
  static constexpr int WANT = STRIDE == 1 ? 16 : 17;   // bank offset between channel planes (see conv3d.hip)
  static constexpr int PLANE = RAW + ((WANT - RAW % 32) + 32) % 32;
  static constexpr int ROWS = 4 * IH;
  static constexpr int NXR = (ROWS + 7) / 8;
  static constexpr int XPASS = (IW + 31) / 32;
};

template <int NT, int STRIDE, int KS, int TR>
constexpr size_t lds_bytes_2d() {
  using S = Stage2<STRIDE, KS, TR>;
  return sizeof(float) * (size_t)(2 * 4 * S::PLANE + 2 * KS * KS * 4 * NT * 16 + 4 * NT * 16 * 17) +
         sizeof(double) * (size_t)(4 * NT * 16 * 2);
}

template <int NT, int STRIDE, int KS, int TR>
__global__ __launch_bounds__(256) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     float* __restrict__ y, Conv2Geom g,
                                                     const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift,
                                                     double* __restrict__ partials) {
  using S = Stage2<STRIDE, KS, TR>;
  constexpr int NCP = NT * 16;
  constexpr int KK = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int WSZ = KK * 4 * NCP;
  constexpr int NWR = (WSZ + 255) / 256;
  constexpr int NXR = S::NXR, XPASS = S::XPASS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int XS = 4 * S::PLANE;
  float* xs0 = lds;
  float* ws0 = lds + 2 * XS;
  float* tile = ws0 + 2 * WSZ;
  double* red = reinterpret_cast<double*>(tile + 4 * NCP * 17);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int n = blockIdx.y;
  const int64_t plane_i = (int64_t)g.Hi * g.Wi;
  const int64_t plane_o = (int64_t)g.Ho * g.Wo;
  const float* xb = x + (int64_t)n * g.Cin * plane_i;
  float* yb = y + (int64_t)n * g.Cout * plane_o;
  const int cgroups = (g.Cin + 3) >> 2;
  const int srow = tid >> 5, scol = tid & 31;
  const float* sc = in_scale ? in_scale + (int64_t)(n / g.sps) * g.Cin : nullptr;
  const float* sh = in_scale ? in_shift + (int64_t)(n / g.sps) * g.Cin : nullptr;

  double ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = 0.0;

  // The block walks a flat sequence of (tile, channel group) steps with the NEXT step's global loads
  // always in flight, across tile boundaries too: with C_in = 8 a tile has only two groups, so a per-tile
  // prologue would expose one full memory latency per tile (measured: 58 us vs a 12 us HBM floor).
  const int total = g.tiles_h * g.tiles_w;
  int gofs[NXR], lofs[NXR];            // staging plan of the tile being LOADED (row = ch * IH + hy)
  int l_iw0 = 0;
  auto plan_tile = [&](int item) {
    const int tw = item % g.tiles_w, th = item / g.tiles_w;
    const int ih0 = th * 4 * TR * STRIDE - PAD;
    l_iw0 = tw * 16 * STRIDE - PAD;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int row = r * 8 + srow;
      const int ch = row / S::IH, hy = row - ch * S::IH;
      const int ih = ih0 + hy;
      const bool in = row < S::ROWS;
      gofs[r] = (in && ih >= 0 && ih < g.Hi) ? (int)((int64_t)ch * plane_i + (int64_t)ih * g.Wi) : -1;
      lofs[r] = in ? ch * S::PLANE + hy * S::IWP : -1;
    }
  };

  float rx[NXR * XPASS], rw[NWR];
  auto load_group = [&](int cg) {
    const float* src = xb + (int64_t)cg * 4 * plane_i;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int c = cg * 4 + (r * 8 + srow) / S::IH;
      const bool cok = gofs[r] >= 0 && c < g.Cin;
      float a = 1.0f, b = 0.0f;
      if (sc != nullptr && cok) {
        a = sc[c];
        b = sh[c];
      }
#pragma unroll
      for (int p = 0; p < XPASS; ++p) {
        const int col = scol + 32 * p;
        const int iw = l_iw0 + col;
        float v = 0.0f;
        if (cok && col < S::IW && iw >= 0 && iw < g.Wi) {
          v = src[gofs[r] + iw];
          if (sc != nullptr) v = fmaxf(fmaf(v, a, b), 0.0f);         // previous layer's BatchNorm + ReLU
        }
        rx[r * XPASS + p] = v;
      }
    }
    const float* wsrc = wp + (int64_t)cg * WSZ;
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      rw[r] = e < WSZ ? wsrc[e] : 0.0f;
    }
  };
  auto store_group = [&](int buf) {
    float* xs = xs0 + buf * XS;
    float* ws = ws0 + buf * WSZ;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      if (lofs[r] >= 0) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
          const int col = scol + 32 * p;
          if (col < S::IW) xs[lofs[r] + col] = rx[r * XPASS + p];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      if (e < WSZ) ws[e] = rw[r];
    }
  };

  f32x4 acc[TR][NT];
  int item = blockIdx.x, cg = 0, buf = 0;
  if (item < total) {
    plan_tile(item);
    load_group(0);
    store_group(0);
  }
  __syncthreads();
  while (item < total) {
    // the step after (item, cg)
    int n_item = item, n_cg = cg + 1;
    if (n_cg == cgroups) {
      n_cg = 0;
      n_item = item + gridDim.x;
    }
    const bool has_next = n_item < total;
    if (has_next) {
      if (n_cg == 0) plan_tile(n_item);
      load_group(n_cg);
    }
    if (cg == 0) {
#pragma unroll
      for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    {
      const float* xs = xs0 + buf * XS + lk * S::PLANE + (wave * TR * STRIDE) * S::IWP + li * STRIDE;
      const float* ws = ws0 + buf * WSZ + lk * NCP + li;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          float b[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) b[t] = ws[(kh * KS + kw) * 4 * NCP + 16 * t];
#pragma unroll
          for (int r = 0; r < TR; ++r) {
            const float a = xs[(r * STRIDE + kh) * S::IWP + kw];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[r][t], 0, 0, 0);
          }
        }
      }
    }
    if (cg == cgroups - 1) {
      // epilogue of tile `item`: per-wave LDS transpose -> 64-byte segments per channel, BN statistics
      const int tw = item % g.tiles_w, th = item / g.tiles_w;
      const int oh0 = th * 4 * TR, ow0 = tw * 16;
      float* tl = tile + wave * NCP * 17;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int oh = oh0 + wave * TR + r;
        const bool row_ok = oh < g.Ho;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float s = 0.0f, q = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int pos = lk * 4 + e;
            const float v = acc[r][t][e];
            tl[(16 * t + li) * 17 + pos] = v;
            if (row_ok && ow0 + pos < g.Wo) {
              s += v;
              q += v * v;
            }
          }
          s += __shfl_xor(s, 16);
          q += __shfl_xor(q, 16);
          s += __shfl_xor(s, 32);
          q += __shfl_xor(q, 32);
          ssum[t] += (double)s;
          ssq[t] += (double)q;
        }
        __builtin_amdgcn_wave_barrier();
        if (row_ok) {
          for (int e = lane; e < NCP * 16; e += 64) {
            const int co = e >> 4, pos = e & 15;
            if (co < g.Cout && ow0 + pos < g.Wo)
              yb[(int64_t)co * plane_o + (int64_t)oh * g.Wo + ow0 + pos] = tl[co * 17 + pos];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (has_next) store_group(buf ^ 1);
    __syncthreads();
    item = n_item;
    cg = n_cg;
    buf ^= 1;
  }

  if (partials != nullptr) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (lane < 16) {
        red[((wave * NCP) + 16 * t + lane) * 2 + 0] = ssum[t];
        red[((wave * NCP) + 16 * t + lane) * 2 + 1] = ssq[t];
      }
    }
    __syncthreads();
    if (tid < g.Cout) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[((w * NCP) + tid) * 2 + 0];
        q += red[((w * NCP) + tid) * 2 + 1];
      }
      double* o = partials + (((int64_t)n * gridDim.x + blockIdx.x) * g.Cout + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

int tr_for(int ks, int nt) { return (ks == 5 && nt == 4) ? 1 : 2; }

// Persistent blocks: ~6 per CU over the whole batch, so that every block streams several tiles through its
// software pipeline (a block that owns a single tile cannot hide its first memory latency).
int blocks_2d(int64_t Ho, int64_t Wo, int tr, int64_t N) {
  const int64_t total = ((Ho + 4 * tr - 1) / (4 * tr)) * ((Wo + 15) / 16);
  int64_t cap = 1536 / (N < 1 ? 1 : N);
  cap = cap < 64 ? 64 : cap;
  return (int)(total < cap ? total : cap);
}

template <int NT, int STRIDE, int KS, int TR>
int launch2d(const float* x, const float* wp, float* y, Conv2Geom g, int64_t N, const float* in_scale,
             const float* in_shift, double* partials, hipStream_t s) {
  constexpr size_t lds_bytes = lds_bytes_2d<NT, STRIDE, KS, TR>();
  static_assert(lds_bytes <= kMaxLds2d, "conv2d tile does not fit the LDS budget");
  if (lds_bytes > 64 * 1024) {
    static bool done = false;
    if (!done) {
      PF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_kernel<NT, STRIDE, KS, TR>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds2d));
      done = true;
    }
  }
  g.tiles_h = (g.Ho + 4 * TR - 1) / (4 * TR);
  g.tiles_w = (g.Wo + 15) / 16;
  dim3 grid((unsigned)blocks_2d(g.Ho, g.Wo, TR, N), (unsigned)N);
  hipLaunchKernelGGL((conv2d_kernel<NT, STRIDE, KS, TR>), grid, dim3(256), lds_bytes, s, x, wp, y, g, in_scale,
                     in_shift, partials);
  return pf_launch_status();
}

}  // namespace

extern "C" {

int pf_conv2d_blocks(int64_t N, int64_t Cout, int64_t Hi, int64_t Wi, int kernel_size, int stride) {
  if (Cout <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int nt = (int)((Cout + 15) / 16);
  const int64_t Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  return blocks_2d(Ho, Wo, tr_for(kernel_size, nt == 3 ? 4 : nt), N);
}

int pf_conv2d_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Hi,
                  int64_t Wi, int kernel_size, int stride, const float* in_scale, const float* in_shift,
                  int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  const bool k3s1 = kernel_size == 3 && stride == 1, k5s2 = kernel_size == 5 && stride == 2;
  if (!(k3s1 || k5s2) || Cout > 64) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cin * Hi * Wi <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  Conv2Geom g;
  g.Cin = (int)Cin;
  g.Cout = (int)Cout;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Ho = (int)((Hi - 1) / stride + 1);
  g.Wo = (int)((Wi - 1) / stride + 1);
  g.tiles_h = g.tiles_w = 0;
  g.sps = samples_per_stat;
  hipStream_t s = (hipStream_t)stream;
  int nt = (int)((Cout + 15) / 16);
  if (nt == 3) nt = 4;
  if (k3s1) {
    if (nt == 1) return launch2d<1, 1, 3, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
    if (nt == 2) return launch2d<2, 1, 3, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
    return launch2d<4, 1, 3, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
  }
  if (nt == 1) return launch2d<1, 2, 5, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
  if (nt == 2) return launch2d<2, 2, 5, 2>(x, wp, y, g, N, in_scale, in_shift, partials, s);
  return launch2d<4, 2, 5, 1>(x, wp, y, g, N, in_scale, in_shift, partials, s);
}

}  // extern "C"
