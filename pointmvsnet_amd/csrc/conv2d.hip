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
#include <stdlib.h>

#include "pf_common.h"
#include "pf_bn_tail.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr size_t kMaxLds2d = 96 * 1024;

struct Conv2Geom {
  int Cin, Cout, Hi, Wi, Ho, Wo, tiles_h, tiles_w, sps;
};

// Staged patch of one input channel for a (4*TR) x 16 output tile; compile-time (see conv3d.hip, v3).
template <int STRIDE, int KS, int TR>
struct Stage2 {
  static constexpr int IH = (4 * TR - 1) * STRIDE + KS;
  static constexpr int IW = 15 * STRIDE + KS;
  static constexpr int IWP = IW + 1;
  static constexpr int RAW = IH * IWP;
  static constexpr int WANT = STRIDE == 1 ? 16 : 17;   // bank offset between channel planes (see conv3d.hip)
  static constexpr int PLANE = RAW + ((WANT - RAW % 32) + 32) % 32;
  static constexpr int ELEMS = IH * IW;               // floats of one channel's patch
  static constexpr int NXR = (ELEMS + 63) / 64;       // ... per lane of the wave that stages the channel
};

template <int NT, int STRIDE, int KS, int TR, int KG>
constexpr size_t lds_bytes_2d() {
  using S = Stage2<STRIDE, KS, TR>;
  return sizeof(float) * (size_t)(2 * KG * 4 * S::PLANE + 2 * KG * KS * KS * 4 * NT * 16 + 4 * NT * 16 * 17) +
         sizeof(double) * (size_t)(4 * NT * 16 * 2);
}

// v3 (same moves as conv3d.hip v3): compile-time patch geometry, so every LDS address of the MFMA loop is a
// base register + immediate; a step covers KG groups of 4 input channels (a 3x3 step of 4 channels is only
// 9*TR*NT MFMAs -- shorter than one global-load latency -- so two groups share a barrier); wave w stages
// channel w of each group, its lanes walking the channel's patch as one flat index (unconditional loads,
// one 32-bit offset per element per tile, wave-uniform channel base and BatchNorm affine).
template <int NT, int STRIDE, int KS, int TR, int KG, int MINW>
__global__ __launch_bounds__(256, MINW) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           float* __restrict__ y, Conv2Geom g,
                                                           const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift,
                                                           double* __restrict__ partials, PfTail tail) {
  using S = Stage2<STRIDE, KS, TR>;
  constexpr int IH = S::IH, IW = S::IW, IWP = S::IWP, PLANE = S::PLANE, ELEMS = S::ELEMS, NXR = S::NXR;
  constexpr int NCP = NT * 16;
  constexpr int KK = KS * KS;
  constexpr int PAD = KS / 2;
  constexpr int WSZ4 = KK * 4 * NCP;                 // weights of one 4-channel group
  constexpr int WSZ = KG * WSZ4;                     // ... of one step
  constexpr int NWR = (WSZ + 255) / 256;
  constexpr int XS = KG * 4 * PLANE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs0 = lds;
  float* ws0 = lds + 2 * XS;
  float* tile = ws0 + 2 * WSZ;
  double* red = reinterpret_cast<double*>(tile + 4 * NCP * 17);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int n = blockIdx.y;
  const int plane_i = g.Hi * g.Wi;                   // Cin * plane_i < 2^31 (checked on the host)
  const int64_t plane_o = (int64_t)g.Ho * g.Wo;
  const float* xb = x + (int64_t)n * g.Cin * plane_i;
  float* yb = y + (int64_t)n * g.Cout * plane_o;
  const int groups4 = (g.Cin + 3) >> 2;
  const int steps = (groups4 + KG - 1) / KG;
  const float* sc = in_scale ? in_scale + (int64_t)(n / g.sps) * g.Cin : nullptr;
  const float* sh = in_scale ? in_shift + (int64_t)(n / g.sps) * g.Cin : nullptr;

  double ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = 0.0;

  // The block walks a flat sequence of (tile, step) pairs with the NEXT pair's global loads always in
  // flight, across tile boundaries too (a tile of an 8-channel layer is a single step).
  const int total = g.tiles_h * g.tiles_w;
  unsigned gofs[NXR];                  // staging plan of the tile being LOADED
  unsigned okmask = 0;
  auto plan_tile = [&](int item) {
    const int tw = item % g.tiles_w, th = item / g.tiles_w;
    const int ih0 = th * 4 * TR * STRIDE - PAD, iw0 = tw * 16 * STRIDE - PAD;
    okmask = 0;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int e = lane + 64 * r;
      const int hy = e / IW, col = e - hy * IW;
      const int ih = ih0 + hy, iw = iw0 + col;
      const bool ok = e < ELEMS && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
      gofs[r] = ok ? (unsigned)(ih * g.Wi + iw) : 0u;
      okmask |= (ok ? 1u : 0u) << r;
    }
  };

  float rx[KG][NXR], rw[NWR];
  auto load_step = [&](int st) {
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const int c = (st * KG + kg) * 4 + wave;                        // wave-uniform channel
      const bool cok = c < g.Cin;
      const float* src = xb + (int64_t)(cok ? c : 0) * plane_i;
      float a = 1.0f, b = 0.0f;
      if (sc != nullptr && cok) {
        a = sc[c];
        b = sh[c];
      }
#pragma unroll
      for (int r = 0; r < NXR; ++r) {
        float v = src[gofs[r]];
        if (sc != nullptr) v = fmaxf(fmaf(v, a, b), 0.0f);           // previous layer's BatchNorm + ReLU
        rx[kg][r] = (cok && ((okmask >> r) & 1u)) ? v : 0.0f;         // zero padding applies AFTER it
      }
    }
    const int valid = min(KG, groups4 - st * KG) * WSZ4;              // a last, partial step: zero weights
    const float* wsrc = wp + (int64_t)st * WSZ;
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      rw[r] = e < valid ? wsrc[e] : 0.0f;
    }
  };
  auto store_step = [&](int buf) {
    float* ws = ws0 + buf * WSZ;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      float* xs = xs0 + buf * XS + (kg * 4 + wave) * PLANE;
#pragma unroll
      for (int r = 0; r < NXR; ++r) {
        const int e = lane + 64 * r;
        if (64 * (r + 1) <= ELEMS || e < ELEMS) xs[e + e / IW] = rx[kg][r];
      }
    }
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      if (256 * (r + 1) <= WSZ || e < WSZ) ws[e] = rw[r];
    }
  };

  f32x4 acc[TR][NT];
  int item = blockIdx.x, st = 0, buf = 0;
  if (item < total) {
    plan_tile(item);
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  while (item < total) {
    // the pair after (item, st)
    int n_item = item, n_st = st + 1;
    if (n_st == steps) {
      n_st = 0;
      n_item = item + gridDim.x;
    }
    const bool has_next = n_item < total;
    if (has_next) {
      if (n_st == 0) plan_tile(n_item);
      load_step(n_st);
    }
    if (st == 0) {
#pragma unroll
      for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const float* xs = xs0 + buf * XS + (kg * 4 + lk) * PLANE + (wave * TR * STRIDE) * IWP + li * STRIDE;
      const float* ws = ws0 + buf * WSZ + kg * WSZ4 + lk * NCP + li;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          float b[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) b[t] = ws[(kh * KS + kw) * 4 * NCP + 16 * t];
#pragma unroll
          for (int r = 0; r < TR; ++r) {
            const float a = xs[(r * STRIDE + kh) * IWP + kw];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[r][t], 0, 0, 0);
          }
        }
      }
    }
    if (st == steps - 1) {
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
    if (has_next) store_step(buf ^ 1);
    __syncthreads();
    item = n_item;
    st = n_st;
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
      if (tail.njobs > 0) {
        pf_row_store(o, s);
        pf_row_store(o + 1, q);
      } else {
        o[0] = s;
        o[1] = q;
      }
    }
    // this layer's BatchNorm finalize by the last block (pf_bn_tail.h); the staging buffers are free by now
    if (tail.njobs > 0) pf_bn_tail<256>(tail, n, blockIdx.x, reinterpret_cast<double*>(lds));
  }
}

// Tuning hook (microbenchmarks only): PF_CONV2D_VARIANT = 100*TR + 10*KG + MINW for the NT <= 2 kernels.
int variant2d() {
  const char* e = getenv("PF_CONV2D_VARIANT");
  return e ? atoi(e) : 0;
}

// Rows per wave.  Measured (profiles/r01j_microbench_conv2d.log): 3x3 layers with <= 16 output channels run
// best on 16 x 16 tiles (TR 4: an A operand is reused for one column tile only, so more rows per weight
// read pay) when that still leaves >= 512 tiles; 32-channel layers and the 5x5/2 layers on 8 x 16 tiles;
// 5x5 layers with 64 output channels only fit with TR = 1.
int tr_for(int ks, int nt, int64_t Ho, int64_t Wo, int64_t N) {
  const int ov = variant2d() / 100;
  if (ov && nt <= 2) return (ks == 5) ? 2 : ov;
  if (ks == 5) return nt == 4 ? 1 : 2;
  if (nt == 1 && ((Ho + 15) / 16) * ((Wo + 15) / 16) * N >= 512) return 4;
  return 2;
}

// Persistent blocks: ~6 per CU over the whole batch, so that every block streams several tiles through its
// software pipeline (a block that owns a single tile cannot hide its first memory latency).
int blocks_2d(int64_t Ho, int64_t Wo, int tr, int64_t N) {
  const int64_t total = ((Ho + 4 * tr - 1) / (4 * tr)) * ((Wo + 15) / 16);
  const char* ce = getenv("PF_CONV2D_CAP");          // tuning hook: total persistent blocks over the batch
  int64_t cap = (ce ? atoi(ce) : 1536) / (N < 1 ? 1 : N);
  cap = cap < 64 ? 64 : cap;
  if (total <= cap) return (int)total;
  // every block the same number of tiles: 640 tiles on 512 blocks would leave 3/4 of the chip idle in the
  // second round (8->16 5x5/2 layer: 43.6 -> 36.1 us with 320 blocks, profiles/r01n_microbench_conv2d_cap.log)
  const int64_t per = (total + cap - 1) / cap;
  return (int)((total + per - 1) / per);
}

struct TailArgs {
  const pf_bn_job* jobs;
  int njobs;
  unsigned* tickets;
};

template <int NT, int STRIDE, int KS, int TR, int KG, int MINW>
int launch2d(const float* x, const float* wp, float* y, Conv2Geom g, int64_t N, const float* in_scale,
             const float* in_shift, double* partials, TailArgs ta, hipStream_t s) {
  constexpr size_t lds_bytes = lds_bytes_2d<NT, STRIDE, KS, TR, KG>();
  static_assert(lds_bytes <= kMaxLds2d, "conv2d tile does not fit the LDS budget");
  if (lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};   // per instantiation, one bit per device
    const int rc = pf_allow_big_lds(reinterpret_cast<const void*>(&conv2d_kernel<NT, STRIDE, KS, TR, KG, MINW>),
                                    (int)kMaxLds2d, done);
    if (rc != PF_OK) return rc;
  }
  g.tiles_h = (g.Ho + 4 * TR - 1) / (4 * TR);
  g.tiles_w = (g.Wo + 15) / 16;
  static_assert(lds_bytes >= sizeof(double) * kTailSmemDoubles, "the BatchNorm tail borrows the staging LDS");
  dim3 grid((unsigned)blocks_2d(g.Ho, g.Wo, TR, N), (unsigned)N);
  PfTail tail;
  {
    const int rc = pf_tail_setup(tail, ta.jobs, ta.njobs, partials, (int)N, (int)grid.x, g.Cout, ta.tickets);
    if (rc != PF_OK) return rc;
  }
  hipLaunchKernelGGL((conv2d_kernel<NT, STRIDE, KS, TR, KG, MINW>), grid, dim3(256), lds_bytes, s, x, wp, y, g,
                     in_scale, in_shift, partials, tail);
  return pf_launch_status();
}

// NT <= 2: the variants the tuning hook can select (TR in {2,4} for 3x3, KG in {1,2}, MINW in {2,3,4})
template <int NT, int STRIDE, int KS>
int launch_variant(int tr, const float* x, const float* wp, float* y, const Conv2Geom& g, int64_t N,
                   const float* in_scale, const float* in_shift, double* partials, TailArgs ta, hipStream_t s) {
  const int ov = variant2d();
  const int kg = ov ? (ov / 10) % 10 : ((KS == 3 && NT == 2) ? 2 : 1);
  const int minw = ov ? ov % 10 : 2;
#define PF_L2(TRV, KGV, MW) return launch2d<NT, STRIDE, KS, TRV, KGV, MW>(x, wp, y, g, N, in_scale, in_shift, partials, ta, s)
  if constexpr (KS == 3) {
    // kg == 4: all 16 input channels of a tile staged in ONE step (one barrier per tile instead of one per four
    // channels); needs the > 64 KiB dynamic-LDS opt-in for the 16 x 16 tile
    if (kg == 4) {
      if constexpr (NT == 1) {
        if (tr == 4) PF_L2(4, 4, 2);
        PF_L2(2, 4, 2);
      } else {
        PF_L2(2, 4, 2);
      }
    }
    if (tr == 4) {
      if (kg == 2) { if (minw == 3) PF_L2(4, 2, 3); PF_L2(4, 2, 2); }
      if (minw == 3) PF_L2(4, 1, 3);
      PF_L2(4, 1, 2);
    }
    if (kg == 2) { if (minw == 4) PF_L2(2, 2, 4); if (minw == 3) PF_L2(2, 2, 3); PF_L2(2, 2, 2); }
    if (minw == 4) PF_L2(2, 1, 4);
    if (minw == 3) PF_L2(2, 1, 3);
    PF_L2(2, 1, 2);
  } else {
    if constexpr (NT == 1) {
      if (kg == 2) PF_L2(2, 2, 2);             // 8 -> 16, 5x5/2: both channel groups in one step
    }
    if (minw == 3) PF_L2(2, 1, 3);
    PF_L2(2, 1, 2);
  }
#undef PF_L2
}

}  // namespace

extern "C" {

int pf_conv2d_blocks(int64_t N, int64_t Cout, int64_t Hi, int64_t Wi, int kernel_size, int stride) {
  if (Cout <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int nt = (int)((Cout + 15) / 16);
  const int64_t Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  return blocks_2d(Ho, Wo, tr_for(kernel_size, nt == 3 ? 4 : nt, Ho, Wo, N), N);
}

int pf_conv2d_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Hi,
                  int64_t Wi, int kernel_size, int stride, const float* in_scale, const float* in_shift,
                  int samples_per_stat, double* partials, const pf_bn_job* bn_jobs, int n_bn_jobs, unsigned* tickets,
                  void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE(n_bn_jobs >= 0 && (n_bn_jobs == 0 || partials != nullptr));
  const TailArgs ta{bn_jobs, n_bn_jobs, tickets};
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
  const int tr = tr_for(kernel_size, nt, g.Ho, g.Wo, N);
  if (k3s1) {
    if (nt == 1) return launch_variant<1, 1, 3>(tr, x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
    if (nt == 2) return launch_variant<2, 1, 3>(tr, x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
    return launch2d<4, 1, 3, 2, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
  }
  if (nt == 1) return launch_variant<1, 2, 5>(tr, x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
  if (nt == 2) return launch_variant<2, 2, 5>(tr, x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
  return launch2d<4, 2, 5, 1, 1, 2>(x, wp, y, g, N, in_scale, in_shift, partials, ta, s);
}

}  // extern "C"
