// Row R, the bottom of VolumeConv's U-Net (reference networks.py:136-141: conv3_0 32 -> 64 stride 2, conv3_1
// 64 -> 64, conv4_0 = ConvTranspose3d 64 -> 32 stride 2) on volumes of a few thousand voxels (6 x 16 x 20 at
// BASELINE config 2).  The library lowers each of them to im2col + GEMM (+ col2im) and needs a separate BatchNorm
// pass: 9 dependent graph nodes, 62 us, for 0.85 GFLOP.  Here each layer is ONE launch:
//
//   * v_mfma_f32_16x16x4_f32 with the reduction index assigned to (MFMA step, lane quarter) as in conv2d_wide.hip:
//     lane (i = lane & 15, kq = lane >> 4) is voxel i of a 4 x 4 patch of one output depth (A operand) and output
//     channel i of the wave's 16 (B operand), and reads 16 bytes = channels 16 kc + 4 kq + {0..3} per operand;
//   * a block = 16 output voxels x 64 channels (conv) -- 120 blocks x 4 waves for 1 920 voxels: no split-K, every
//     output is one exact f32 fmaf chain, and still a wave for half of the chip's SIMDs;
//   * the input patch (3 depths x 6 x 6 or 9 x 9 voxels, channel-last, <= 36 KB) sits in LDS with the PREVIOUS
//     layer's BatchNorm + ReLU applied while it is staged -- from (scale, shift) rows or resolved by the block itself
//     from the producer's statistics rows (pf_bn_resolve, pf_bn_resolve.h); the weights never touch LDS: each is used by
//     one wave of a block only, so every lane streams its own 16-byte pieces from L2 through a ring of 32 registers
//     (the kernels are bound by that stream's latency, not by the 0.1 GFLOP of matrix work);
//   * the transposed convolution is the same GEMM with the 2 x 2 x 2 input neighbourhood as reduction index and
//     (output parity class, channel) as columns; only the 27 non-zero (neighbour, class) blocks are executed, the
//     classes are dealt to the four waves by work (8 | 4+2 | 4+2 | 4+2+1 neighbour blocks), and the weights -- used
//     once per block -- go from L2 straight into registers;
//   * BatchNorm statistics of the output in the epilogue (float64 partial rows, 120 per sample).
// Bound: latency (a block is 3-6 us of dependent MFMA work); algorithmic bytes 4 (C_in V_in + C_out V_out) + weights.
#include <stdlib.h>

#include "pf_common.h"
#include "pf_bn_resolve.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BottomGeom {
  int Di, Hi, Wi, Do, Ho, Wo, tiles_w, sps;
};

// The staged patch: PD x PH x PW voxels, channel-last rows of RS floats; NC(D)HW planes -> LDS with the pending
// BatchNorm + ReLU on the way (AFFINE as in conv2d_wide.hip: 0 none, 1 rows, 2 resolved here); zero padding AFTER it.
template <int CIN, int PD, int PH, int PW, int RS, int AFFINE>
__device__ __forceinline__ void stage_patch3d(const float* __restrict__ xb, int64_t plane_c, int plane_d, int id0,
                                              int ih0, int iw0, int dstep, int Di, int Hi, int Wi, float* patch,
                                              float* aff, const float* __restrict__ in_scale,
                                              const float* __restrict__ in_shift, int stat, const pf_bn_job& in_bn,
                                              double* scratch) {
  constexpr int NPIX = PD * PH * PW;
  constexpr int ITEMS = NPIX * (CIN / 4);
  constexpr int NIT = (ITEMS + 255) / 256;
  const int tid = threadIdx.x;
  float rx[NIT][4];
  bool rok[NIT];
#pragma unroll
  for (int r = 0; r < NIT; ++r) {
    const int it = tid + 256 * r;
    const int itc = it < ITEMS ? it : ITEMS - 1;
    const int q = itc / NPIX, p = itc - q * NPIX;
    const int pd = p / (PH * PW), pp = p - pd * (PH * PW);
    const int pr = pp / PW, pc = pp - pr * PW;
    const int id = id0 + pd * dstep, ih = ih0 + pr, iw = iw0 + pc;
    rok[r] = id >= 0 && id < Di && ih >= 0 && ih < Hi && iw >= 0 && iw < Wi;
    const float* src = xb + (int64_t)(4 * q) * plane_c + (rok[r] ? id * plane_d + ih * Wi + iw : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) rx[r][j] = src[j * plane_c];
  }
  if (AFFINE == 1) {
    const float* sc = in_scale + (int64_t)stat * CIN;
    const float* sh = in_shift + (int64_t)stat * CIN;
    if (tid < CIN) aff[tid] = sc[tid];
    else if (tid < 2 * CIN) aff[tid] = sh[tid - CIN];
    __syncthreads();
  }
  if (AFFINE == 2) pf_bn_resolve<256>(in_bn, stat, aff, aff + CIN, scratch);
#pragma unroll
  for (int r = 0; r < NIT; ++r) {
    const int it = tid + 256 * r;
    const int itc = it < ITEMS ? it : ITEMS - 1;
    const int q = itc / NPIX, p = itc - q * NPIX;
    f32x4 v = {rx[r][0], rx[r][1], rx[r][2], rx[r][3]};
    if (AFFINE) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(aff + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(aff + CIN + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], a[j], b[j]), 0.0f);
    }
    if (!rok[r]) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    if (256 * (r + 1) <= ITEMS || it < ITEMS) *reinterpret_cast<f32x4*>(patch + p * RS + 4 * q) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3x3 / pad 1 / stride 1|2 convolution, C_out = 64
// ------------------------------------------------------------------------------------------------
template <int STRIDE, int CIN>
struct BottomCfg {
  static constexpr int COUT = 64;
  static constexpr int PH = 3 * STRIDE + 3, PW = PH;   // input rows / columns behind a 4 x 4 output patch
  static constexpr int RS = CIN + 4;
  static constexpr int PATCH = 3 * PH * PW * RS;
  static constexpr int KC = CIN / 16;
  static constexpr int STEPS = 27 * KC;                // 16-byte operand pairs of a wave: [tap][kc]
  static constexpr int PF = 32;                        // weight pieces in flight per lane (~2 us of L2 latency)
  static constexpr size_t LDS = sizeof(float) * (size_t)(PATCH + 1024 + 2 * CIN);
  static_assert(CIN % 16 == 0 && LDS <= 64 * 1024 && STEPS > PF, "shape");
};

// Every weight is used by exactly ONE wave of a block (wave w owns output channels 16 w .. 16 w + 15), so the B
// operands never touch LDS: each lane streams its own 16-byte pieces from L2 through a ring of PF registers, the
// load of step s + PF issued when step s is consumed -- no barrier after the patch is staged.  (A first version
// staged one (kd, kh) row of weights per barrier through LDS, double buffered: 18 us for conv3_1 on 6 x 8 x 10,
// each of the 9 rows waiting ~1.8 us for 49 KB per block against 0.6 us of MFMA work,
// profiles/archive/r02/r02ai_unet_bottom.txt.)
template <int STRIDE, int CIN, int AFFINE>
__global__ __launch_bounds__(256) void conv3d_bottom_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            float* __restrict__ y, BottomGeom g,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            double* __restrict__ partials, pf_bn_job in_bn) {
  using C = BottomCfg<STRIDE, CIN>;
  constexpr int PH = C::PH, PW = C::PW, RS = C::RS, COUT = 64, KC = C::KC, STEPS = C::STEPS, PF = C::PF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;
  double* scratch = reinterpret_cast<double*>(lds + C::PATCH);
  float* aff = lds + C::PATCH + 1024;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int n = blockIdx.z, od = blockIdx.y;
  const int tw = blockIdx.x % g.tiles_w, th = blockIdx.x / g.tiles_w;
  const int oh0 = 4 * th, ow0 = 4 * tw;
  const int plane_d = g.Hi * g.Wi;
  const int64_t plane_c = (int64_t)g.Di * plane_d;
  const float* xb = x + (int64_t)n * CIN * plane_c;

  // packed weights [kd][kh][kw][kc][kq][c_out][4]: step s = (tap, kc) of this lane is piece wg[s * 4 * COUT]
  const f32x4* wg = reinterpret_cast<const f32x4*>(wp) + kq * COUT + 16 * wave + li;
  f32x4 ring[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) ring[s] = wg[s * 4 * COUT];

  stage_patch3d<CIN, 3, PH, PW, RS, AFFINE>(xb, plane_c, plane_d, od * STRIDE - 1, oh0 * STRIDE - 1, ow0 * STRIDE - 1, 1,
                                            g.Di, g.Hi, g.Wi, patch, aff, in_scale, in_shift, n / g.sps, in_bn, scratch);
  __syncthreads();

  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  const float* abase = patch + (((li >> 2) * STRIDE) * PW + (li & 3) * STRIDE) * RS + 4 * kq;
  auto a_ptr = [&](int s) {
    const int tap = s / KC, kc = s - tap * KC;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    return abase + ((kd * PH + kh) * PW + kw) * RS + 16 * kc;
  };
  f32x4 a = *reinterpret_cast<const f32x4*>(a_ptr(0));
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const f32x4 b = ring[s % PF];
    if (s + PF < STEPS) ring[s % PF] = wg[(s + PF) * 4 * COUT];
    f32x4 an = a;
    if (s + 1 < STEPS) an = *reinterpret_cast<const f32x4*>(a_ptr(s + 1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
  }

  // C/D layout: column (channel) = lane & 15, rows (voxels of the 4 x 4 patch) 4 kq + {0..3} = row kq, columns 0..3
  const int co = 16 * wave + li;
  const int oh = oh0 + kq;
  float s = 0.0f, q = 0.0f;
  if (oh < g.Ho) {
    float* dst = y + (((int64_t)n * COUT + co) * g.Do + od) * ((int64_t)g.Ho * g.Wo) + (int64_t)oh * g.Wo + ow0;
    if ((g.Wo & 3) == 0) {
      *reinterpret_cast<f32x4*>(dst) = acc;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += acc[e];
        q += acc[e] * acc[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ow0 + e < g.Wo) {
          dst[e] = acc[e];
          s += acc[e];
          q += acc[e] * acc[e];
        }
    }
  }
  if (partials != nullptr) {
    s += __shfl_xor(s, 16);
    q += __shfl_xor(q, 16);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lane < 16) {
      double* o = partials + ((((int64_t)n * gridDim.y + od) * gridDim.x + blockIdx.x) * COUT + co) * 2;
      o[0] = (double)s;
      o[1] = (double)q;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose3d 3x3x3 / stride 2 / pad 1 / output_padding 1, 64 -> 32
// ------------------------------------------------------------------------------------------------
// Output o = 2 i - 1 + k per dimension: parity 0 takes (neighbour offset 0, tap 1); parity 1 takes (offset 1, tap 0)
// and (offset 0, tap 2).  Class c = (pd, ph, pw) has (1 + pd)(1 + ph)(1 + pw) neighbour blocks of 64 x 32 weights.
struct DeconvCfg {
  static constexpr int CIN = 64, COUT = 32, RS = CIN + 4, KC = CIN / 16;
  static constexpr int PATCH = 2 * 5 * 5 * RS;
  static constexpr size_t LDS = sizeof(float) * (size_t)(PATCH + 1024 + 2 * CIN);
};

// The (class, neighbour) blocks of each wave, in execution order: class bits (pd ph pw), neighbour bits (sd sh sw)
// and "the class is complete after this item".  8 | 4+2 | 4+2 | 4+2+1 items.
struct DcItem {
  unsigned char cls, nb, end;
};
__constant__ DcItem kDcItems[4][8] = {
    {{7, 0, 0}, {7, 1, 0}, {7, 2, 0}, {7, 3, 0}, {7, 4, 0}, {7, 5, 0}, {7, 6, 0}, {7, 7, 1}},
    {{3, 0, 0}, {3, 1, 0}, {3, 2, 0}, {3, 3, 1}, {1, 0, 0}, {1, 1, 1}, {0, 0, 0}, {0, 0, 0}},
    {{5, 0, 0}, {5, 1, 0}, {5, 4, 0}, {5, 5, 1}, {2, 0, 0}, {2, 2, 1}, {0, 0, 0}, {0, 0, 0}},
    {{6, 0, 0}, {6, 2, 0}, {6, 4, 0}, {6, 6, 1}, {4, 0, 0}, {4, 4, 1}, {0, 0, 1}, {0, 0, 0}}};
__constant__ int kDcCount[4] = {8, 6, 6, 7};

__device__ __forceinline__ int deconv_tap(int parity, int offset) { return parity == 0 ? 1 : (offset == 1 ? 0 : 2); }

template <int AFFINE>
__global__ __launch_bounds__(256) void deconv3d_bottom_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                              float* __restrict__ y, BottomGeom g,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift,
                                                              double* __restrict__ partials, pf_bn_job in_bn) {
  using C = DeconvCfg;
  constexpr int RS = C::RS, CIN = C::CIN, COUT = C::COUT, KC = C::KC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;
  double* scratch = reinterpret_cast<double*>(lds + C::PATCH);     // 4 KB for pf_bn_resolve
  float* aff = lds + C::PATCH + 1024;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  const int n = blockIdx.z, id = blockIdx.y;
  const int tw = blockIdx.x % g.tiles_w, th = blockIdx.x / g.tiles_w;
  const int ih0 = 4 * th, iw0 = 4 * tw;
  const int plane_d = g.Hi * g.Wi;
  const int64_t plane_c = (int64_t)g.Di * plane_d;
  const float* xb = x + (int64_t)n * CIN * plane_c;

  // weights of an item: [tap][kc][kq][c_out 32][4] floats, straight from L2 (used once per block), two items ahead
  const f32x4* w4 = reinterpret_cast<const f32x4*>(wp) + kq * COUT + li;
  const int nitems = kDcCount[wave];
  auto item_tap = [&](int it) {
    const DcItem d = kDcItems[wave][it];
    const int pd = d.cls >> 2, ph = (d.cls >> 1) & 1, pw = d.cls & 1;
    const int sd = d.nb >> 2, sh = (d.nb >> 1) & 1, sw = d.nb & 1;
    return (deconv_tap(pd, sd) * 3 + deconv_tap(ph, sh)) * 3 + deconv_tap(pw, sw);
  };
  f32x4 bring[3][KC][2];
  auto load_b = [&](int it, f32x4 (*b)[2]) {
    const f32x4* wt = w4 + (int64_t)item_tap(it) * KC * 4 * COUT;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      b[kc][0] = wt[kc * 4 * COUT];
      b[kc][1] = wt[kc * 4 * COUT + 16];
    }
  };
  load_b(0, bring[0]);
  load_b(1, bring[1]);

  stage_patch3d<CIN, 2, 5, 5, RS, AFFINE>(xb, plane_c, plane_d, id, ih0, iw0, 1, g.Di, g.Hi, g.Wi, patch, aff, in_scale,
                                          in_shift, n / g.sps, in_bn, scratch);
  __syncthreads();

  const float* abase = patch + ((li >> 2) * 5 + (li & 3)) * RS + 4 * kq;
  const int64_t Vo = (int64_t)g.Do * g.Ho * g.Wo;
  const int ih = ih0 + kq;                                          // the lane's output rows come from input row ih
  double ssum[2] = {0.0, 0.0}, ssq[2] = {0.0, 0.0};
  f32x4 acc[2];
  acc[0] = acc[1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    if (it < nitems) {
      if (it + 2 < nitems) load_b(it + 2, bring[(it + 2) % 3]);
      const DcItem d = kDcItems[wave][it];
      const int sd = d.nb >> 2, sh = (d.nb >> 1) & 1, sw = d.nb & 1;
      const float* ap = abase + ((sd * 5 + sh) * 5 + sw) * RS;
      f32x4 a[KC];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) a[kc] = *reinterpret_cast<const f32x4*>(ap + 16 * kc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][j], bring[it % 3][kc][0][j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc][j], bring[it % 3][kc][1][j], acc[1], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (d.end) {
        // lane: channel li (+16), input cells (row kq, columns 0..3) -> outputs (2 id + pd, 2 ih + ph, 2 iw + pw)
        const int pd = d.cls >> 2, ph = (d.cls >> 1) & 1, pw = d.cls & 1;
        if (ih < g.Hi) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            float* dst =
                y + ((int64_t)n * COUT + 16 * t + li) * Vo + ((int64_t)(2 * id + pd) * g.Ho + 2 * ih + ph) * g.Wo + pw;
            float s = 0.0f, q = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (iw0 + e < g.Wi) {
                dst[2 * (iw0 + e)] = acc[t][e];
                s += acc[t][e];
                q += acc[t][e] * acc[t][e];
              }
            ssum[t] += (double)s;
            ssq[t] += (double)q;
          }
        }
        acc[0] = acc[1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  }
  if (partials != nullptr) {
    // per wave: channels 16 t + li over its classes; the four lane quarters hold different input rows
    __syncthreads();                                   // the patch is dead: its space serves as the reduction buffer
    double* red = reinterpret_cast<double*>(patch);    // [wave][32][2]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double s = ssum[t], q = ssq[t];
      s += __shfl_xor(s, 16);
      q += __shfl_xor(q, 16);
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lane < 16) {
        red[((wave * 32) + 16 * t + lane) * 2 + 0] = s;
        red[((wave * 32) + 16 * t + lane) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < COUT) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[(w * 32 + tid) * 2 + 0];
        q += red[(w * 32 + tid) * 2 + 1];
      }
      double* o = partials + ((((int64_t)n * gridDim.y + id) * gridDim.x + blockIdx.x) * COUT + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

template <int STRIDE, int CIN, int AFFINE>
int launch_bottom_mode(const float* x, const float* wp, float* y, BottomGeom g, int64_t N, const float* in_scale,
                       const float* in_shift, double* partials, const pf_bn_job& in_bn, hipStream_t s) {
  using C = BottomCfg<STRIDE, CIN>;
  g.tiles_w = (g.Wo + 3) / 4;
  dim3 grid((unsigned)(((g.Ho + 3) / 4) * g.tiles_w), (unsigned)g.Do, (unsigned)N);
  hipLaunchKernelGGL((conv3d_bottom_kernel<STRIDE, CIN, AFFINE>), grid, dim3(256), C::LDS, s, x, wp, y, g, in_scale,
                     in_shift, partials, in_bn);
  return pf_launch_status();
}

template <int STRIDE, int CIN>
int launch_bottom(const float* x, const float* wp, float* y, BottomGeom g, int64_t N, const float* in_scale,
                  const float* in_shift, double* partials, const pf_bn_job* in_bn, hipStream_t s) {
  pf_bn_job none = {};
  if (in_bn != nullptr) {
    const int rc = pf_bn_in_check(in_bn, CIN, (int)(N / g.sps));
    if (rc != PF_OK) return rc;
    return launch_bottom_mode<STRIDE, CIN, 2>(x, wp, y, g, N, nullptr, nullptr, partials, *in_bn, s);
  }
  if (in_scale != nullptr) return launch_bottom_mode<STRIDE, CIN, 1>(x, wp, y, g, N, in_scale, in_shift, partials, none, s);
  return launch_bottom_mode<STRIDE, CIN, 0>(x, wp, y, g, N, nullptr, nullptr, partials, none, s);
}

template <int AFFINE>
int launch_deconv_mode(const float* x, const float* wp, float* y, BottomGeom g, int64_t N, const float* in_scale,
                       const float* in_shift, double* partials, const pf_bn_job& in_bn, hipStream_t s) {
  g.tiles_w = (g.Wi + 3) / 4;
  dim3 grid((unsigned)(((g.Hi + 3) / 4) * g.tiles_w), (unsigned)g.Di, (unsigned)N);
  hipLaunchKernelGGL((deconv3d_bottom_kernel<AFFINE>), grid, dim3(256), DeconvCfg::LDS, s, x, wp, y, g, in_scale,
                     in_shift, partials, in_bn);
  return pf_launch_status();
}

}  // namespace

extern "C" {

int pf_conv3d_bottom_supported(int64_t Cin, int64_t Cout, int stride) {
  return Cout == 64 && ((stride == 1 && Cin == 64) || (stride == 2 && Cin == 32));
}

int pf_conv3d_bottom_blocks(int64_t Di, int64_t Hi, int64_t Wi, int stride) {
  if (Di <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int64_t Do = (Di - 1) / stride + 1, Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  return (int)(Do * ((Ho + 3) / 4) * ((Wo + 3) / 4));
}

int pf_conv3d_bottom_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                         int64_t Hi, int64_t Wi, int stride, const float* in_scale, const float* in_shift,
                         const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (in_bn == nullptr || in_scale == nullptr));
  PF_REQUIRE(N % samples_per_stat == 0 || in_bn == nullptr);
  if (!pf_conv3d_bottom_supported(Cin, Cout, stride)) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Di * Hi * Wi <= INT32_MAX / 4 && Di <= 65535);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  BottomGeom g;
  g.Di = (int)Di;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Do = (int)((Di - 1) / stride + 1);
  g.Ho = (int)((Hi - 1) / stride + 1);
  g.Wo = (int)((Wi - 1) / stride + 1);
  g.tiles_w = 0;
  g.sps = samples_per_stat;
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1) return launch_bottom<1, 64>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
  return launch_bottom<2, 32>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
}

int pf_deconv3d_bottom_supported(int64_t Cin, int64_t Cout) { return Cin == 64 && Cout == 32; }

int pf_deconv3d_bottom_blocks(int64_t Di, int64_t Hi, int64_t Wi) {
  if (Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
  return (int)(Di * ((Hi + 3) / 4) * ((Wi + 3) / 4));
}

int pf_deconv3d_bottom_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                           int64_t Hi, int64_t Wi, const float* in_scale, const float* in_shift, const pf_bn_job* in_bn,
                           int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (in_bn == nullptr || in_scale == nullptr));
  PF_REQUIRE(N % samples_per_stat == 0 || in_bn == nullptr);
  if (!pf_deconv3d_bottom_supported(Cin, Cout)) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Di * Hi * Wi <= INT32_MAX / 32 && Di <= 65535);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  BottomGeom g;
  g.Di = (int)Di;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Do = (int)(2 * Di);
  g.Ho = (int)(2 * Hi);
  g.Wo = (int)(2 * Wi);
  g.tiles_w = 0;
  g.sps = samples_per_stat;
  hipStream_t s = (hipStream_t)stream;
  pf_bn_job none = {};
  if (in_bn != nullptr) {
    const int rc = pf_bn_in_check(in_bn, 64, (int)(N / samples_per_stat));
    if (rc != PF_OK) return rc;
    return launch_deconv_mode<2>(x, wp, y, g, N, nullptr, nullptr, partials, *in_bn, s);
  }
  if (in_scale != nullptr) return launch_deconv_mode<1>(x, wp, y, g, N, in_scale, in_shift, partials, none, s);
  return launch_deconv_mode<0>(x, wp, y, g, N, nullptr, nullptr, partials, none, s);
}

}  // extern "C"
