// ImageConv, all eleven layers of a tower (reference networks.py:84-124).  First the
// 32- and 64-channel layers (conv2d_wide_kernel); the 8- and 16-channel ones follow further down
// (conv2d_wide16_kernel, the same idea on the 16x16x4 MFMA).  KxK conv2d (3x3 stride 1 or 5x5 stride 2, pad K/2, no
// bias) as an implicit GEMM on v_mfma_f32_32x32x2_f32, with the previous layer's BatchNorm+ReLU applied while the input
// patch is staged and this layer's BatchNorm batch statistics in the epilogue:
//
//   * the staged input patch is channel-last in LDS ([pixel][C_in + 4]) and the reduction index is assigned to
//     (MFMA step, lane half) as  c = 8 kc + 4 h + j : lane (pixel m, half h) reads the 16 bytes
//     patch[pixel(m) + tap][8 kc + 4 h ..] with ONE ds_read_b128 and feeds element j to step (kc, j) -- a quarter
//     of an LDS instruction per MFMA and, the patch geometry being compile-time, no address arithmetic at all
//     (base register + immediate for every tap and chunk).  Round 1's 16x16x4 loop issued 1.25 ds_read_b32 and ~7 VALU
//     per MFMA of half the size (SQ counters, profiles/archive/r02/r02r_sq_counters.md);
//   * the weights are packed on the host in the matching order [kh][kw][kc][h][c_out][j], so a lane's B operand for
//     four steps is one 16-byte piece; a wave reads it from global memory / L2 as two contiguous 512-byte runs, six
//     pieces ahead (round 3; round 2 staged one kernel row of weights at a time through 80-100 KB of LDS);
//   * a 256-thread block owns (2 * NWM) x 16 output pixels x C_out channels: wave (wm, wn) the two rows
//     2 wm, 2 wm + 1 (32 pixels) x channels [32 wn, 32 wn + 32); C_out = 64: 2 x 2 waves (4 x 16 pixels), C_out = 32:
//     4 x 1 (8 x 16 pixels).  The accumulator layout (lane = channel, 4 consecutive registers = 4 consecutive
//     pixels of a row) stores straight to NCHW as 16-byte pieces -- no transpose through LDS; with cl_out the same
//     registers go out channel-last (a half-wave = 32 consecutive channels of one pixel) for the coarse warp;
//   * statistics: per lane over its 16 outputs (float), the two lane halves by one shuffle, waves through LDS,
//     one float64 partial row per block -- the layout pf_bn_finalize_jobs_f32 / pf_bn_resolve consume.
//
// Exact float32: every output is one fmaf chain over (kh, kw, kc, j, h); the order differs from the library's, so
// results agree with it to rounding (tests compare with float64).
// Bound: the f32 matrix peak (157 TF): 2 K^2 C_in C_out flop per output pixel against 4 (C_in / S^2 + C_out) bytes.
#include <stdlib.h>

#include "pf_common.h"
#include "pf_bn_resolve.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));   // (HIP's float4 struct in a register ARRAY ends up in scratch)

struct WideGeom {
  int Hi, Wi, Ho, Wo, tiles_w, tiles_h, sps;
  int cl_out;      // bit s: parameter set s writes y as (Ho, Wo, C_out) per sample instead of (C_out, Ho, Wo)
  // Parameter sets (pf_conv2d_wide_sets_f32: the two towers of the model in ONE launch): samples
  // [s * spset, (s + 1) * spset) convolve with the weights at wp + s * w_stride and normalise their input with the
  // s-th pending BatchNorm.  x_mode: where sample n = s * spset + i reads its input -- 0: sample n; 1: sample i
  // (every set reads the SAME spset samples); 2: sample i * sets + s (set-interleaved: x is (spset, sets, C_in, H, W),
  // what ONE convolution with the sets' stacked output channels wrote)
  int sets, spset, x_mode;
  int64_t w_stride;
};

// which parameter set sample n belongs to (block-uniform), and which input sample it reads
__device__ __forceinline__ int wide_set(const WideGeom& g, int n) { return (g.sets > 1 && n >= g.spset) ? 1 : 0; }
__device__ __forceinline__ int wide_input_sample(const WideGeom& g, int n, int set) {
  const int i = n - set * g.spset;
  return g.x_mode == 0 ? n : (g.x_mode == 1 ? i : i * g.sets + set);
}

template <int KS, int STRIDE, int CIN, int COUT>
struct WideCfg {
  static constexpr int NWN = COUT / 32;                 // waves along the channels
  static constexpr int NWM = 4 / NWN;                   // ... along the pixels (each: 2 rows x 16 columns)
  static constexpr int TH = 2 * NWM, TW = 16;           // output tile of a block
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
  static constexpr int NPIX = PH * PW;
  static constexpr int RS = CIN + 4;                    // floats per staged pixel (+4: 16-byte reads of 16
                                                        // consecutive pixels fall on 16 distinct bank quads)
  static constexpr int PATCH = NPIX * RS;
  static constexpr int KC = CIN / 8;
  static constexpr size_t LDS = sizeof(float) * (size_t)(PATCH + 2 * CIN) + sizeof(double) * 4 * 32 * 2;
  static_assert(COUT == 32 || COUT == 64, "C_out is 32 or 64");
  static_assert(CIN % 8 == 0 && PATCH % 4 == 0, "16-byte pieces");
  static_assert(PATCH * sizeof(float) >= 4096, "pf_bn_resolve's scratch lives in the (still empty) patch");
  static_assert(LDS <= 80 * 1024, "two blocks per CU");
};

// The input patch of a block: NCHW planes -> [pixel][channel] in LDS, the previous BatchNorm+ReLU on the way.
// Two phases so that a block can have the NEXT tile's loads in flight while it works on the current one: load()
// issues all of a thread's loads (NIT x 4 independent dwords), commit() applies the affine and writes 16-byte pieces.
template <int CIN, int NPIX, int PW, int RS>   // (CIN % 4 != 0: the last quad is zero-padded)
struct PatchStager {
  static constexpr int ITEMS = NPIX * ((CIN + 3) / 4);   // (pixel, channel quad) pairs of the patch
  static constexpr int NIT = (ITEMS + 255) / 256;
  static constexpr int PH = NPIX / PW;
  static_assert(CIN % 4 == 0 || CIN < 4, "a padded quad: one quad only");
  static_assert(PH < 256 && PW < 256, "patch coordinates are packed in bytes");
  float rx[NIT][4];
  // Round 4 (profiles/archive/r03/r03i: ~9 vector instructions per load and ~25 per committed item of index arithmetic, bounds and
  // padding selects cost the 8- and 16-channel layers as much time as their matrix instructions): everything about an
  // item that does not depend on the tile -- its offset inside a tile's input window, its patch coordinates, its
  // channel quad -- is computed ONCE per block (init); a tile then costs one add per load, and a tile whose window
  // lies inside the map (all but the border tiles) takes no bounds test and no padding select at all.
  int goff[NIT];          // 4 q * plane + pr * Wi + pc
  int meta[NIT];          // pr | pc << 8 | q << 16
  unsigned okmask;
  bool interior;

  __device__ __forceinline__ void init(int plane_i, int Wi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int it = tid + 256 * r;
      const int itc = it < ITEMS ? it : ITEMS - 1;
      const int q = itc / NPIX, p = itc - q * NPIX;     // lanes walk the patch's pixels: coalesced along a patch row
      const int pr = p / PW, pc = p - pr * PW;
      goff[r] = 4 * q * plane_i + pr * Wi + pc;
      meta[r] = pr | (pc << 8) | (q << 16);
    }
  }

  // (the same call with init() folded in, for the kernels that stage one tile per block)
  __device__ __forceinline__ void load(const float* __restrict__ xb, int plane_i, int ih0, int iw0, int Hi, int Wi) {
    interior = ih0 >= 0 && iw0 >= 0 && ih0 + PH <= Hi && iw0 + PW <= Wi;        // block-uniform
    const int origin = ih0 * Wi + iw0;
    if (interior) {
      okmask = ~0u;
#pragma unroll
      for (int r = 0; r < NIT; ++r) {
        const float* src = xb + (goff[r] + origin);
#pragma unroll
#ifdef PF_DBG_NOLOAD
        for (int j = 0; j < 4; ++j) rx[r][j] = (float)(threadIdx.x + j) + (float)(src - xb) * 1e-9f;
#else
        for (int j = 0; j < 4; ++j) rx[r][j] = (CIN % 4 == 0 || j < CIN) ? src[j * plane_i] : 0.0f;
#endif
      }
      return;
    }
    okmask = 0;
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int pr = meta[r] & 255, pc = (meta[r] >> 8) & 255, q = meta[r] >> 16;
      const int ih = ih0 + pr, iw = iw0 + pc;
      const bool ok = ih >= 0 && ih < Hi && iw >= 0 && iw < Wi;
      okmask |= (ok ? 1u : 0u) << r;
      const float* src = xb + (ok ? goff[r] + origin : 4 * q * plane_i);
#pragma unroll
#ifdef PF_DBG_NOLOAD
      for (int j = 0; j < 4; ++j) rx[r][j] = (float)(threadIdx.x + j) + (float)(src - xb) * 1e-9f;
#else
      for (int j = 0; j < 4; ++j) rx[r][j] = (CIN % 4 == 0 || j < CIN) ? src[j * plane_i] : 0.0f;
#endif
    }
  }

  template <bool AFF>
  __device__ __forceinline__ void commit(float* patch, const float* aff) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
      const int it = tid + 256 * r;
      const int pr = meta[r] & 255, pc = (meta[r] >> 8) & 255, q = meta[r] >> 16;
      f32x4 v = {rx[r][0], rx[r][1], rx[r][2], rx[r][3]};
      if (AFF) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(aff + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(aff + CIN + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], a[j], b[j]), 0.0f);
      }
      if (!interior && !((okmask >> r) & 1u)) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};   // zero padding applies AFTER it
      if (256 * (r + 1) <= ITEMS || it < ITEMS) *reinterpret_cast<f32x4*>(patch + (pr * PW + pc) * RS + 4 * q) = v;
    }
  }
};

// The pending BatchNorm of the input -> aff[0..CIN) scale, aff[CIN..2 CIN) shift (LDS).  AFFINE: 0 = none;
// 1 = rows (N/sps, CIN) from memory; 2 = computed here from the PRODUCER's statistics (pf_bn_resolve; `scratch`:
// 4 KB of LDS nobody uses yet).  Ends with a barrier when AFFINE != 0.
template <int CIN, int AFFINE>
__device__ __forceinline__ void wide_affine_prologue(float* aff, const float* __restrict__ in_scale,
                                                     const float* __restrict__ in_shift, int stat,
                                                     const pf_bn_job& in_bn, double* scratch) {
  static_assert(CIN % 4 == 0 || AFFINE == 0, "a padded channel quad takes no affine");
  const int tid = threadIdx.x;
  if (AFFINE == 1) {
    const float* sc = in_scale + (int64_t)stat * CIN;
    const float* sh = in_shift + (int64_t)stat * CIN;
    if (tid < CIN) aff[tid] = sc[tid];
    else if (tid < 2 * CIN) aff[tid] = sh[tid - CIN];
    __syncthreads();
  }
  if (AFFINE == 2) pf_bn_resolve<256>(in_bn, stat, aff, aff + CIN, scratch);
}

template <int CIN, int NPIX, int PW, int RS, int AFFINE>
__device__ __forceinline__ void wide_stage_patch(const float* __restrict__ xb, int plane_i, int ih0, int iw0, int Hi,
                                                 int Wi, float* patch, float* aff, const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift, int stat, const pf_bn_job& in_bn,
                                                 double* scratch) {
  PatchStager<CIN, NPIX, PW, RS> st;
  st.init(plane_i, Wi);
  st.load(xb, plane_i, ih0, iw0, Hi, Wi);
  wide_affine_prologue<CIN, AFFINE>(aff, in_scale, in_shift, stat, in_bn, scratch);
  st.template commit<(AFFINE != 0)>(patch, aff);
}

// Epilogue shared by the two kernels: C/D layout column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 h;
// stores straight to NCHW as 16-byte pieces (or channel-last), statistics per lane -> halves -> waves -> one row.
template <class C, int COUT>
__device__ __forceinline__ void wide_epilogue(const f32x16& acc, float* __restrict__ y, const WideGeom& g,
                                              double* __restrict__ partials, double* red, int n, int bx, int oh0,
                                              int ow0, int wm, int wn, int wave, int m, int h, int tid, bool cl_out) {
  // ---- epilogue: C/D layout column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 h -----------------
  const int co = wn * 32 + m;
  float* yb = y + ((int64_t)n * COUT + co) * ((int64_t)g.Ho * g.Wo);
  const bool vec_ok = (g.Wo & 3) == 0;
  float s = 0.0f, q = 0.0f;
  if (cl_out) {
    // channel-last output (the coarse tower's last layer feeds the warp, which samples channel-last maps): for one
    // accumulator element the 32 lanes of a half-wave hold 32 consecutive channels of one pixel = one 128-byte row
    float* ycl = y + (int64_t)n * g.Ho * g.Wo * COUT + co;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int oh = oh0 + 2 * wm + (rq >> 1);
      const int ow = ow0 + (rq & 1) * 8 + 4 * h;
      if (oh < g.Ho) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ow + e < g.Wo) {
            ycl[((int64_t)oh * g.Wo + ow + e) * COUT] = acc[4 * rq + e];
            s += acc[4 * rq + e];
            q += acc[4 * rq + e] * acc[4 * rq + e];
          }
        }
      }
    }
  } else
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const int oh = oh0 + 2 * wm + (rq >> 1);
    const int ow = ow0 + (rq & 1) * 8 + 4 * h;
    if (oh < g.Ho) {
      float* dst = yb + (int64_t)oh * g.Wo + ow;
      if (vec_ok && ow + 3 < g.Wo) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s += acc[4 * rq + e];
          q += acc[4 * rq + e] * acc[4 * rq + e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ow + e < g.Wo) {
            dst[e] = acc[4 * rq + e];
            s += acc[4 * rq + e];
            q += acc[4 * rq + e] * acc[4 * rq + e];
          }
        }
      }
    }
  }
  if (partials != nullptr) {
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (h == 0) {
      red[(wave * 32 + m) * 2 + 0] = (double)s;
      red[(wave * 32 + m) * 2 + 1] = (double)q;
    }
    __syncthreads();
    if (tid < COUT) {
      const int cn = tid >> 5, cm = tid & 31;
      double ds = 0.0, dq = 0.0;
#pragma unroll
      for (int w = 0; w < C::NWM; ++w) {
        ds += red[((w * C::NWN + cn) * 32 + cm) * 2 + 0];
        dq += red[((w * C::NWN + cn) * 32 + cm) * 2 + 1];
      }
      double* o = partials + (((int64_t)n * gridDim.x + bx) * COUT + tid) * 2;
      o[0] = ds;
      o[1] = dq;
    }
  }
}

// AFFINE: 0 = x is taken as is; 1 = relu(x * in_scale + in_shift), rows (N/sps, CIN); 2 = the same with the rows
// computed here from the PRODUCER's statistics (pf_bn_resolve: the pending BatchNorm never gets its own launch).
// The B operands (weights) come straight from global memory / L2 into a ring of registers: no weight rows in LDS
// (27-55 KB of patch per block instead of the 75-130 KB of round 2's row-staged form, so several blocks -- waves per
// SIMD -- share a CU and cover each other's prologue, LDS and memory latencies; two towers' worth of the 64-channel
// layer in one launch: 25.9 us against 31.8, profiles/archive/r03/r03c_microbench_conv2d_wide.log) and no barrier inside the tile.
// A lane's B operand for four MFMA steps is one 16-byte load; a wave's load is two contiguous 512-byte runs of the
// host-packed weights.

template <int KS, int STRIDE, int CIN, int COUT, int AFFINE>
__global__ __launch_bounds__(256) void conv2d_wide_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            float* __restrict__ y, WideGeom g,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            double* __restrict__ partials, pf_bn_job in_bn,
                                                            pf_bn_job in_bn1) {
  using C = WideCfg<KS, STRIDE, CIN, COUT>;
  constexpr int PW = C::PW, RS = C::RS, NPIX = C::NPIX;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;
  float* aff = lds + C::PATCH;
  double* red = reinterpret_cast<double*>(aff + 2 * CIN);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m = lane & 31, h = lane >> 5;
  const int wm = wave / C::NWN, wn = wave % C::NWN;
  unsigned xb_, xn_;
  pf_xcd_xy<PF_XCD_TOWER>(xb_, xn_);                  // XCD x owns a band of tile rows, not every eighth tile (pf_common.h)
  const int n = (int)xn_, bx = (int)xb_;
  const int tw = bx % g.tiles_w, th = bx / g.tiles_w;
  const int oh0 = th * C::TH, ow0 = tw * C::TW;
  const int ih0 = oh0 * STRIDE - C::PAD, iw0 = ow0 * STRIDE - C::PAD;
  const int plane_i = g.Hi * g.Wi;
  const int set = wide_set(g, n);
  const float* xb = x + (int64_t)wide_input_sample(g, n, set) * CIN * plane_i;

  constexpr int NP = KS * KS * C::KC;                  // 16-byte operand pairs of the tile
  constexpr int D = 6;                                 // B pieces in flight per lane
  const f32x4* bg = reinterpret_cast<const f32x4*>(wp + set * g.w_stride) + (h * COUT + wn * 32 + m);
  f32x4 bq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) bq[d] = bg[(d < NP ? d : 0) * 2 * COUT];

  // (AFFINE 1: affine rows are indexed by the global statistic group; AFFINE 2: by the group inside the set's job)
  wide_stage_patch<CIN, NPIX, PW, RS, AFFINE>(xb, plane_i, ih0, iw0, g.Hi, g.Wi, patch, aff, in_scale, in_shift,
                                              AFFINE == 2 ? (n - set * g.spset) / g.sps : n / g.sps,
                                              set ? in_bn1 : in_bn, reinterpret_cast<double*>(patch));
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const float* abase = patch + ((2 * wm + (m >> 4)) * STRIDE * PW + (m & 15) * STRIDE) * RS + 4 * h;
  f32x4 a = *reinterpret_cast<const f32x4*>(abase);
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    f32x4 an = a;
    if (t + 1 < NP) {
      const int tap = (t + 1) / C::KC, kc = (t + 1) % C::KC;
      const int kh = tap / KS, kw = tap % KS;
      an = *reinterpret_cast<const f32x4*>(abase + (kh * PW + kw) * RS + 8 * kc);
    }
    const f32x4 b = bq[t % D];
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + D < NP) bq[t % D] = bg[(t + D) * 2 * COUT];
    a = an;
  }
  wide_epilogue<C, COUT>(acc, y, g, partials, red, n, bx, oh0, ow0, wm, wn, wave, m, h, tid, (g.cl_out >> set) & 1);
}

template <int KS, int STRIDE, int CIN, int COUT, int AFFINE>
int launch_wide_mode(const float* x, const float* wp, float* y, WideGeom g, int64_t N, const float* in_scale,
                     const float* in_shift, double* partials, const pf_bn_job& in_bn, const pf_bn_job& in_bn1,
                     hipStream_t s) {
  using C = WideCfg<KS, STRIDE, CIN, COUT>;
  if (C::LDS > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};   // per instantiation, one bit per device
    const int rc = pf_allow_big_lds(reinterpret_cast<const void*>(&conv2d_wide_kernel<KS, STRIDE, CIN, COUT, AFFINE>),
                                    (int)C::LDS, done);
    if (rc != PF_OK) return rc;
  }
  g.tiles_w = (g.Wo + C::TW - 1) / C::TW;
  const int tiles_h = (g.Ho + C::TH - 1) / C::TH;
  dim3 grid((unsigned)(tiles_h * g.tiles_w), (unsigned)N);
  hipLaunchKernelGGL((conv2d_wide_kernel<KS, STRIDE, CIN, COUT, AFFINE>), grid, dim3(256), C::LDS, s, x, wp, y, g,
                     in_scale, in_shift, partials, in_bn, in_bn1);
  return pf_launch_status();
}

template <int KS, int STRIDE, int CIN, int COUT>
int launch_wide(const float* x, const float* wp, float* y, WideGeom g, int64_t N, const float* in_scale,
                const float* in_shift, double* partials, const pf_bn_job* in_bn, hipStream_t s) {
  if (in_bn != nullptr) {
    for (int k = 0; k < g.sets; ++k) {
      const int rc = pf_bn_in_check(in_bn + k, CIN, g.spset / g.sps);
      if (rc != PF_OK) return rc;
    }
    return launch_wide_mode<KS, STRIDE, CIN, COUT, 2>(x, wp, y, g, N, nullptr, nullptr, partials, in_bn[0],
                                                      in_bn[g.sets - 1], s);
  }
  pf_bn_job none = {};
  if (in_scale != nullptr)
    return launch_wide_mode<KS, STRIDE, CIN, COUT, 1>(x, wp, y, g, N, in_scale, in_shift, partials, none, none, s);
  return launch_wide_mode<KS, STRIDE, CIN, COUT, 0>(x, wp, y, g, N, nullptr, nullptr, partials, none, none, s);
}

// ------------------------------------------------------------------------------------------------
// C_out = 16 (conv1.x of the towers: 8 -> 16 5x5/2 and 16 -> 16 3x3 on 256 x 320 maps): the same idea on
// v_mfma_f32_16x16x4_f32 -- lane (i = lane & 15, kq = lane >> 4) is pixel i of a 16-pixel output row for the A
// operand and output channel i for B; its reduction slice is  c = (C_in / 4) kq + j : one 16-byte (C_in = 16) or
// 8-byte (C_in = 8) LDS read feeds C_in / 4 MFMA steps of a tap.  A block owns 16 x 16 output pixels, wave w the
// rows 4 w .. 4 w + 3 (four M tiles that share every B read); ALL weights (9-13 KB) stay in LDS, no barrier inside
// the tile.  960 blocks of 26-72 KB LDS on the cfg2 maps: 2-4 blocks per CU cover each other's latencies.
// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int CIN, int COUT>
struct Wide16Cfg {
  static constexpr int TH = 16, TW = 16, NCOL = 16;    // NCOL: columns of the MFMA tile (COUT = 8: half of them zero)
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
  static constexpr int NPIX = PH * PW;
  static constexpr int CINP = (CIN + 3) / 4 * 4;        // 3 -> 4
  static constexpr int RS = CINP + 4;
  static constexpr int PATCH = NPIX * RS;
  static constexpr int CPL = CINP / 4;                 // channels per lane and tap (4: b128 read, 2: b64, 1: b32)
  // C_out = 8 (round 4; the trick of conv3d_pair.hip): the 16 MFMA columns are 8 channels x TWO adjacent output rows
  // -- column co + 8 s is output row 2 rp + s -- which read the same KS + 1 patch rows: KS + 1 row taps instead of KS
  // for two rows at once, 1.5x fewer matrix instructions than half-empty tiles (3x3: 24 instead of 36 per row pair)
  // (the 3 -> 8 image layer keeps the plain tile: its statistics then sum exactly like the stacked 3 -> 8 + 8 launch of
  // the towers' shared first layer, which the bit-equality test of the shared launches relies on)
  static constexpr bool PAIR = COUT == 8 && CIN >= 8;
  static constexpr int KH = PAIR ? KS + 1 : KS;        // row taps of the MFMA loop
  static constexpr int WALL = KH * KS * CINP * NCOL;   // packed weights [row tap][kw][kq][column][CPL]
  static constexpr int WSPACE = WALL > 1024 ? WALL : 1024;   // (>= 4 KB: pf_bn_resolve's scratch before W lands)
  static constexpr size_t LDS = sizeof(float) * (size_t)(PATCH + WSPACE + 2 * CINP) + sizeof(double) * 4 * 16 * 2;
  static_assert(CIN == 3 || CIN == 8 || CIN == 16, "C_in is 3, 8 or 16");
  static_assert(COUT == 8 || COUT == 16, "C_out is 8 or 16");
  static_assert(!PAIR || STRIDE == 1, "paired rows: stride 1");
  static_assert(PATCH % 4 == 0 && WALL % 4 == 0, "16-byte pieces");
  static_assert(LDS <= 80 * 1024, "two blocks per CU");
};

template <int KS, int STRIDE, int CIN, int COUT, int AFFINE>
__global__ __launch_bounds__(256) void conv2d_wide16_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                            float* __restrict__ y, WideGeom g,
                                                            const float* __restrict__ in_scale,
                                                            const float* __restrict__ in_shift,
                                                            double* __restrict__ partials, pf_bn_job in_bn,
                                                            pf_bn_job in_bn1) {
  using C = Wide16Cfg<KS, STRIDE, CIN, COUT>;
  constexpr int PW = C::PW, RS = C::RS, NPIX = C::NPIX, CPL = C::CPL, NCOL = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* patch = lds;
  float* wl = lds + C::PATCH;
  float* aff = wl + C::WSPACE;
  double* red = reinterpret_cast<double*>(aff + 2 * C::CINP);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, kq = lane >> 4;
  unsigned xb_, xn_;
  pf_xcd_xy<PF_XCD_TOWER>(xb_, xn_);
  const int n = (int)xn_, bx = (int)xb_;
  const int plane_i = g.Hi * g.Wi;
  const int set = wide_set(g, n);
  const float* xb = x + (int64_t)wide_input_sample(g, n, set) * CIN * plane_i;
  wp += set * g.w_stride;
  const int tiles = g.tiles_h * g.tiles_w;
  const int nb = gridDim.x;

  // all weights: global -> registers now, -> LDS after the affine prologue (the resolve borrows the space until then)
  constexpr int W4 = C::WALL / 4, NWR = (W4 + 255) / 256;
  f32x4 rw[NWR];
#pragma unroll
  for (int r = 0; r < NWR; ++r) {
    const int e = tid + 256 * r;
    rw[r] = reinterpret_cast<const f32x4*>(wp)[e < W4 ? e : W4 - 1];
  }
  // A block walks its tiles (blockIdx.x, + gridDim.x, ...) with the NEXT tile's patch loads in flight while the
  // matrix cores work on the current one: with one tile per block every block of the launch is resident at once and
  // they all load, then all multiply, then all store -- each phase leaving the other units idle.
  PatchStager<CIN, NPIX, PW, RS> st;
  st.init(plane_i, g.Wi);
  int tile = bx;
  auto tile_origin = [&](int t, int& oh0, int& ow0) {
    const int th = t / g.tiles_w;
    oh0 = th * C::TH;
    ow0 = (t - th * g.tiles_w) * C::TW;
  };
  {
    int oh0, ow0;
    tile_origin(tile, oh0, ow0);
    st.load(xb, plane_i, oh0 * STRIDE - C::PAD, ow0 * STRIDE - C::PAD, g.Hi, g.Wi);
  }
  wide_affine_prologue<CIN, AFFINE>(aff, in_scale, in_shift, AFFINE == 2 ? (n - set * g.spset) / g.sps : n / g.sps,
                                    set ? in_bn1 : in_bn, reinterpret_cast<double*>(wl));
#pragma unroll
  for (int r = 0; r < NWR; ++r) {
    const int e = tid + 256 * r;
    if (256 * (r + 1) <= W4 || e < W4) reinterpret_cast<f32x4*>(wl)[e] = rw[r];
  }

  constexpr bool PAIR = C::PAIR;
  constexpr int NR = PAIR ? 2 : 4;                     // M tiles of a wave: row pairs, or rows
  const float* abase = patch + ((4 * wave) * STRIDE * PW + li * STRIDE) * RS + CPL * kq;
  const float* bbase = wl + (kq * NCOL + li) * CPL;
  constexpr int TAPS = C::KH * KS;
  struct Op {
    float v[CPL];
  };
  auto read_op = [&](const float* p) {
    Op o;
    if (CPL == 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p);
      o.v[0] = t[0];
      o.v[1 % CPL] = t[1];
      o.v[2 % CPL] = t[2];
      o.v[3 % CPL] = t[3];
    } else if (CPL == 2) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(p);
      o.v[0] = t[0];
      o.v[1 % CPL] = t[1];
    } else {
      o.v[0] = p[0];
    }
    return o;
  };
  auto read_a = [&](int t, Op* a) {
    const int kh = t / KS, kw = t - kh * KS;
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = read_op(abase + (((PAIR ? 2 * r : r) * STRIDE + kh) * PW + kw) * RS);
  };
  const bool vec_ok = (g.Wo & 3) == 0;
  // the output channel and row of this lane's accumulator column: li, row r -- or, paired, li & 7 and row 2 r + (li >> 3)
  const int my_co = PAIR ? (li & 7) : li;
  const int my_row = PAIR ? (li >> 3) : 0;
  const bool col_ok = my_co < COUT;
  const int plane_o = g.Ho * g.Wo;                     // (C_out * plane_o < 2^31: C_out <= C_in * stride^2, checked on the host)
  float* const yn = y + (int64_t)n * COUT * plane_o;
  double ds = 0.0, dq = 0.0;
#pragma unroll 1
  for (; tile < tiles; tile += nb) {
    int oh0, ow0;
    tile_origin(tile, oh0, ow0);
    st.template commit<(AFFINE != 0)>(patch, aff);
    __syncthreads();
    if (tile + nb < tiles) {
      int noh0, now0;
      tile_origin(tile + nb, noh0, now0);
      st.load(xb, plane_i, noh0 * STRIDE - C::PAD, now0 * STRIDE - C::PAD, g.Hi, g.Wi);
    }
    f32x4 acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    Op a[NR], b;
    read_a(0, a);
    b = read_op(bbase);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      Op an[NR], bn = b;
#pragma unroll
      for (int r = 0; r < NR; ++r) an[r] = a[r];
      if (t + 1 < TAPS) {
        read_a(t + 1, an);
        bn = read_op(bbase + (t + 1) * 4 * NCOL * CPL);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < CPL; ++j)
#pragma unroll
#ifdef PF_DBG_NOMFMA
        for (int r = 0; r < NR; ++r) acc[r][(j + t) & 3] += a[r].v[j] * b.v[j];
#else
        for (int r = 0; r < NR; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].v[j], b.v[j], acc[r], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < NR; ++r) a[r] = an[r];
      b = bn;
    }
    // epilogue: C/D layout column = lane & 15, rows (pixels of the output row) 4 kq + {0..3}
    float s = 0.0f, q = 0.0f;
    const int ow = ow0 + 4 * kq;
    if (vec_ok && oh0 + C::TH <= g.Ho && ow0 + C::TW <= g.Wo) {
      // a whole tile (block-uniform; every tile of the model's shapes): one base offset, no bounds, 16-byte stores
      float* dst = yn + (my_co * plane_o + (oh0 + 4 * wave + my_row) * g.Wo + ow);
      if (col_ok) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#ifdef PF_DBG_NOSTORE
          if (acc[r][0] == 123.456f)
#endif
          *reinterpret_cast<f32x4*>(dst + (PAIR ? 2 * r : r) * g.Wo) = acc[r];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s += acc[r][e];
            q += acc[r][e] * acc[r][e];
          }
        }
      }
    } else {
      float* yb = yn + (int64_t)my_co * plane_o;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int oh = oh0 + 4 * wave + (PAIR ? 2 * r : r) + my_row;
        if (oh < g.Ho && col_ok) {
          float* dst = yb + (int64_t)oh * g.Wo + ow;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (ow + e < g.Wo) {
              dst[e] = acc[r][e];
              s += acc[r][e];
              q += acc[r][e] * acc[r][e];
            }
          }
        }
      }
    }
    ds += (double)s;
    dq += (double)q;
    __syncthreads();                                   // every wave is done with the patch: the next commit may land
  }
  if (PAIR && partials != nullptr) {                   // the two rows of a pair hold the same channel
    ds += __shfl_xor(ds, 8);
    dq += __shfl_xor(dq, 8);
  }
  if (partials != nullptr) {
    ds += __shfl_xor(ds, 16);
    dq += __shfl_xor(dq, 16);
    ds += __shfl_xor(ds, 32);
    dq += __shfl_xor(dq, 32);
    if (lane < 16) {
      red[(wave * 16 + lane) * 2 + 0] = ds;
      red[(wave * 16 + lane) * 2 + 1] = dq;
    }
    __syncthreads();
    if (tid < COUT) {
      double ts = 0.0, tq = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        ts += red[(w * 16 + tid) * 2 + 0];
        tq += red[(w * 16 + tid) * 2 + 1];
      }
      double* o = partials + (((int64_t)n * gridDim.x + bx) * COUT + tid) * 2;
      o[0] = ts;
      o[1] = tq;
    }
  }
}

// Blocks per sample of the 16-wide kernel: a block walks PF_W16_TPB tiles with a stride of the block count (neighbouring
// blocks stay on neighbouring tiles).  Two tiles per block were best for ONE forward at a time (1 / 3 / 4 slower,
// profiles/archive/r02/r02aj_small_ab.txt: too few blocks on the small maps); with four scene lanes in flight the chip is full
// anyway and the ~190-instruction block prologue (vector instructions cost matrix time, profiles/archive/r03/r03j) is worth
// amortising: 2 / 3 / 4 / 5 / 8 tiles per block = 1 065 / 1 078 / 1 083 depth maps/s on one box, 994 / - / 1 015 / 1 016 /
// 1 015 on a slower one (profiles/archive/r03/r03l_block_policies_ab.log).
#ifndef PF_W16_TPB
#define PF_W16_TPB 4
#endif
int wide16_blocks(int tiles) { return (tiles + PF_W16_TPB - 1) / PF_W16_TPB; }

template <int KS, int STRIDE, int CIN, int COUT, int AFFINE>
int launch_wide16_mode(const float* x, const float* wp, float* y, WideGeom g, int64_t N, const float* in_scale,
                       const float* in_shift, double* partials, const pf_bn_job& in_bn, const pf_bn_job& in_bn1,
                       hipStream_t s) {
  using C = Wide16Cfg<KS, STRIDE, CIN, COUT>;
  if (C::LDS > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};
    const int rc = pf_allow_big_lds(
        reinterpret_cast<const void*>(&conv2d_wide16_kernel<KS, STRIDE, CIN, COUT, AFFINE>), (int)C::LDS, done);
    if (rc != PF_OK) return rc;
  }
  g.tiles_w = (g.Wo + C::TW - 1) / C::TW;
  g.tiles_h = (g.Ho + C::TH - 1) / C::TH;
  dim3 grid((unsigned)wide16_blocks(g.tiles_h * g.tiles_w), (unsigned)N);
  hipLaunchKernelGGL((conv2d_wide16_kernel<KS, STRIDE, CIN, COUT, AFFINE>), grid, dim3(256), C::LDS, s, x, wp, y, g,
                     in_scale, in_shift, partials, in_bn, in_bn1);
  return pf_launch_status();
}

template <int KS, int STRIDE, int CIN, int COUT>
int launch_wide16(const float* x, const float* wp, float* y, WideGeom g, int64_t N, const float* in_scale,
                  const float* in_shift, double* partials, const pf_bn_job* in_bn, hipStream_t s) {
  pf_bn_job none = {};
  if constexpr (CIN % 4 == 0) {
    if (in_bn != nullptr) {
      for (int k = 0; k < g.sets; ++k) {
        const int rc = pf_bn_in_check(in_bn + k, CIN, g.spset / g.sps);
        if (rc != PF_OK) return rc;
      }
      return launch_wide16_mode<KS, STRIDE, CIN, COUT, 2>(x, wp, y, g, N, nullptr, nullptr, partials, in_bn[0],
                                                          in_bn[g.sets - 1], s);
    }
    if (in_scale != nullptr)
      return launch_wide16_mode<KS, STRIDE, CIN, COUT, 1>(x, wp, y, g, N, in_scale, in_shift, partials, none, none, s);
  } else {
    if (in_bn != nullptr || in_scale != nullptr) return PF_ERR_UNSUPPORTED;   // (the image layer has no pending BatchNorm)
  }
  return launch_wide16_mode<KS, STRIDE, CIN, COUT, 0>(x, wp, y, g, N, nullptr, nullptr, partials, none, none, s);
}

int wide_tile_rows(int64_t Cout) { return Cout == 64 ? 4 : (Cout == 32 ? 8 : 16); }   // 8 and 16 channels: 16 x 16 tiles

}  // namespace

extern "C" {

int pf_conv2d_wide_supported(int64_t Cin, int64_t Cout, int kernel_size, int stride) {
  if (kernel_size == 3 && stride == 1)
    return (Cin == 64 && Cout == 64) || (Cin == 32 && Cout == 32) || (Cin == 16 && Cout == 16) || (Cin == 8 && Cout == 8) ||
           (Cin == 3 && (Cout == 8 || Cout == 16));
  if (kernel_size == 5 && stride == 2) return (Cin == 32 && Cout == 64) || (Cin == 16 && Cout == 32) || (Cin == 8 && Cout == 16);
  return 0;
}

int pf_conv2d_wide_blocks(int64_t Cout, int64_t Hi, int64_t Wi, int stride) {
  if ((Cout != 8 && Cout != 16 && Cout != 32 && Cout != 64) || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int64_t Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  const int th = wide_tile_rows(Cout);
  const int tiles = (int)(((Ho + th - 1) / th) * ((Wo + 15) / 16));
  return Cout <= 16 ? wide16_blocks(tiles) : tiles;
}

int pf_conv2d_wide_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Hi,
                       int64_t Wi, int kernel_size, int stride, const float* in_scale, const float* in_shift,
                       const pf_bn_job* in_bn, int samples_per_stat, double* partials, int out_channel_last,
                       void* stream) {
  return pf_conv2d_wide_sets_f32(x, 0, wp, 0, 1, y, N, Cin, Cout, Hi, Wi, kernel_size, stride, in_scale, in_shift, in_bn,
                                 samples_per_stat, partials, out_channel_last ? 1 : 0, stream);
}

int pf_conv2d_wide_sets_f32(const float* x, int x_layout, const float* wp, int64_t wp_set_stride, int sets, float* y,
                            int64_t N, int64_t Cin, int64_t Cout, int64_t Hi, int64_t Wi, int kernel_size, int stride,
                            const float* in_scale, const float* in_shift, const pf_bn_job* in_bn, int samples_per_stat,
                            double* partials, int out_channel_last, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535 && samples_per_stat >= 1);
  PF_REQUIRE((sets == 1 || sets == 2) && N % sets == 0 && wp_set_stride >= 0 && (wp_set_stride & 3) == 0);
  PF_REQUIRE(out_channel_last >= 0 && out_channel_last < (1 << sets) && x_layout >= 0 && x_layout <= 2);
  if (out_channel_last && Cout < 32) return PF_ERR_UNSUPPORTED;     // (built for the 32x32x2 kernels only)
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (in_bn == nullptr || in_scale == nullptr));
  PF_REQUIRE((N / sets) % samples_per_stat == 0 || (in_bn == nullptr && sets == 1));
  if (!pf_conv2d_wide_supported(Cin, Cout, kernel_size, stride)) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cin * Hi * Wi <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  WideGeom g;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Ho = (int)((Hi - 1) / stride + 1);
  g.Wo = (int)((Wi - 1) / stride + 1);
  g.tiles_w = g.tiles_h = 0;
  g.sps = samples_per_stat;
  g.cl_out = out_channel_last;
  g.sets = sets;
  g.spset = (int)(N / sets);
  g.x_mode = sets > 1 ? x_layout : 0;
  g.w_stride = wp_set_stride;
  hipStream_t s = (hipStream_t)stream;
  if (Cout == 8) {
    if (Cin == 3) return launch_wide16<3, 1, 3, 8>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
    return launch_wide16<3, 1, 8, 8>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
  }
  if (Cout == 16) {
    if (Cin == 3) return launch_wide16<3, 1, 3, 16>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
    if (kernel_size == 3) return launch_wide16<3, 1, 16, 16>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
    return launch_wide16<5, 2, 8, 16>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
  }
  if (kernel_size == 3) {
    if (Cin == 64) return launch_wide<3, 1, 64, 64>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
    return launch_wide<3, 1, 32, 32>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
  }
  if (Cin == 32) return launch_wide<5, 2, 32, 64>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
  return launch_wide<5, 2, 16, 32>(x, wp, y, g, N, in_scale, in_shift, partials, in_bn, s);
}

}  // extern "C"
