// Row R (VolumeConv, SURVEY.md section 8(a) "R", 8(f) item 2): 3x3x3 / pad 1 / stride 1|2 conv3d as an
// implicit GEMM on the f32 matrix cores, with the BatchNorm batch statistics of the output produced in
// the epilogue.
//
// Why: the library path picked for the cost-volume regulariser's first layer (64 -> 8 channels over
// 48x64x80 voxels, 6.8 of VolumeConv's 9.87 GFLOP) runs at 7 TF/s on MI355X (919 us + 4 layout
// transposes, profiles/r01a); the second (64 -> 16, stride 2) goes through an explicit im2col
// (171 us) + GEMM (103 us).  M = output voxels, N = output channels, K = 27 taps x C_in; nothing is
// materialised.
//
// Mapping: one wave owns 16 consecutive output voxels along W of one (d, h) row and all output channels
// (NT column tiles of 16).  v_mfma_f32_16x16x4_f32: the 4-deep k-group is 4 INPUT CHANNELS at one tap, so
// lane l (row i = l&15, k = l>>4) reads x[ci0 + k][d'][h'][w' + i]: four 64-byte row segments per
// instruction straight from the NCDHW tensor (taps overlap, so L1/L2 serve the 27x reuse), and B is the
// host-packed weight wp[tap][ci][co] (64 contiguous bytes per k).  d/h padding is a wave-uniform skip, w
// padding a per-lane zero.  Exact float32 (an fmaf chain over (tap, ci)).  The 4 waves of a block take 4
// consecutive h rows so they share two of their three input rows in L1.
// Bound: fp32 MFMA (2*27*C_in*C_out flop per voxel vs 4*(C_in + C_out) bytes).
#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int STRIDE>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        float* __restrict__ y, int Cin, int Cout, int Di, int Hi,
                                                        int Wi, int Do, int Ho, int Wo,
                                                        double* __restrict__ partials) {
  constexpr int NCP = NT * 16;
  __shared__ float tile[4][NCP][17];          // per-wave transpose buffer for coalesced stores
  __shared__ double red[4][NCP][2];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int n = blockIdx.y;
  const int tiles_w = (Wo + 15) >> 4;
  const int hgroups = (Ho + 3) >> 2;
  const int64_t total = (int64_t)Do * tiles_w * hgroups;      // block-level work items: (do, wtile, 4 h rows)
  const int64_t plane_i = (int64_t)Hi * Wi, vol_i = plane_i * Di;
  const int64_t plane_o = (int64_t)Ho * Wo, vol_o = plane_o * Do;
  const float* xb = x + (int64_t)n * Cin * vol_i;
  float* yb = y + (int64_t)n * Cout * vol_o;
  const int cgroups = Cin >> 2;

  double ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = 0.0;

  for (int64_t item = blockIdx.x; item < total; item += gridDim.x) {
    const int hg = (int)(item % hgroups);
    const int64_t rest = item / hgroups;
    const int wt = (int)(rest % tiles_w);
    const int od = (int)(rest / tiles_w);
    const int oh = hg * 4 + wave;
    const int ow0 = wt * 16;
    const bool row_ok = oh < Ho;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    if (row_ok) {
      const int ow = ow0 + li;
      for (int kd = 0; kd < 3; ++kd) {
        const int id = od * STRIDE + kd - 1;
        if (id < 0 || id >= Di) continue;
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = oh * STRIDE + kh - 1;
          if (ih < 0 || ih >= Hi) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * STRIDE + kw - 1;
            const bool ok = (ow < Wo) && (iw >= 0) && (iw < Wi);
            const float* ap = xb + (int64_t)lk * vol_i + (int64_t)id * plane_i + (int64_t)ih * Wi + (ok ? iw : 0);
            const float* bp = wp + ((int64_t)((kd * 3 + kh) * 3 + kw) * Cin + lk) * NCP + li;
#pragma unroll 8
            for (int cg = 0; cg < cgroups; ++cg) {
              const float a = ok ? ap[(int64_t)cg * 4 * vol_i] : 0.0f;
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                const float b = bp[(int64_t)cg * 4 * NCP + 16 * t];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
              }
            }
          }
        }
      }
    }

    // epilogue: C/D layout col = lane&15 (channel), row = (lane>>4)*4 + r (voxel).  Transpose through LDS
    // so that each channel's 16 voxels leave as one 64-byte segment; accumulate BN statistics on the way.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s = 0.0f, q = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pos = lk * 4 + r;
        const float v = acc[t][r];
        tile[wave][16 * t + li][pos] = v;
        if (row_ok && ow0 + pos < Wo) {
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
    // tile[wave] is private to the wave and LDS operations of one wave execute in order: no workgroup
    // barrier, just keep the compiler from moving the reads above the writes
    __builtin_amdgcn_wave_barrier();
    if (row_ok) {
      for (int e = lane; e < NCP * 16; e += 64) {
        const int co = e >> 4, pos = e & 15;
        if (co < Cout && ow0 + pos < Wo)
          yb[(int64_t)co * vol_o + (int64_t)od * plane_o + (int64_t)oh * Wo + ow0 + pos] = tile[wave][co][pos];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  if (partials != nullptr) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (lane < 16) {
        red[wave][16 * t + lane][0] = ssum[t];
        red[wave][16 * t + lane][1] = ssq[t];
      }
    }
    __syncthreads();
    if (tid < Cout) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[w][tid][0];
        q += red[w][tid][1];
      }
      double* o = partials + (((int64_t)n * gridDim.x + blockIdx.x) * Cout + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

int blocks_for(int64_t Do, int64_t Ho, int64_t Wo) {
  const int64_t total = Do * ((Wo + 15) / 16) * ((Ho + 3) / 4);
  return (int)(total < 1024 ? total : 1024);
}

}  // namespace

extern "C" {

int pf_conv3d_blocks(int64_t Do, int64_t Ho, int64_t Wo) {
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return blocks_for(Do, Ho, Wo);
}

int pf_conv3d_k3_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                     int64_t Hi, int64_t Wi, int stride, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 4 && Cout >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535);
  PF_REQUIRE(stride == 1 || stride == 2);
  if ((Cin % 4) != 0 || Cout > 64) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Di * Hi * Wi <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  const int64_t Do = (Di + 2 - 3) / stride + 1, Ho = (Hi + 2 - 3) / stride + 1, Wo = (Wi + 2 - 3) / stride + 1;
  dim3 grid((unsigned)blocks_for(Do, Ho, Wo), (unsigned)N);
  hipStream_t s = (hipStream_t)stream;
  const int NT = (int)((Cout + 15) / 16);
#define PF_CONV_LAUNCH(NTV, SV)                                                                                  \
  hipLaunchKernelGGL((conv3d_k3_kernel<NTV, SV>), grid, dim3(256), 0, s, x, wp, y, (int)Cin, (int)Cout, (int)Di, \
                     (int)Hi, (int)Wi, (int)Do, (int)Ho, (int)Wo, partials)
  if (stride == 1) {
    if (NT == 1) PF_CONV_LAUNCH(1, 1);
    else if (NT == 2) PF_CONV_LAUNCH(2, 1);
    else if (NT == 3) PF_CONV_LAUNCH(3, 1);
    else PF_CONV_LAUNCH(4, 1);
  } else {
    if (NT == 1) PF_CONV_LAUNCH(1, 2);
    else if (NT == 2) PF_CONV_LAUNCH(2, 2);
    else if (NT == 3) PF_CONV_LAUNCH(3, 2);
    else PF_CONV_LAUNCH(4, 2);
  }
#undef PF_CONV_LAUNCH
  return pf_launch_status();
}

}  // extern "C"
