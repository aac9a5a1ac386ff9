// Row R decoder (VolumeConv conv4_0 / conv5_0 / conv6_0, reference networks.py:141-143; nn/conv.py:189-216):
// ConvTranspose3d, 3x3x3 kernel, stride 2, padding 1, output_padding 1 (output = exactly twice the input in
// every dimension), with the skip-connection add of the decoder (networks.py:163-165, "up + half") applied
// while loading and the BatchNorm batch statistics of the output produced in the epilogue.
//
// Why: the library lowers this layer to a GEMM into a 27x-expanded column buffer plus a col2im scatter
// (conv6_0 on cfg2: 27 + 25 us, plus 4.6 us for the add and 4.5 us for the statistics pass,
// profiles/r01f); the layer itself is 0.2 GFLOP over 10 MB.
//
// Structure: output voxel o = 2 i - 1 + k per dimension, so an output of even coordinate 2i takes only tap
// k=1 of input i and an output of odd coordinate 2i+1 takes tap k=0 of input i+1 and tap k=2 of input i.
// One lane owns one input cell (id, ih, iw): it reads the 2x2x2 input neighbourhood once per input channel
// and produces the 2x2x2 output cell (1+2+2+4+2+4+4+8 = 27 taps) for CG output channels; the weights of
// (input channel, channel group) are wave-uniform and come through the scalar cache.  The neighbour loads
// of 8 input channels are issued together.  Lanes are consecutive along W: loads coalesce and each lane
// stores two adjacent floats per output row.
// Bound: latency / issue (tiny); algorithmic bytes 4*(Cin*vol*(1 or 2) + Cout*8*vol).
#include "pf_common.h"
#include "pf_bn_resolve.h"

namespace {

constexpr int kDcThreads = 128;
constexpr int kDcUnroll = 8;      // input channels whose loads are issued together

// The pending BatchNorm + ReLU of xa (the previous decoder layer's raw output), applied to every loaded value BEFORE
// the skip add: mode 0 none, 1 rows (N / sps, Cin), 2 resolved by the block from the producer's statistics rows.
struct DcAffine {
  int mode, sps;
  const float* scale;
  const float* shift;
  pf_bn_job bn;
};
constexpr int kDcAffMax = 64;       // input channels with a pending BatchNorm

template <int CG, bool ADD, int U>
__global__ __launch_bounds__(kDcThreads) void deconv3d_k3s2_kernel(const float* __restrict__ xa,
                                                                   const float* __restrict__ xb,
                                                                   const float* __restrict__ w,
                                                                   float* __restrict__ y, int Cin, int Cout, int D,
                                                                   int H, int W, double* __restrict__ partials,
                                                                   DcAffine A) {
  __shared__ double red[kDcThreads / 64][2 * CG];
  __shared__ float aff[2 * kDcAffMax];
  __shared__ double aff_red[2 * kDcThreads];
  if (A.mode == 1) {
    const int stat = blockIdx.z / A.sps;
    for (int e = threadIdx.x; e < 2 * Cin; e += kDcThreads)
      aff[e < Cin ? e : kDcAffMax + e - Cin] = e < Cin ? A.scale[(int64_t)stat * Cin + e] : A.shift[(int64_t)stat * Cin + e - Cin];
    __syncthreads();
  } else if (A.mode == 2) {
    pf_bn_resolve<kDcThreads>(A.bn, blockIdx.z / A.sps, aff, aff + kDcAffMax, aff_red);
  }
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * CG;
  const int n = blockIdx.z;
  const int plane = H * W, vol = plane * D;
  const int cell = blockIdx.x * kDcThreads + tid;
  const bool live = cell < vol;
  const int c0 = live ? cell : 0;
  const int id = c0 / plane;
  const int rem = c0 - id * plane;
  const int ih = rem / W, iw = rem - ih * W;
  const bool vd = id + 1 < D, vh = ih + 1 < H, vw = iw + 1 < W;

  // neighbour s = (sd, sh, sw): offset of input (id+sd, ih+sh, iw+sw), or the cell itself (weight 0) outside
  int off[8];
  float keep[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int sd = s >> 2, sh = (s >> 1) & 1, sw = s & 1;
    const bool ok = live && (sd == 0 || vd) && (sh == 0 || vh) && (sw == 0 || vw);
    off[s] = ok ? c0 + sd * plane + sh * W + sw : c0;
    keep[s] = ok ? 1.0f : 0.0f;
  }

  float acc[8][CG];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[p][c] = 0.0f;

  const float* xan = xa + (int64_t)n * Cin * vol;
  const float* xbn = ADD ? xb + (int64_t)n * Cin * vol : nullptr;
  // U input channels per round: all of a round's neighbour loads are in flight together (the layer is a
  // chain of Cin dependent global-load latencies otherwise: 35 us for 32 channels)
  for (int ci0 = 0; ci0 < Cin; ci0 += U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ci = min(ci0 + u, Cin - 1);                       // tail: reload the last channel, unused
      const float* pa = xan + (int64_t)ci * vol;
#pragma unroll
      for (int s = 0; s < 8; ++s) v[u][s] = pa[off[s]];
      if (A.mode) {                                               // block-uniform
        const float sa = aff[ci], sb = aff[kDcAffMax + ci];
#pragma unroll
        for (int s = 0; s < 8; ++s) v[u][s] = fmaxf(fmaf(v[u][s], sa, sb), 0.0f);
      }
      if (ADD) {
        const float* pb = xbn + (int64_t)ci * vol;
#pragma unroll
        for (int s = 0; s < 8; ++s) v[u][s] += pb[off[s]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ci0 + u < Cin) {                                        // wave-uniform
#pragma unroll
        for (int s = 0; s < 8; ++s) v[u][s] *= keep[s];
        const float* __restrict__ wg = w + ((int64_t)(ci0 + u) * Cout + co0) * 27;   // wave-uniform: scalar loads
        // per dimension: parity 0 -> (shift 0, tap 1); parity 1 -> (shift 1, tap 0), (shift 0, tap 2)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
#pragma unroll
          for (int a = 0; a <= pd; ++a) {
            const int sd = pd ? 1 - a : 0, kd = pd ? 2 * a : 1;
#pragma unroll
            for (int b = 0; b <= ph; ++b) {
              const int sh = ph ? 1 - b : 0, kh = ph ? 2 * b : 1;
#pragma unroll
              for (int e = 0; e <= pw; ++e) {
                const int sw = pw ? 1 - e : 0, kw = pw ? 2 * e : 1;
                const float xv = v[u][(sd << 2) | (sh << 1) | sw];
#pragma unroll
                for (int c = 0; c < CG; ++c)
                  acc[p][c] = fmaf(xv, wg[c * 27 + (kd * 3 + kh) * 3 + kw], acc[p][c]);
              }
            }
          }
        }
      }
    }
  }

  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t plane_o = (int64_t)Ho * Wo, vol_o = plane_o * 2 * D;
  float* yn = y + ((int64_t)n * Cout + co0) * vol_o;
  float ssum[CG], ssq[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) ssum[c] = ssq[c] = 0.0f;
  if (live) {
#pragma unroll
    for (int c = 0; c < CG; ++c) {
#pragma unroll
      for (int p = 0; p < 8; p += 2) {
        const int pd = p >> 2, ph = (p >> 1) & 1;
        const float2 o = make_float2(acc[p][c], acc[p + 1][c]);
        *reinterpret_cast<float2*>(yn + (int64_t)c * vol_o + (int64_t)(2 * id + pd) * plane_o +
                                   (int64_t)(2 * ih + ph) * Wo + 2 * iw) = o;
        ssum[c] += o.x + o.y;
        ssq[c] += o.x * o.x + o.y * o.y;
      }
    }
  }
  if (partials != nullptr) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      double s = (double)ssum[c], q = (double)ssq[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
      }
      if (lane == 0) {
        red[wave][2 * c] = s;
        red[wave][2 * c + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 2 * CG) {
      double v = 0.0;
#pragma unroll
      for (int wv = 0; wv < kDcThreads / 64; ++wv) v += red[wv][tid];
      partials[(((int64_t)n * gridDim.x + blockIdx.x) * Cout + co0 + (tid >> 1)) * 2 + (tid & 1)] = v;
    }
  }
}

template <int CG>
int launch_dc(const float* xa, const float* xb, const float* w, float* y, int64_t N, int Cin, int Cout, int D, int H,
              int W, double* partials, const DcAffine& A, hipStream_t s) {
  dim3 grid((unsigned)pf_cdiv((int64_t)D * H * W, kDcThreads), (unsigned)(Cout / CG), (unsigned)N);
  if (xb != nullptr)
    hipLaunchKernelGGL((deconv3d_k3s2_kernel<CG, true, kDcUnroll>), grid, dim3(kDcThreads), 0, s, xa, xb, w, y, Cin, Cout, D, H, W,
                       partials, A);
  else
    hipLaunchKernelGGL((deconv3d_k3s2_kernel<CG, false, kDcUnroll>), grid, dim3(kDcThreads), 0, s, xa, xb, w, y, Cin, Cout, D, H,
                       W, partials, A);
  return pf_launch_status();
}

}  // namespace

extern "C" {

int pf_deconv3d_blocks(int64_t D, int64_t H, int64_t W) {
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  return (int)pf_cdiv(D * H * W, kDcThreads);
}

int pf_deconv3d_k3s2_f32(const float* xa, const float* xb, const float* w, float* y, int64_t N, int64_t Cin,
                         int64_t Cout, int64_t D, int64_t H, int64_t W, const float* in_scale, const float* in_shift,
                         const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && D >= 1 && H >= 1 && W >= 1 && N <= 65535);
  PF_REQUIRE(Cin * D * H * W <= INT32_MAX && Cout <= 65535 * 4);
  PF_REQUIRE(samples_per_stat >= 1 && (in_scale == nullptr) == (in_shift == nullptr));
  PF_REQUIRE(in_bn == nullptr || in_scale == nullptr);
  if ((in_bn != nullptr || in_scale != nullptr) && Cin > kDcAffMax) return PF_ERR_UNSUPPORTED;
  if (N == 0) return PF_OK;
  PF_REQUIRE(xa && w && y);
  DcAffine A;
  A.mode = in_bn ? 2 : (in_scale ? 1 : 0);
  A.sps = samples_per_stat;
  A.scale = in_scale;
  A.shift = in_shift;
  A.bn = pf_bn_job{};
  if (in_bn != nullptr) {
    PF_REQUIRE(N % samples_per_stat == 0);
    const int rc = pf_bn_in_check(in_bn, (int)Cin, (int)(N / samples_per_stat));
    if (rc != PF_OK) return rc;
    A.bn = *in_bn;
  }
  hipStream_t s = (hipStream_t)stream;
  // channel group: 4 when that still leaves >= 2 blocks per CU, else 2 / 1 (more, smaller work items)
  const int64_t cell_blocks = pf_cdiv(D * H * W, kDcThreads);
  if ((Cout % 4) == 0 && cell_blocks * (Cout / 4) * N >= 512)
    return launch_dc<4>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
  if ((Cout % 2) == 0 && cell_blocks * (Cout / 2) * N >= 512)
    return launch_dc<2>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
  return launch_dc<1>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
}

}  // extern "C"
