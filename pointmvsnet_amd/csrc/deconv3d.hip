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
#include <stdlib.h>

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


// ------------------------------------------------------------------------------------------------
// The same layer on the f32 matrix cores (round 6; VERDICT r5 item 4: the lane-per-cell form above runs at 0.03-0.07 of
// the MFMA peak and re-reads its input once per channel group).  Implicit GEMM per output parity class: with
// o = 2 i - 1 + k, class p = (pd, ph, pw) of input cell i is  y[co][2 i + p] = sum_{taps of p} sum_ci w[ci][co][tap] *
// x[ci][i + shift(tap)]  -- 1, 2, 2, 4, 2, 4, 4, 8 taps, 27 in all, each reading one of the 8 shifted copies of the input.
//   * a wave owns 16 consecutive input cells and 16 output channels: 8 accumulator tiles D[co][cell] of
//     v_mfma_f32_16x16x4_f32 (rows = output channels so that a lane's four values are four channels of ONE cell and the 16
//     lanes of a row group write 16 adjacent cells: 128-byte runs with the pw pair folded into an 8-byte store);
//   * reduction step = 4 input channels: lane (cell n = lane & 15, k = lane >> 4) loads x[4 q + k][cell n + shift] for the
//     8 shifts ONCE per quad (pending BatchNorm + ReLU, skip add and the zero outside the volume applied on the way) and
//     feeds all 27 (class, tap) products of the quad from registers;
//   * the weights of the block's 16 output channels sit in LDS as [quad][tap][k][co]: a lane's A operand is one
//     conflict-free 4-byte read;
//   * block = 8 waves = 128 cells (the statistics rows keep pf_deconv3d_blocks' shape), grid.y = channel groups of 16.
// Arithmetic: exact float32 fmaf chains in (quad, tap) order.  C_out = 8 fills half of the tile rows.
// ------------------------------------------------------------------------------------------------
typedef float dc_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kDmThreads = 512;
constexpr int kDmCo = 16;

template <bool ADD>
__global__ __launch_bounds__(kDmThreads) void deconv3d_k3s2_mfma_kernel(const float* __restrict__ xa,
                                                                          const float* __restrict__ xb,
                                                                          const float* __restrict__ w,
                                                                          float* __restrict__ y, int Cin, int Cout, int D,
                                                                          int H, int W, double* __restrict__ partials,
                                                                          DcAffine A) {
  extern __shared__ __attribute__((aligned(16))) float dm_lds[];
  float* wl = dm_lds;                                   // [Cin / 4][27][4][16]
  float* aff = wl + Cin * 27 * kDmCo;                   // [2][kDcAffMax]
  double* red = reinterpret_cast<double*>(aff + 2 * kDcAffMax);   // [8 waves][16][2]; first: pf_bn_resolve's scratch
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int nl = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.y * kDmCo;
  const int n = blockIdx.z;
  const int plane = H * W, vol = plane * D;

  // weights of this block's channel group: w (Cin, Cout, 27) -> wl[((ci >> 2) * 27 + tap) * 64 + (ci & 3) * 16 + c]
  for (int e = tid; e < Cin * kDmCo * 27; e += kDmThreads) {
    const int tap = e % 27, r = e / 27;
    const int c = r % kDmCo, ci = r / kDmCo;
    const float v = co0 + c < Cout ? w[((int64_t)ci * Cout + co0 + c) * 27 + tap] : 0.0f;
    wl[((ci >> 2) * 27 + tap) * 64 + (ci & 3) * kDmCo + c] = v;
  }
  if (A.mode == 1) {
    const int stat = n / A.sps;
    for (int e = tid; e < 2 * Cin; e += kDmThreads)
      aff[e < Cin ? e : kDcAffMax + e - Cin] = e < Cin ? A.scale[(int64_t)stat * Cin + e] : A.shift[(int64_t)stat * Cin + e - Cin];
  } else if (A.mode == 2) {
    pf_bn_resolve<kDmThreads>(A.bn, n / A.sps, aff, aff + kDcAffMax, red);
  }
  __syncthreads();

  const int cell = blockIdx.x * 128 + wave * 16 + nl;
  const bool live = cell < vol;
  const int c0 = live ? cell : 0;
  const int id = c0 / plane;
  const int rem = c0 - id * plane;
  const int ih = rem / W, iw = rem - ih * W;
  const bool vd = id + 1 < D, vh = ih + 1 < H, vw = iw + 1 < W;
  int off[8];
  bool keep[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int sd = s >> 2, sh = (s >> 1) & 1, sw = s & 1;
    keep[s] = live && (sd == 0 || vd) && (sh == 0 || vh) && (sw == 0 || vw);
    off[s] = keep[s] ? c0 + sd * plane + sh * W + sw : c0;
  }
  dc_f32x4 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) acc[p] = (dc_f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  const float* xan = xa + (int64_t)n * Cin * vol;
  const float* xbn = ADD ? xb + (int64_t)n * Cin * vol : nullptr;
  const int nq = Cin >> 2;
  // this lane's B operands of quad q: x[4 q + kq][cell + shift s], s = 0..7
  auto load_quad = [&](int q, float* v) {
    const int ci = 4 * q + kq;
    const float* pa = xan + (int64_t)ci * vol;
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = pa[off[s]];
    if (ADD) {
      const float* pb = xbn + (int64_t)ci * vol;
#pragma unroll
      for (int s = 0; s < 8; ++s) v[8 + s] = pb[off[s]];
    }
  };
  auto finish_quad = [&](int q, float* v) {
    const int ci = 4 * q + kq;
    if (A.mode) {                                                 // block-uniform
      const float sa = aff[ci], sb = aff[kDcAffMax + ci];
#pragma unroll
      for (int s = 0; s < 8; ++s) v[s] = fmaxf(fmaf(v[s], sa, sb), 0.0f);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (ADD) v[s] += v[8 + s];
      v[s] = keep[s] ? v[s] : 0.0f;
    }
  };
  constexpr int NV = ADD ? 16 : 8;
  float cur[NV], nxt[NV];
  load_quad(0, cur);
  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) load_quad(q + 1, nxt);                        // the next quad's loads fly under this quad's MFMAs
    finish_quad(q, cur);
    const float* wq = wl + q * 27 * 64 + lane;                    // A operand of (q, tap): wq[tap * 64]
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
            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[((kd * 3 + kh) * 3 + kw) * 64],
                                                          cur[(sd << 2) | (sh << 1) | sw], acc[p], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NV; ++s) cur[s] = nxt[s];
  }

  // D layout: lane holds rows (channels) 4 kq + r, column (cell) nl
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t plane_o = (int64_t)Ho * Wo, vol_o = plane_o * 2 * D;
  float ssum[4], ssq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ssum[r] = ssq[r] = 0.0f;
  if (live) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 4 * kq + r;
      if (co < Cout) {
        float* yc = y + ((int64_t)n * Cout + co) * vol_o + (int64_t)(2 * id) * plane_o + (int64_t)(2 * ih) * Wo + 2 * iw;
#pragma unroll
        for (int p = 0; p < 8; p += 2) {
          const int pd = p >> 2, ph = (p >> 1) & 1;
          const float2 o = make_float2(acc[p][r], acc[p + 1][r]);
          *reinterpret_cast<float2*>(yc + pd * plane_o + ph * Wo) = o;
          ssum[r] += o.x + o.y;
          ssq[r] += o.x * o.x + o.y * o.y;
        }
      }
    }
  }
  if (partials != nullptr) {
    __syncthreads();                                              // (red doubled as the resolve's scratch)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s = (double)ssum[r], q = (double)ssq[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {                          // the 16 cells of this row group
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
      }
      if (nl == 0) {
        red[(wave * kDmCo + 4 * kq + r) * 2 + 0] = s;
        red[(wave * kDmCo + 4 * kq + r) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < 2 * kDmCo && co0 + (tid >> 1) < Cout) {
      double v = 0.0;
#pragma unroll
      for (int wv = 0; wv < kDmThreads / 64; ++wv) v += red[(wv * kDmCo + (tid >> 1)) * 2 + (tid & 1)];
      partials[(((int64_t)n * gridDim.x + blockIdx.x) * Cout + co0 + (tid >> 1)) * 2 + (tid & 1)] = v;
    }
  }
}

size_t dm_lds_bytes(int Cin) {
  // weights + affine rows + max(statistics rows of the 8 waves, pf_bn_resolve's scratch for 512 threads)
  return sizeof(float) * ((size_t)Cin * 27 * kDmCo + 2 * kDcAffMax) + sizeof(double) * 2 * kDmThreads;
}

int launch_dm(const float* xa, const float* xb, const float* w, float* y, int64_t N, int Cin, int Cout, int D, int H, int W,
              double* partials, const DcAffine& A, hipStream_t s) {
  const size_t lds = dm_lds_bytes(Cin);
  dim3 grid((unsigned)pf_cdiv((int64_t)D * H * W, 128), (unsigned)pf_cdiv(Cout, kDmCo), (unsigned)N);
  if (xb != nullptr)
    hipLaunchKernelGGL((deconv3d_k3s2_mfma_kernel<true>), grid, dim3(kDmThreads), lds, s, xa, xb, w, y, Cin, Cout, D, H, W,
                       partials, A);
  else
    hipLaunchKernelGGL((deconv3d_k3s2_mfma_kernel<false>), grid, dim3(kDmThreads), lds, s, xa, xb, w, y, Cin, Cout, D, H, W,
                       partials, A);
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
  // The matrix-core form (whole channel quads, weights of 16 output channels within 64 KB of LDS) where it was measured
  // faster (profiles/r06f_deconv_mfma.md): every block stages Cin * 27 * 16 weights, which only pays with >= 2 channel
  // groups of 16 and a grid of >= 512 blocks -- config 4's data gradient of conv1_0 (16 -> 64 on 24x32x40: 35.6 us
  // against 49.1); the decoder's 32 -> 16 and 16 -> 8 layers (30 / 240 blocks) stay on the lane-per-cell form (12.5 /
  // 10.8 us against 17.8 / 18.7).  A shape always takes the same form, so results are reproducible run to run.
  // PF_DECONV_MFMA=1 / PF_DECONV_VALU=1 force one form wherever it is valid (tests, tools/microbench_deconv3d.py).
  const bool dm_ok = (Cin & 3) == 0 && Cin <= 32;
  const bool dm_pays = Cout >= 32 && pf_cdiv(D * H * W, 128) * pf_cdiv(Cout, kDmCo) * N >= 512;
  if (dm_ok && getenv("PF_DECONV_VALU") == nullptr && (dm_pays || getenv("PF_DECONV_MFMA") != nullptr))
    return launch_dm(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
  // channel group: 4 when that still leaves >= 2 blocks per CU, else 2 / 1 (more, smaller work items)
  const int64_t cell_blocks = pf_cdiv(D * H * W, kDcThreads);
  if ((Cout % 4) == 0 && cell_blocks * (Cout / 4) * N >= 512)
    return launch_dc<4>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
  if ((Cout % 2) == 0 && cell_blocks * (Cout / 2) * N >= 512)
    return launch_dc<2>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
  return launch_dc<1>(xa, xb, w, y, N, (int)Cin, (int)Cout, (int)D, (int)H, (int)W, partials, A, s);
}

}  // extern "C"
