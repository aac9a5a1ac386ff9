// Row Z (training step, BASELINE config 4): the DATA gradients the forward kernels cannot produce by a weight
// transform.  (Stride-1 same-width layers: dx = conv(dy, flipped W^T) on the forward kernels; 3-D stride-2 layers:
// the transposed-convolution kernels of deconv3d.hip / conv3d_bottom.hip; ConvTranspose3d layers: the stride-2
// forward convolution -- pointmvsnet_amd/train_ops.py does those mappings.)  What is left is the data gradient of
// ImageConv's 5x5 / stride 2 / pad 2 convolutions (reference networks.py:93,98,103 through nn/conv.py:62-77):
//
//   dx[n][ci][Y][X] = sum_{co} sum_{kh = Y mod 2 (2) 4} sum_{kw = X mod 2 (2) 4}
//                     dy[n][co][(Y + 2 - kh) / 2][(X + 2 - kw) / 2] * W[co][ci][kh][kw]            (zero outside dy)
//
// i.e. ConvTranspose2d(5, stride 2, pad 2, output_padding 1).  The four output parity classes (py, px) are four
// stride-1 convolutions of dy with 3x3 / 3x2 / 2x3 / 2x2 sub-kernels (25 taps in all): an implicit GEMM per class
// with M = half-resolution positions, N = Cin, K = Cout x taps of the class, on v_mfma_f32_16x16x4_f32.
//   * block = 4 x 16 half-resolution positions (wave w owns row w) -> 8 x 32 output pixels x Cin channels;
//   * K walked in groups of 4 dy channels: patch 4 x 6 x 18 (zero filled) + the group's 25 x 4 x Cin weights in
//     LDS, double buffered through registers like conv3d.hip;
//   * the nine (row, column) offsets of the patch are loaded once per group and feed the 25 taps of the four
//     classes (A operand reuse 2.8x);
//   * epilogue: a lane holds, per (class row, channel), 4 positions x 2 column parities = 8 consecutive output
//     floats: two 16-byte stores.
// Bound: fp32 MFMA (2 * 25 * Cin * Cout flop per dy position); Cin = 8 fills half of the 16 MFMA columns.
#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPR = 6, kPC = 18, kPCP = 19;          // patch rows, columns, padded row
constexpr int kPlane = 6 * 19 + 14;                  // 128: channel planes 0 banks apart would collide -> +16 below
constexpr int kPlaneP = kPlane + 16;                 // 144 = 16 mod 32: the two lk planes of a 32-lane group differ

template <int NT>
__global__ __launch_bounds__(256) void deconv2d_k5s2_kernel(const float* __restrict__ dy, const float* __restrict__ wp,
                                                            float* __restrict__ dx, int Cout, int Cin, int Ho, int Wo,
                                                            int tiles_h, int tiles_w) {
  constexpr int NCP = NT * 16;
  constexpr int WSZ = 25 * 4 * NCP;
  constexpr int NWR = (WSZ + 255) / 256;
  constexpr int XS = 4 * kPlaneP;
  __shared__ __attribute__((aligned(16))) float xs0[2 * XS];
  __shared__ __attribute__((aligned(16))) float ws0[2 * WSZ];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int n = blockIdx.y;
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w;
  const int y0 = th * 4, x0 = tw * 16;
  const int plane_o = Ho * Wo;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int64_t plane_i = (int64_t)Hi * Wi;
  const float* yb = dy + (int64_t)n * Cout * plane_o;
  float* xb = dx + (int64_t)n * Cin * plane_i;
  const int cgroups = Cout >> 2;

  // staging plan (same for every channel group): wave w stages channel w; lane takes elements l, l + 64 of 6 x 18
  unsigned gofs[2];
  unsigned okmask = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int e = lane + 64 * r;
    const int row = e / kPC, col = e - row * kPC;
    const int iy = y0 - 1 + row, ix = x0 - 1 + col;
    const bool ok = e < kPR * kPC && iy >= 0 && iy < Ho && ix >= 0 && ix < Wo;
    gofs[r] = ok ? (unsigned)(iy * Wo + ix) : 0u;
    okmask |= (ok ? 1u : 0u) << r;
  }
  float rx[2], rw[NWR];
  auto load_group = [&](int cg) {
    const float* src = yb + (int64_t)(cg * 4 + wave) * plane_o;
#pragma unroll
    for (int r = 0; r < 2; ++r) rx[r] = src[gofs[r]];
    const float* wsrc = wp + (int64_t)cg * WSZ;
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      rw[r] = e < WSZ ? wsrc[e] : 0.0f;
    }
  };
  auto store_group = [&](int buf) {
    float* xs = xs0 + buf * XS + wave * kPlaneP;
    float* ws = ws0 + buf * WSZ;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int e = lane + 64 * r;
      if (e < kPR * kPC) xs[e + e / kPC] = ((okmask >> r) & 1u) ? rx[r] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int e = tid + 256 * r;
      if (e < WSZ) ws[e] = rw[r];
    }
  };

  f32x4 acc[2][2][NT];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[py][px][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  load_group(0);
  store_group(0);
  __syncthreads();
  for (int cg = 0; cg < cgroups; ++cg) {
    const int buf = cg & 1;
    if (cg + 1 < cgroups) load_group(cg + 1);
    const float* xs = xs0 + buf * XS + lk * kPlaneP + wave * kPCP + li;
    const float* ws = ws0 + buf * WSZ + lk * NCP + li;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        // dy at (y' + 1 - a, x' + 1 - b): patch row w + 2 - a, column li + 2 - b
        const float av = xs[(2 - a) * kPCP + (2 - b)];
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          if (py + 2 * a > 4) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            if (px + 2 * b > 4) continue;
            const int tap = (py + 2 * a) * 5 + (px + 2 * b);
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[py][px][t] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(av, ws[tap * 4 * NCP + 16 * t], acc[py][px][t], 0, 0, 0);
          }
        }
      }
    }
    if (cg + 1 < cgroups) store_group(buf ^ 1);
    __syncthreads();
  }

  // C/D layout: col = lane & 15 (channel 16 t + li), row = (lane >> 4) * 4 + r (position x' = x0 + 4 lk + r)
  const int yh = y0 + wave;
  if (yh < Ho) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ci = 16 * t + li;
      if (ci >= Cin) continue;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float* row = xb + (int64_t)ci * plane_i + (int64_t)(2 * yh + py) * Wi + 2 * (x0 + 4 * lk);
        const f32x4 lo = (f32x4){acc[py][0][t][0], acc[py][1][t][0], acc[py][0][t][1], acc[py][1][t][1]};
        const f32x4 hi = (f32x4){acc[py][0][t][2], acc[py][1][t][2], acc[py][0][t][3], acc[py][1][t][3]};
        const int xq = x0 + 4 * lk;
        if (xq + 3 < Wo && ((Wi & 3) == 0)) {
          *reinterpret_cast<f32x4*>(row) = lo;
          *reinterpret_cast<f32x4*>(row + 4) = hi;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (xq + r < Wo) {
              row[2 * r] = acc[py][0][t][r];
              row[2 * r + 1] = acc[py][1][t][r];
            }
          }
        }
      }
    }
  }
}

// 3x3x3 / pad 1 / stride 1 conv3d with ONE input channel and up to 8 output channels: the data gradient of
// VolumeConv's 8 -> 1 output layer (reference networks.py:147; dx = conv(dy, flipped W^T)): lane = voxel.
__global__ __launch_bounds__(256) void conv3d_k3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int Cout, int D, int H, int W) {
  __shared__ float wl[8 * 27];
  for (int e = threadIdx.x; e < Cout * 27; e += 256) wl[e] = w[e];
  __syncthreads();
  const int64_t plane = (int64_t)H * W, vol = plane * D;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= vol) return;
  const int od = (int)(i / plane);
  const int rem = (int)(i - (int64_t)od * plane);
  const int oh = rem / W, ow = rem - oh * W;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0f;
  const float* xc = x + (int64_t)n * vol;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd) {
    const int id = od + kd - 1;
    const bool dok = id >= 0 && id < D;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh + kh - 1;
      const bool hok = dok && ih >= 0 && ih < H;
      const float* row = xc + (int64_t)id * plane + (int64_t)ih * W;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow + kw - 1;
        const float v = (hok && iw >= 0 && iw < W) ? row[iw] : 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < Cout) acc[c] = fmaf(v, wl[c * 27 + (kd * 3 + kh) * 3 + kw], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (c < Cout) y[((int64_t)n * Cout + c) * vol + i] = acc[c];
}

}  // namespace

extern "C" {

int pf_deconv2d_k5s2_supported(int64_t Cin_dy, int64_t Cout_dx) {
  return (Cin_dy >= 4 && (Cin_dy % 4) == 0 && Cout_dx >= 1 && Cout_dx <= 32) ? 1 : 0;
}

int pf_deconv2d_k5s2_f32(const float* dy, const float* wp, float* dx, int64_t N, int64_t Cout, int64_t Cin, int64_t Ho,
                         int64_t Wo, void* stream) {
  PF_REQUIRE(N >= 0 && Cout >= 1 && Cin >= 1 && Ho >= 1 && Wo >= 1 && N <= 65535);
  if (!pf_deconv2d_k5s2_supported(Cout, Cin)) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cout * Ho * Wo <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(dy && wp && dx);
  const int tiles_h = (int)((Ho + 3) / 4), tiles_w = (int)((Wo + 15) / 16);
  dim3 grid((unsigned)(tiles_h * tiles_w), (unsigned)N);
  hipStream_t s = (hipStream_t)stream;
  if (Cin <= 16)
    hipLaunchKernelGGL(deconv2d_k5s2_kernel<1>, grid, dim3(256), 0, s, dy, wp, dx, (int)Cout, (int)Cin, (int)Ho, (int)Wo,
                       tiles_h, tiles_w);
  else
    hipLaunchKernelGGL(deconv2d_k5s2_kernel<2>, grid, dim3(256), 0, s, dy, wp, dx, (int)Cout, (int)Cin, (int)Ho, (int)Wo,
                       tiles_h, tiles_w);
  return pf_launch_status();
}

int pf_conv3d_k3_c1_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cout, int64_t D, int64_t H,
                        int64_t W, void* stream) {
  PF_REQUIRE(N >= 0 && Cout >= 1 && D >= 1 && H >= 1 && W >= 1 && N <= 65535);
  if (Cout > 8) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(H * W <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && w && y);
  dim3 grid((unsigned)pf_cdiv(D * H * W, 256), (unsigned)N);
  hipLaunchKernelGGL(conv3d_k3_c1_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, y, (int)Cout, (int)D, (int)H,
                     (int)W);
  return pf_launch_status();
}

}  // extern "C"
