// Row R, the first layer of VolumeConv (conv0_1: 64 -> 8 channels over the full 48x64x80 cost volume, 69 % of the
// regulariser's FLOPs): 3x3x3 / pad 1 / stride 1 conv3d with C_out <= 8 on the f32 matrix cores WITHOUT the
// half-empty tile.
//
// conv3d.hip maps N of v_mfma_f32_16x16x4_f32 to 16 output channels; with 8 of them half of every MFMA
// multiplies zeros, and SQ counters show that kernel's matrix pipe 62 % busy -- it is MFMA-bound on wasted work
// (profiles/archive/r02/r02r_sq_counters.md).  Here N = 8 channels x 2 ADJACENT OUTPUT ROWS (h = 2p + s, s in {0,1}):
//     out[c][2p+s][x] = sum_{kd, kh, kw, ci} in[ci][..][2p + s + kh - 1][x + kw - 1] W[c][ci][kd][kh][kw]
// with kh' = s + kh in [0,3] both rows read the SAME four input rows 2p - 1 + kh', so one A operand
// (16 x-positions x 4 channels at (kd, kh', kw)) feeds both; the weights are packed as
//     B[(kd, kh', kw), ci][c + 8 s] = W[c][ci][kd][kh' - s][kw]   (zero where kh' - s is outside [0,2]).
// K grows from 27 to 36 taps, M halves: 1.5x fewer MFMA cycles, every column of the tile used.
// Everything else -- channel groups of 4 double-buffered through LDS, wave w staging channel w as one flat index,
// compile-time sub-volume geometry, per-wave LDS transpose in the epilogue, float64 BatchNorm partials -- is
// conv3d.hip's v3 structure.  A block owns TD x 8 x 16 output voxels; wave w owns the row pair (2w, 2w+1).
// Same exact float32 arithmetic as an fmaf chain; the summation order over (group, kd, kh', kw, channel) differs
// from conv3d.hip's, so the two agree to rounding.
#include <stdlib.h>

#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PairGeom {
  int Cin, Cout, D, H, W;
  int tiles_d, tiles_h, tiles_w;
};

template <int TD>
struct PStage {
  static constexpr int ID = TD + 2;
  static constexpr int IH = 10;                      // 8 output rows + halo
  static constexpr int IW = 18;
  static constexpr int IWP = IW + 1;
  static constexpr int RAW = ID * IH * IWP;
  static constexpr int PLANE = RAW + ((16 - RAW % 32) + 32) % 32;   // channel planes 16 banks apart
  static constexpr int ELEMS = ID * IH * IW;
  static constexpr int NXR = (ELEMS + 63) / 64;
};

constexpr int kTaps = 36;                            // (kd, kh', kw) = 3 x 4 x 3
constexpr int kWsz = kTaps * 4 * 16;                 // weights of one channel group

template <int TD>
constexpr size_t pair_lds_bytes() {
  using St = PStage<TD>;
  const size_t staging = sizeof(float) * (size_t)(2 * 4 * St::PLANE + 2 * kWsz);
  const size_t epilogue = sizeof(float) * (size_t)(4 * 16 * 17 + 1) + sizeof(double) * (size_t)(4 * 16 * 2);
  return staging > epilogue ? staging : epilogue;
}

template <int TD, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3d_k3_pair_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ wp,
                                                                   float* __restrict__ y, PairGeom g,
                                                                   double* __restrict__ partials) {
  using St = PStage<TD>;
  constexpr int ID = St::ID, IH = St::IH, IW = St::IW, IWP = St::IWP, PLANE = St::PLANE;
  constexpr int ELEMS = St::ELEMS, NXR = St::NXR;
  constexpr int NWR = (kWsz + 255) / 256;
  constexpr int XS = 4 * PLANE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs0 = lds;
  float* ws0 = lds + 2 * XS;
  float* tile = lds;                                 // [4 waves][16][17], aliases the staging buffers
  double* red = reinterpret_cast<double*>(tile + 4 * 16 * 17 + ((4 * 16 * 17) & 1));   // [4][16][2]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  unsigned xb_, xn_;
  pf_xcd_xy<PF_XCD_CONV3D>(xb_, xn_);                 // XCD x owns a contiguous run of each pass's items (pf_common.h)
  const int n = (int)xn_, bx = (int)xb_;
  const int plane_i = g.H * g.W, vol = plane_i * g.D;            // Cin * vol < 2^31 (checked on the host)
  const float* xb = x + (int64_t)n * g.Cin * vol;
  float* yb = y + (int64_t)n * g.Cout * vol;
  const int cgroups = g.Cin >> 2;

  double ssum = 0.0, ssq = 0.0;

  const int total = g.tiles_d * g.tiles_h * g.tiles_w;
  for (int item = bx; item < total; item += gridDim.x) {
    const int tw = item % g.tiles_w;
    const int rest = item / g.tiles_w;
    const int th = rest % g.tiles_h;
    const int td = rest / g.tiles_h;
    const int od0 = td * TD, oh0 = th * 8, ow0 = tw * 16;
    const int id0 = od0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;

    unsigned gofs[NXR];
    unsigned okmask = 0;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int e = lane + 64 * r;
      const int row = e / IW, col = e - row * IW;
      const int dz = row / IH, hy = row - dz * IH;
      const int id = id0 + dz, ih = ih0 + hy, iw = iw0 + col;
      const bool ok = e < ELEMS && id >= 0 && id < g.D && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      gofs[r] = ok ? (unsigned)(id * plane_i + ih * g.W + iw) : 0u;
      okmask |= (ok ? 1u : 0u) << r;
    }

    float rx[NXR], rw[NWR];
    auto load_group = [&](int cg) {
      const float* src = xb + (int64_t)(cg * 4 + wave) * vol;        // wave-uniform base
#pragma unroll
      for (int r = 0; r < NXR; ++r) rx[r] = src[gofs[r]];
      const float* wsrc = wp + (int64_t)cg * kWsz;
#pragma unroll
      for (int r = 0; r < NWR; ++r) {
        const int e = tid + 256 * r;
        rw[r] = (256 * (r + 1) <= kWsz || e < kWsz) ? wsrc[e] : 0.0f;
      }
    };
    auto store_group = [&](int buf) {
      float* xs = xs0 + buf * XS + wave * PLANE;
      float* ws = ws0 + buf * kWsz;
#pragma unroll
      for (int r = 0; r < NXR; ++r) {
        const int e = lane + 64 * r;
        if (64 * (r + 1) <= ELEMS || e < ELEMS) xs[e + e / IW] = ((okmask >> r) & 1u) ? rx[r] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < NWR; ++r) {
        const int e = tid + 256 * r;
        if (256 * (r + 1) <= kWsz || e < kWsz) ws[e] = rw[r];
      }
    };

    f32x4 acc[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d) acc[d] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    __syncthreads();                       // the previous tile's epilogue / last group has been consumed
    load_group(0);
    store_group(0);
    __syncthreads();
    for (int cg = 0; cg < cgroups; ++cg) {
      const int buf = cg & 1;
      if (cg + 1 < cgroups) load_group(cg + 1);
      // A: lane (x = li, channel lk) at input row 2*wave + kh' of the staged rows [oh0 - 1, oh0 + 9)
      const float* xs = xs0 + buf * XS + lk * PLANE + (2 * wave) * IWP + li;
      const float* ws = ws0 + buf * kWsz + lk * 16 + li;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tap = (kd * 4 + kh) * 3 + kw;
            const float b = ws[tap * 4 * 16];
#pragma unroll
            for (int d = 0; d < TD; ++d) {
              const float a = xs[((d + kd) * IH + kh) * IWP + kw];
              acc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[d], 0, 0, 0);
            }
          }
        }
      }
      if (cg + 1 < cgroups) store_group(buf ^ 1);
      __syncthreads();
    }

    // epilogue: C/D layout col = lane&15 = (channel c = col & 7, row parity s = col >> 3), row = (lane>>4)*4 + r
    // = voxel along W
    const int oh = oh0 + 2 * wave + (li >> 3);
    float* tl = tile + wave * 16 * 17;
#pragma unroll
    for (int d = 0; d < TD; ++d) {
      const int od = od0 + d;
      const bool row_ok = oh < g.H && od < g.D;
      float s = 0.0f, q = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pos = lk * 4 + r;
        const float v = acc[d][r];
        tl[li * 17 + pos] = v;
        if (row_ok && ow0 + pos < g.W) {
          s += v;
          q += v * v;
        }
      }
      s += __shfl_xor(s, 8);               // the other row of the pair: same channel
      q += __shfl_xor(q, 8);
      s += __shfl_xor(s, 16);
      q += __shfl_xor(q, 16);
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      ssum += (double)s;
      ssq += (double)q;
      __builtin_amdgcn_wave_barrier();
      if (od < g.D) {
        for (int e = lane; e < 16 * 16; e += 64) {
          const int col = e >> 4, pos = e & 15;
          const int co = col & 7, ohh = oh0 + 2 * wave + (col >> 3);
          if (co < g.Cout && ohh < g.H && ow0 + pos < g.W)
            yb[(int64_t)co * vol + (int64_t)od * plane_i + (int64_t)ohh * g.W + ow0 + pos] = tl[col * 17 + pos];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  if (partials != nullptr) {
    if (lane < 8) {
      red[(wave * 16 + lane) * 2 + 0] = ssum;
      red[(wave * 16 + lane) * 2 + 1] = ssq;
    }
    __syncthreads();
    if (tid < g.Cout) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[(w * 16 + tid) * 2 + 0];
        q += red[(w * 16 + tid) * 2 + 1];
      }
      double* o = partials + (((int64_t)n * gridDim.x + bx) * g.Cout + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

#ifndef PF_PAIR_TD
#define PF_PAIR_TD 2
#endif
#ifndef PF_PAIR_MINW
#define PF_PAIR_MINW 3
#endif
constexpr int kPairTD = PF_PAIR_TD;

PairGeom make_pair_geom(int64_t Cin, int64_t Cout, int64_t D, int64_t H, int64_t W) {
  PairGeom g;
  g.Cin = (int)Cin;
  g.Cout = (int)Cout;
  g.D = (int)D;
  g.H = (int)H;
  g.W = (int)W;
  g.tiles_d = (g.D + kPairTD - 1) / kPairTD;
  g.tiles_h = (g.H + 7) / 8;
  g.tiles_w = (g.W + 15) / 16;
  return g;
}

int pair_blocks(const PairGeom& g) {
  const int64_t total = (int64_t)g.tiles_d * g.tiles_h * g.tiles_w;
#ifndef PF_CONV3D_CAP
#define PF_CONV3D_CAP 2048
#endif
  if (total <= PF_CONV3D_CAP) return (int)total;
  const int64_t per = (total + PF_CONV3D_CAP - 1) / PF_CONV3D_CAP;      // every block the same number of tiles
  return (int)((total + per - 1) / per);
}

}  // namespace

extern "C" {

int pf_conv3d_pair_blocks(int64_t Cin, int64_t Cout, int64_t D, int64_t H, int64_t W) {
  if (Cin <= 0 || Cout <= 0 || Cout > 8 || D <= 0 || H <= 0 || W <= 0) return 0;
  return pair_blocks(make_pair_geom(Cin, Cout, D, H, W));
}

int pf_conv3d_k3_pair_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t D,
                          int64_t H, int64_t W, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 4 && Cout >= 1 && D >= 1 && H >= 1 && W >= 1 && N <= 65535);
  if ((Cin % 4) != 0 || Cout > 8) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cin * D * H * W <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  const PairGeom g = make_pair_geom(Cin, Cout, D, H, W);
  constexpr size_t lds_bytes = pair_lds_bytes<kPairTD>();
  static_assert(lds_bytes <= 64 * 1024, "pair tile must fit the default dynamic LDS limit");
  dim3 grid((unsigned)pair_blocks(g), (unsigned)N);
  hipLaunchKernelGGL((conv3d_k3_pair_kernel<kPairTD, PF_PAIR_MINW>), grid, dim3(256), lds_bytes, (hipStream_t)stream, x, wp, y, g,
                     partials);
  return pf_launch_status();
}

}  // extern "C"
