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
// Structure (v2; v1 fed the MFMA A/B operands straight from global memory and was texture-address bound
// at 16 TF/s):
//   * a 256-thread block owns an output tile of TD x 4 x 16 voxels (d x h x w); wave w owns h-row w, i.e.
//     TD segments of 16 voxels along W, and all NT column tiles of 16 output channels;
//   * K is walked in groups of 4 input channels.  For each group the block stages the input sub-volume
//     it needs -- 4 x ((TD-1)s+3) x (3s+3) x (15s+3) voxels, zero-filled outside the tensor, so the
//     compute loop has no boundary code -- and the group's 27 x 4 x 16NT weights into LDS, double
//     buffered: group g+1 travels global -> registers while the matrix cores work on group g.  Wave w
//     stages channel w of the group, its lanes walking the channel's sub-volume as one flat index (no lane
//     idles on row padding; one 32-bit offset per element, computed once per tile, wave-uniform base);
//     the sub-volume geometry is a compile-time function of (stride, TD), so every LDS address in the
//     MFMA loop is one base register + an immediate (v3: 132 -> 42 us on the 64 -> 16 stride-2 layer,
//     168 -> 131 us on 64 -> 8, profiles/archive/r01/r01g_microbench_conv3d.log);
//   * v_mfma_f32_16x16x4_f32: lane l (row i = l&15, k = l>>4) reads A = xs[k][d*s+kd][h*s+kh][i*s+kw] and
//     B = ws[tap][k][16t + i] from LDS (plane stride padded to 16 mod 32 banks: conflict-free); a B value
//     is reused for the TD depth slices.  Exact float32: an fmaf chain over (channel group, tap, channel).
//   * epilogue: accumulators go through a per-wave LDS transpose so that each channel's 16 voxels leave as
//     one 64-byte segment, and per-channel sum / sum-of-squares are folded into float64 block partials
//     (the (N, T, C, 2) layout pf_bn_finalize_f32 / pf_channel_bn_apply_f32 consume).
// Bound: fp32 MFMA (2*27*C_in*C_out flop per voxel vs 4*(C_in + C_out) bytes).
#include <stdlib.h>

#include "pf_common.h"
#include "pf_bn_resolve.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
  int Cin, Cout, Di, Hi, Wi, Do, Ho, Wo;
  int ID, IH, IW, IWP, plane;   // staged sub-volume: depth, height, width, padded row, padded channel stride
  int tiles_d, tiles_h, tiles_w;
  // the pending BatchNorm + ReLU of the INPUT, applied while a channel is staged (zero padding after it):
  // aff_mode 0 = none, 1 = (scale, shift) rows (N / sps, Cin), 2 = resolved by every block from the producer's
  // statistics rows (pf_bn_resolve, pf_bn_resolve.h).  aff_off: float offset of the 2 * Cin affine values in LDS.
  int aff_mode, sps, aff_off;
  const float* in_scale;
  const float* in_shift;
  pf_bn_job in_bn;
};

// Staged sub-volume of one input channel for a TD x 4 x 16 output tile: compile-time so that every LDS
// address of the MFMA loop is base register + immediate.
template <int STRIDE, int TD>
struct Stage {
  static constexpr int ID = (TD - 1) * STRIDE + 3;
  static constexpr int IH = 3 * STRIDE + 3;
  static constexpr int IW = 15 * STRIDE + 3;
  static constexpr int IWP = IW + 1;
  static constexpr int RAW = ID * IH * IWP;
  // channel planes 16 banks apart for unit-stride reads, 17 for stride-2 reads (the 16 lanes of a plane
  // then use every other bank): the two planes of a 32-lane LDS group never collide
  static constexpr int WANT = STRIDE == 1 ? 16 : 17;
  static constexpr int PLANE = RAW + ((WANT - RAW % 32) + 32) % 32;
  static constexpr int ELEMS = ID * IH * IW;        // floats of one channel's sub-volume
  static constexpr int NXR = (ELEMS + 63) / 64;     // ... per lane of the wave that stages the channel
};

template <int NT, int STRIDE, int TD, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3d_k3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                        float* __restrict__ y, ConvGeom g,
                                                        double* __restrict__ partials) {
  using St = Stage<STRIDE, TD>;
  constexpr int ID = St::ID, IH = St::IH, IW = St::IW, IWP = St::IWP, PLANE = St::PLANE;
  constexpr int ELEMS = St::ELEMS, NXR = St::NXR;
  constexpr int NCP = NT * 16;
  constexpr int WSZ = 27 * 4 * NCP;                 // weights of one channel group
  constexpr int NWR = (WSZ + 255) / 256;            // weight floats staged per thread
  constexpr int XS = 4 * PLANE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs0 = lds;
  float* ws0 = lds + 2 * XS;
  // the epilogue buffers alias the staging buffers (all of a tile's MFMA reads are behind the last barrier
  // of its channel loop)
  float* tile = lds;                                 // [4 waves][NCP][17]
  double* red = reinterpret_cast<double*>(tile + 4 * NCP * 17 + ((4 * NCP * 17) & 1));   // [4][NCP][2]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  unsigned xb_, xn_;
  pf_xcd_xy<PF_XCD_CONV3D>(xb_, xn_);                 // XCD x owns a contiguous run of each pass's items (pf_common.h)
  const int n = (int)xn_, bx = (int)xb_;
  const int plane_i = g.Hi * g.Wi, vol_i = plane_i * g.Di;       // Cin * vol_i < 2^31 (checked on the host)
  const int64_t plane_o = (int64_t)g.Ho * g.Wo, vol_o = plane_o * g.Do;
  const float* xb = x + (int64_t)n * g.Cin * vol_i;
  float* yb = y + (int64_t)n * g.Cout * vol_o;
  const int cgroups = g.Cin >> 2;

  double ssum[NT], ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ssum[t] = ssq[t] = 0.0;

  float* aff = lds + g.aff_off;                      // [scale Cin | shift Cin] of the input's pending BatchNorm
  if (g.aff_mode == 1) {
    const int stat = n / g.sps;
    for (int e = tid; e < 2 * g.Cin; e += 256)
      aff[e] = e < g.Cin ? g.in_scale[(int64_t)stat * g.Cin + e] : g.in_shift[(int64_t)stat * g.Cin + e - g.Cin];
  } else if (g.aff_mode == 2) {
    pf_bn_resolve<256>(g.in_bn, n / g.sps, aff, aff + g.Cin, reinterpret_cast<double*>(lds));
  }

  const int total = g.tiles_d * g.tiles_h * g.tiles_w;
  for (int item = bx; item < total; item += gridDim.x) {
    const int tw = item % g.tiles_w;
    const int rest = item / g.tiles_w;
    const int th = rest % g.tiles_h;
    const int td = rest / g.tiles_h;
    const int od0 = td * TD, oh0 = th * 4, ow0 = tw * 16;
    const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;

    // Staging plan of this tile, the same for every channel group: wave w stages channel w of the group;
    // lane l takes the elements e = l + 64 r of the channel's ID x IH x IW sub-volume (flat, so no lane is
    // wasted on row padding).  Outside the tensor: offset 0 (a valid address) and a cleared mask bit.
    unsigned gofs[NXR];
    unsigned okmask = 0;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int e = lane + 64 * r;
      const int row = e / IW, col = e - row * IW;
      const int dz = row / IH, hy = row - dz * IH;
      const int id = id0 + dz, ih = ih0 + hy, iw = iw0 + col;
      const bool ok = e < ELEMS && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi;
      gofs[r] = ok ? (unsigned)(id * plane_i + ih * g.Wi + iw) : 0u;
      okmask |= (ok ? 1u : 0u) << r;
    }

    float rx[NXR], rw[NWR];
    auto load_group = [&](int cg) {
      const float* src = xb + (int64_t)(cg * 4 + wave) * vol_i;       // wave-uniform base
#pragma unroll
      for (int r = 0; r < NXR; ++r) rx[r] = src[gofs[r]];
      const float* wsrc = wp + (int64_t)cg * WSZ;
#pragma unroll
      for (int r = 0; r < NWR; ++r) {
        const int e = tid + 256 * r;
        rw[r] = (256 * (r + 1) <= WSZ || e < WSZ) ? wsrc[e] : 0.0f;
      }
    };
    auto store_group = [&](int buf, int cg) {
      float* xs = xs0 + buf * XS + wave * PLANE;
      float* ws = ws0 + buf * WSZ;
      float sa = 1.0f, sb = 0.0f;
      if (g.aff_mode) {                                // wave-uniform: this wave stages channel 4 cg + wave
        sa = aff[cg * 4 + wave];
        sb = aff[g.Cin + cg * 4 + wave];
      }
#pragma unroll
      for (int r = 0; r < NXR; ++r) {
        const int e = lane + 64 * r;
        float v = rx[r];
        if (g.aff_mode) v = fmaxf(fmaf(v, sa, sb), 0.0f);
        if (64 * (r + 1) <= ELEMS || e < ELEMS) xs[e + e / IW] = ((okmask >> r) & 1u) ? v : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < NWR; ++r) {
        const int e = tid + 256 * r;
        if (256 * (r + 1) <= WSZ || e < WSZ) ws[e] = rw[r];
      }
    };

    f32x4 acc[TD][NT];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[d][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    __syncthreads();                       // the previous tile's epilogue / last group has been consumed
    load_group(0);
    store_group(0, 0);
    __syncthreads();
    for (int cg = 0; cg < cgroups; ++cg) {
      const int buf = cg & 1;
      if (cg + 1 < cgroups) load_group(cg + 1);
      const float* xs = xs0 + buf * XS + lk * PLANE + (wave * STRIDE) * IWP + li * STRIDE;
      const float* ws = ws0 + buf * WSZ + lk * NCP + li;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tap = (kd * 3 + kh) * 3 + kw;
            float b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = ws[tap * 4 * NCP + 16 * t];
#pragma unroll
            for (int d = 0; d < TD; ++d) {
              const float a = xs[((d * STRIDE + kd) * IH + kh) * IWP + kw];
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[d][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[d][t], 0, 0, 0);
            }
          }
        }
      }
      if (cg + 1 < cgroups) store_group(buf ^ 1, cg + 1);
      __syncthreads();
    }

    // epilogue: C/D layout col = lane&15 (channel), row = (lane>>4)*4 + r (voxel along W)
    const int oh = oh0 + wave;
    float* tl = tile + wave * NCP * 17;
#pragma unroll
    for (int d = 0; d < TD; ++d) {
      const int od = od0 + d;
      const bool row_ok = oh < g.Ho && od < g.Do;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float s = 0.0f, q = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pos = lk * 4 + r;
          const float v = acc[d][t][r];
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
      // `tl` is private to the wave and a wave's LDS operations execute in order: no workgroup barrier,
      // only keep the compiler from moving the reads above the writes
      __builtin_amdgcn_wave_barrier();
      if (row_ok) {
        for (int e = lane; e < NCP * 16; e += 64) {
          const int co = e >> 4, pos = e & 15;
          if (co < g.Cout && ow0 + pos < g.Wo)
            yb[(int64_t)co * vol_o + (int64_t)od * plane_o + (int64_t)oh * g.Wo + ow0 + pos] = tl[co * 17 + pos];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
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
      double* o = partials + (((int64_t)n * gridDim.x + bx) * g.Cout + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

// Few output channels (VolumeConv's last layer, 8 -> 1: reference networks.py:147): no GEMM shape to speak
// of -- 216 multiply-adds per voxel against 36 bytes moved -- so one lane per output voxel, weights in
// LDS, taps served by L1 (neighbouring lanes share 2 of 3 taps along W, neighbouring rows/planes by L2).
template <int COUT>
__global__ __launch_bounds__(256) void conv3d_k3_few_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, int Cin, int D, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float wl[];     // [COUT][Cin][27]
  for (int e = threadIdx.x; e < COUT * Cin * 27; e += 256) wl[e] = w[e];
  __syncthreads();
  const int64_t plane = (int64_t)H * W, vol = plane * D;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= vol) return;
  const int od = (int)(i / plane);
  const int rem = (int)(i - (int64_t)od * plane);
  const int oh = rem / W, ow = rem - oh * W;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.0f;
  const float* xb = x + (int64_t)n * Cin * vol;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xb + (int64_t)ci * vol;
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
          for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v, wl[(c * Cin + ci) * 27 + (kd * 3 + kh) * 3 + kw], acc[c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < COUT; ++c) y[((int64_t)n * COUT + c) * vol + i] = acc[c];
}

ConvGeom make_geom(int64_t Cin, int64_t Cout, int64_t Di, int64_t Hi, int64_t Wi, int stride, int td) {
  ConvGeom g;
  g.Cin = (int)Cin;
  g.Cout = (int)Cout;
  g.Di = (int)Di;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.Do = (int)((Di - 1) / stride + 1);
  g.Ho = (int)((Hi - 1) / stride + 1);
  g.Wo = (int)((Wi - 1) / stride + 1);
  g.ID = (td - 1) * stride + 3;
  g.IH = 3 * stride + 3;
  g.IW = 15 * stride + 3;
  g.IWP = g.IW + 1;
  const int raw = g.ID * g.IH * g.IWP;
  // channel planes 16 banks apart for unit-stride reads, 17 for stride-2 reads (the 16 lanes of a plane
  // then use every other bank): the two planes of a 32-lane LDS group never collide
  const int want = stride == 1 ? 16 : 17;
  g.plane = raw + ((want - raw % 32) + 32) % 32;
  g.tiles_d = (g.Do + td - 1) / td;
  g.tiles_h = (g.Ho + 3) / 4;
  g.tiles_w = (g.Wo + 15) / 16;
  g.aff_mode = 0;
  g.sps = 1;
  g.aff_off = 0;
  g.in_scale = g.in_shift = nullptr;
  g.in_bn = pf_bn_job{};
  return g;
}

constexpr size_t kMaxLds = 80 * 1024;    // two blocks per CU still fit in the 160 KiB of a CU

size_t lds_work_bytes(const ConvGeom& g, int NT) {
  const int NCP = NT * 16;
  const size_t staging = sizeof(float) * (size_t)(2 * 4 * g.plane + 2 * 27 * 4 * NCP);
  const size_t epilogue = sizeof(float) * (size_t)(4 * NCP * 17 + 1) + sizeof(double) * (size_t)(4 * NCP * 2);
  const size_t m = staging > epilogue ? staging : epilogue;
  return (m + 15) / 16 * 16;
}
size_t lds_bytes_for(const ConvGeom& g, int NT) {      // + the input BatchNorm's [scale | shift] behind the work area
  return lds_work_bytes(g, NT) + sizeof(float) * 2 * (size_t)g.Cin;
}

// Deepest admissible tile (TD in {4,2,1}) whose LDS image fits and that still leaves >= 512 blocks of work
// (2 per CU); if none has 512 blocks, the shallowest that fits.
int pick_td(int64_t Cin, int64_t Cout, int64_t Di, int64_t Hi, int64_t Wi, int stride) {
  const int NT = (int)((Cout + 15) / 16);
  int best = 0;
  // stride-2 TD=4 would need > 80 KiB of LDS; with <= 16 output channels TD=2 at twice the occupancy beats TD=4
  for (int td = (stride == 1 && NT > 1) ? 4 : 2; td >= 1; td >>= 1) {
    const ConvGeom g = make_geom(Cin, Cout, Di, Hi, Wi, stride, td);
    if (lds_bytes_for(g, NT) > kMaxLds) continue;
    best = td;
    if ((int64_t)g.tiles_d * g.tiles_h * g.tiles_w >= 512) return td;
  }
  return best;
}

int blocks_for(const ConvGeom& g) {
  const int64_t total = (int64_t)g.tiles_d * g.tiles_h * g.tiles_w;
#ifndef PF_CONV3D_CAP
#define PF_CONV3D_CAP 2048
#endif
  if (total <= PF_CONV3D_CAP) return (int)total;
  const int64_t per = (total + PF_CONV3D_CAP - 1) / PF_CONV3D_CAP;      // every block the same number of tiles
  return (int)((total + per - 1) / per);
}

template <int NT, int STRIDE, int TD, int MINW>
int launch(const float* x, const float* wp, float* y, const ConvGeom& g, int64_t N, double* partials,
           hipStream_t s) {
  const size_t lds_bytes = lds_bytes_for(g, NT);
  if (lds_bytes > kMaxLds) return PF_ERR_UNSUPPORTED;
  ConvGeom gk = g;
  gk.aff_off = (int)(lds_work_bytes(g, NT) / sizeof(float));
  if (lds_bytes > 64 * 1024) {           // opt in to more than the default 64 KiB of dynamic LDS (160 KiB per CU)
    static std::atomic<unsigned long long> done{0};   // per instantiation, one bit per device
    const int rc = pf_allow_big_lds(reinterpret_cast<const void*>(&conv3d_k3_kernel<NT, STRIDE, TD, MINW>),
                                    (int)kMaxLds, done);
    if (rc != PF_OK) return rc;
  }
  dim3 grid((unsigned)blocks_for(g), (unsigned)N);
  hipLaunchKernelGGL((conv3d_k3_kernel<NT, STRIDE, TD, MINW>), grid, dim3(256), lds_bytes, s, x, wp, y, gk, partials);
  return pf_launch_status();
}

template <int NT, int STRIDE, int TD>
int launch_w(const float* x, const float* wp, float* y, const ConvGeom& g, int64_t N, double* partials,
             hipStream_t s) {
  // measured (profiles/archive/r01/r01g_microbench_conv3d.log): the 64 -> 8 layer runs 151 us at TD 4 / 2 waves per
  // SIMD and 131 us at TD 2 / 4 waves per SIMD (124 VGPRs, no spills); the stride-2 tile is best left alone
  constexpr int kDefault = (NT == 1 && STRIDE == 1 && TD == 2) ? 4 : 2;
  return launch<NT, STRIDE, TD, kDefault>(x, wp, y, g, N, partials, s);
}

template <int NT, int STRIDE>
int launch_td(int td, const float* x, const float* wp, float* y, const ConvGeom& g, int64_t N, double* partials,
              hipStream_t s) {
  if constexpr (STRIDE == 1) {
    if (td == 4) return launch_w<NT, STRIDE, 4>(x, wp, y, g, N, partials, s);
  }
  if (td == 2) return launch_w<NT, STRIDE, 2>(x, wp, y, g, N, partials, s);
  return launch_w<NT, STRIDE, 1>(x, wp, y, g, N, partials, s);
}

}  // namespace

extern "C" {

int pf_conv3d_blocks(int64_t Cin, int64_t Cout, int64_t Di, int64_t Hi, int64_t Wi, int stride) {
  if (Cin <= 0 || Cout <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return 0;
  const int td = pick_td(Cin, Cout, Di, Hi, Wi, stride);
  if (td == 0) return 0;
  return blocks_for(make_geom(Cin, Cout, Di, Hi, Wi, stride, td));
}

int pf_conv3d_k3_f32(const float* x, const float* wp, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t Di,
                     int64_t Hi, int64_t Wi, int stride, const float* in_scale, const float* in_shift,
                     const pf_bn_job* in_bn, int samples_per_stat, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 4 && Cout >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && N <= 65535);
  PF_REQUIRE(stride == 1 || stride == 2);
  PF_REQUIRE(samples_per_stat >= 1 && (in_scale == nullptr) == (in_shift == nullptr));
  PF_REQUIRE(in_bn == nullptr || in_scale == nullptr);
  if (in_bn != nullptr && N > 0) {
    PF_REQUIRE(N % samples_per_stat == 0);
    const int rc = pf_bn_in_check(in_bn, (int)Cin, (int)(N / samples_per_stat));
    if (rc != PF_OK) return rc;
  }
  if ((Cin % 4) != 0 || Cout > 32) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(Cin * Di * Hi * Wi <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && wp && y);
  const int td = pick_td(Cin, Cout, Di, Hi, Wi, stride);
  if (td == 0) return PF_ERR_UNSUPPORTED;
  ConvGeom g = make_geom(Cin, Cout, Di, Hi, Wi, stride, td);
  g.aff_mode = in_bn ? 2 : (in_scale ? 1 : 0);
  g.sps = samples_per_stat;
  g.in_scale = in_scale;
  g.in_shift = in_shift;
  if (in_bn) g.in_bn = *in_bn;
  hipStream_t s = (hipStream_t)stream;
  const int NT = (int)((Cout + 15) / 16);
  if (stride == 1)
    return NT == 1 ? launch_td<1, 1>(td, x, wp, y, g, N, partials, s) : launch_td<2, 1>(td, x, wp, y, g, N, partials, s);
  return NT == 1 ? launch_td<1, 2>(td, x, wp, y, g, N, partials, s) : launch_td<2, 2>(td, x, wp, y, g, N, partials, s);
}

int pf_conv3d_k3_few_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t D,
                         int64_t H, int64_t W, void* stream) {
  PF_REQUIRE(N >= 0 && Cin >= 1 && Cout >= 1 && D >= 1 && H >= 1 && W >= 1 && N <= 65535);
  if (Cout > 4 || Cout * Cin * 27 * sizeof(float) > 48 * 1024) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE(H * W <= INT32_MAX);
  if (N == 0) return PF_OK;
  PF_REQUIRE(x && w && y);
  const size_t lds = sizeof(float) * (size_t)(Cout * Cin * 27);
  dim3 grid((unsigned)pf_cdiv(D * H * W, 256), (unsigned)N);
  hipStream_t s = (hipStream_t)stream;
  switch ((int)Cout) {
    case 1: hipLaunchKernelGGL(conv3d_k3_few_kernel<1>, grid, dim3(256), lds, s, x, w, y, (int)Cin, (int)D, (int)H, (int)W); break;
    case 2: hipLaunchKernelGGL(conv3d_k3_few_kernel<2>, grid, dim3(256), lds, s, x, w, y, (int)Cin, (int)D, (int)H, (int)W); break;
    case 3: hipLaunchKernelGGL(conv3d_k3_few_kernel<3>, grid, dim3(256), lds, s, x, w, y, (int)Cin, (int)D, (int)H, (int)W); break;
    default: hipLaunchKernelGGL(conv3d_k3_few_kernel<4>, grid, dim3(256), lds, s, x, w, y, (int)Cin, (int)D, (int)H, (int)W); break;
  }
  return pf_launch_status();
}

}  // extern "C"
