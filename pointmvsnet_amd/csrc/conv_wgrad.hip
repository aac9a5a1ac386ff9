// Row Z (training step, BASELINE config 4): WEIGHT GRADIENTS of every convolution of the step on the f32 matrix
// cores -- ImageConv's 2-D convolutions (reference networks.py:84-124), VolumeConv's 3-D convolutions and transposed
// convolutions (networks.py:127-167), and the 1x1 convolutions of EdgeConv / the flow MLP (networks.py:13-14,
// model.py:40-43) -- replacing the library kernels ATen's convolution_backward reaches (reference train.py:80; the
// MIOpen / CK weight-gradient solvers took 84 of the step's 117 ms, profiles/archive/r02/r02al_cfg4_last_steps_eager.md).
//
// One formulation for all of them.  With a COARSE-grid tensor Gr (N, Cg, Do, Ho, Wo) and a FINE-grid tensor X
// (N, Cx, Di, Hi, Wi):
//       dW[cg][cx][kd][kh][kw] = sum_{n, o} Gr[n, cg, o] * X[n, cx, o * stride + k - pad]        (zero outside X)
//   * convolution  y = conv(x, W (Cout, Cin, k..)):   Gr = dL/dy, X = x            -> dW in nn.ConvNd's layout;
//   * transposed convolution (stride 2, pad 1, output_padding 1)  y = convT(x, W (Cin, Cout, k..)):
//                                                      Gr = x,    X = dL/dy        -> dW in nn.ConvTransposeNd's layout
//     (y[co][2 i - 1 + k] += x[ci][i] * W[ci][co][k], so dW[ci][co][k] = sum_i x[ci][i] * dy[co][2 i - 1 + k]);
//   * 1x1 convolution over points: the same with one tap, on point-major rows (P, ld).
// As a GEMM: M = Cg rows, N = Cx * taps columns, reduction over (sample, position): a reduction 10^4..10^6 long into a
// tiny output.  Mapping:
//   * a 256-thread block owns a slice of the positions (`split`), a block of input channels (blockIdx.y) and up to
//     64 rows (blockIdx.z).  It walks its position tiles (TD x TH x 16) with ALL its accumulators resident in
//     registers: acc[row tile][column tile] of v_mfma_f32_16x16x4_f32, rows = 16 channels of Gr, columns = 16
//     (channel, tap) pairs, reduction step = 4 consecutive positions along W;
//   * per tile the block stages the Gr tile [cg][position] and the X patch [cx][(TD-1)s+KD][(TH-1)s+KH][15s+KW]
//     (zero outside the tensor, optional pending BatchNorm + ReLU of x applied on the way) into LDS; a lane's B operand
//     for column (cx, tap) is the patch read at a per-lane constant offset + the position, so no im2col exists anywhere.
//     Staging moves 16-byte pieces (dword-aligned global_load_dwordx4 -> two ds_write_b64) through a per-block table of
//     patch rows, and a tile whose patch lies inside the tensor takes a path without a single bounds test: the first
//     version tested and addressed every float on its own and spent 10 VALU + 6 SALU instructions per MFMA on it
//     (SQ counters, profiles/r04b_wgrad_sq_counters.md) -- the f32 matrix pipe shares its issue port with the VALU;
//   * reduction step u of a position row covers positions lk + 4u (lk = lane >> 4): with planes = 2 (mod 32) floats
//     the 32 lanes of a half-wave read 32 different LDS banks for A and for B (stride 1);
//   * column tiles are dealt round-robin to the four waves; every wave reads the A operands (Gr) of a position row
//     once and reuses them for all its column tiles;
//   * at the end each block writes its partial dW to a workspace (split, Cg, taps, Cx) -- channel fastest (round 6): the
//     16 lanes of a column tile hold 16 consecutive channels, so a store instruction writes four 64-byte runs where
//     (split, Cg, Cx, taps) made it 64 four-byte pieces 36-100 bytes apart; a second kernel adds the splits in split
//     order and puts the sums into nn.ConvNd's (Cg, Cx, taps) order: plain stores, fixed order -- the gradient is
//     bit-reproducible run to run (the library's split-K solvers use float atomics).
// Bound: fp32 MFMA (2 * taps * Cg * Cx flop per position against 4 * (Cg + Cx) bytes); the 8-channel layers fill half
// of the 16 MFMA rows.
#include <stdlib.h>

#include "pf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4 {      // 16 bytes at dword alignment: one global_load_dwordx4
  float v[4];
};

constexpr int kMaxNTW = 7;            // column tiles per wave (27 taps x 16 channels = 27 tiles over 4 waves)

struct WgGeom {
  int N, Cg, Cx;
  int Do, Ho, Wo, Di, Hi, Wi;
  int KD, KH, KW, T;
  int pd, ph, pw;
  int TD, TH, lgTH, lgR;              // tile rows R = TD * TH (a power of two)
  int ID, IH, IW, IWP, XPLANE, GPLANE;
  int QX, lgGS;                       // 16-byte pieces per patch row; log2 of the lanes that share a row (8 or 16)
  int CBP, lgCBP, TPT, TPC, CBLK, NTILES;
  int tiles_d, tiles_h, tiles_w;
  int total_tiles, per_split;
  int gs_floats, xs_floats, ntasks;
  // pending BatchNorm + ReLU of X: relu(x * scale[s, c] + shift[s, c]); s = n / x_sps (planar) or p / x_pps (rows)
  const float* x_scale;
  const float* x_shift;
  int x_sps;
  int64_t x_pps;
  // point-major rows (taps == 1): Gr (P, ldg), X (P, ldx)
  int point_major;
  int64_t P, ldg, ldx;
  int dbg;                            // PF_WGRAD_DBG (tools only): 1 = no staging, 2 = no MFMA loop, 4 = no partial store
};

// NTW: column tiles per wave = ceil(NTILES / 4).  Every wave runs NTW tiles -- one that does not exist reads column 0's
// operands and is never stored -- so the MFMA loop has no branch and the accumulators stay in the matrix pipe's
// registers: with a per-tile `if (tile exists)` the compiler moved 24 accumulator registers between the two register
// files on every position row (~16 VALU instructions per MFMA, profiles/r04b_wgrad_sq_counters.md).  The block waits
// for its fullest wave either way.
template <int MT, int STRIDE, int NTW>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ Gr, const float* __restrict__ X,
                                           float* __restrict__ part, const WgGeom& g, int bx, int by, int bz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* gs = lds;                                    // [MT*16][GPLANE]
  float* xs = lds + g.gs_floats;                      // [CBLK*CBP][XPLANE]
  int* tab = reinterpret_cast<int*>(xs + g.xs_floats);   // [ntasks][4]: global offset, LDS offset, dz, hy | c << 16

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int split = bx;
  const int cbtot = g.CBLK * g.CBP;
  const int ci0 = by * cbtot;
  const int cg0 = bz * (MT * 16);
  const int R = g.TD * g.TH;
  const int plane_i = g.Hi * g.Wi, vol_i = plane_i * g.Di;
  const int plane_o = g.Ho * g.Wo, vol_o = plane_o * g.Do;

  // per-lane B offsets of this wave's column tiles: column li of tile nt = (channel sub-block, tap group)
  int boff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int nt = wave + 4 * t;
    int off = 0;
    if (nt < g.NTILES) {
      const int sub = nt / g.TPC, tg = nt - sub * g.TPC;
      const int cil = li & (g.CBP - 1), tl = li >> g.lgCBP;
      int tap = tg * g.TPT + tl;
      tap = tap < g.T ? tap : g.T - 1;                 // (a padded column: computed, never stored)
      const int kd = tap / (g.KH * g.KW);
      const int r2 = tap - kd * (g.KH * g.KW);
      const int kh = r2 / g.KW, kw = r2 - kh * g.KW;
      off = (sub * g.CBP + cil) * g.XPLANE + (kd * g.IH + kh) * g.IWP + kw + lk * STRIDE;
    }
    boff[t] = off;
  }
  const int aoff = li * g.GPLANE + lk;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  if (!g.point_major) {
    // row tasks of the X patch: (channel, dz, hy) -> one row of IW floats; the same for every tile of the block
    const int rows_per_c = g.ID * g.IH;
    for (int task = tid; task < g.ntasks; task += 256) {
      const int c = task / rows_per_c;
      const int rr = task - c * rows_per_c;
      const int dz = rr / g.IH, hy = rr - dz * g.IH;
      tab[task * 4 + 0] = c * vol_i + dz * plane_i + hy * g.Wi;
      tab[task * 4 + 1] = c * g.XPLANE + (dz * g.IH + hy) * g.IWP;
      tab[task * 4 + 2] = dz;
      tab[task * 4 + 3] = hy | (c << 16);
    }
  }

  // Gr rows (planar mode): row (cg, position row) of the tile's window sits at a per-block constant offset, so the
  // offsets are tabulated once (after the X table) and a tile inside the tensor costs one table read + one add per load
  int* gtab = tab + 4 * g.ntasks;                            // [MT*16*R][2]: element offset in the window, LDS offset
  constexpr int kGT = 4 * MT;                                // <= 16 MT rows x 16 position rows x 4 pieces / 256 threads
  const int grows_blk = min(MT * 16, g.Cg - cg0);
  if (!g.point_major) {
    for (int row = tid; row < grows_blk * R; row += 256) {
      const int cgl = row >> g.lgR, r = row & (R - 1);
      const int d = r >> g.lgTH, h = r & (g.TH - 1);
      gtab[2 * row] = cgl * vol_o + d * plane_o + h * g.Wo;
      gtab[2 * row + 1] = cgl * g.GPLANE + r * 16;
    }
  }

  const int t_lo = split * g.per_split;
  const int t_hi = min(g.total_tiles, t_lo + g.per_split);
  for (int tile = t_lo; tile < t_hi; ++tile) {
    __syncthreads();                    // the previous tile's MFMA reads (and the table) are done
    // Staging issues its global loads in batches (kU independent loads per lane in flight) and stores to LDS afterwards:
    // one load -> one store per iteration made the kernel latency-bound (conv0_1's gradient: 525 us for 22 us of MFMA).
    constexpr int kU = MT == 4 ? 4 : 8, kG = 4, kUR = 4;
    const int grows = min(MT * 16, g.Cg - cg0);          // rows / channels that exist: the rest of the LDS tiles is
    const int creal = min(cbtot, g.Cx - ci0);            // never stored from (their products are not written)
    if (g.dbg & 1) {
    } else if (g.point_major) {
      // R * 16 consecutive points; thread = (point, 4 channels): coalesced 16-byte row pieces, transposed into LDS
      const int PT = R * 16;
      const int64_t p0 = (int64_t)tile * PT;
      const int qg = (grows + 3) >> 2;
      for (int e0 = tid; e0 < PT * qg; e0 += 256 * kUR) {
        f32x4 v[kUR];
#pragma unroll
        for (int u = 0; u < kUR; ++u) {
          const int e = e0 + 256 * u;
          const int pt = e / qg, c4 = (e - pt * qg) * 4;
          const int64_t p = p0 + pt;
          v[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          if (e < PT * qg && p < g.P) v[u] = *reinterpret_cast<const f32x4*>(Gr + p * g.ldg + cg0 + c4);
        }
#pragma unroll
        for (int u = 0; u < kUR; ++u) {
          const int e = e0 + 256 * u;
          const int pt = e / qg, c4 = (e - pt * qg) * 4;
          if (e < PT * qg) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gs[(c4 + j) * g.GPLANE + pt] = v[u][j];
          }
        }
      }
      const int qx = (creal + 3) >> 2;
      for (int e0 = tid; e0 < PT * qx; e0 += 256 * kUR) {
        f32x4 v[kUR];
#pragma unroll
        for (int u = 0; u < kUR; ++u) {
          const int e = e0 + 256 * u;
          const int pt = e / qx, c4 = (e - pt * qx) * 4;
          const int64_t p = p0 + pt;
          v[u] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
          if (e < PT * qx && p < g.P) {
            v[u] = *reinterpret_cast<const f32x4*>(X + p * g.ldx + ci0 + c4);
            if (g.x_scale != nullptr) {
              const int64_t so = (p / g.x_pps) * g.Cx + ci0 + c4;
              const f32x4 sc = *reinterpret_cast<const f32x4*>(g.x_scale + so);
              const f32x4 sh = *reinterpret_cast<const f32x4*>(g.x_shift + so);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[u][j] = fmaxf(fmaf(v[u][j], sc[j], sh[j]), 0.0f);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kUR; ++u) {
          const int e = e0 + 256 * u;
          const int pt = e / qx, c4 = (e - pt * qx) * 4;
          if (e < PT * qx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xs[(c4 + j) * g.XPLANE + pt] = v[u][j];
          }
        }
      }
    } else {
      int rest = tile;
      const int tw = rest % g.tiles_w;
      rest /= g.tiles_w;
      const int th = rest % g.tiles_h;
      rest /= g.tiles_h;
      const int td = rest % g.tiles_d;
      const int n = rest / g.tiles_d;
      const int od0 = td * g.TD, oh0 = th * g.TH, ow0 = tw * 16;
      const int id0 = od0 * STRIDE - g.pd, ih0 = oh0 * STRIDE - g.ph, iw0 = ow0 * STRIDE - g.pw;
      // Gr tile: 4 lanes per row of 16 positions along W, 16 bytes each
      const bool g_inside = od0 + g.TD <= g.Do && oh0 + g.TH <= g.Ho && ow0 + 16 <= g.Wo;
      if (g_inside) {
        const float* gbu = Gr + ((int64_t)n * g.Cg + cg0) * vol_o + ((int64_t)od0 * plane_o + oh0 * g.Wo + ow0);   // uniform
        const int q4 = (tid & 3) * 4, rg = tid >> 2;
        const int gtasks = grows_blk * R;
        constexpr int kGB = kGT < 4 ? kGT : 4;                 // loads in flight per lane
#pragma unroll
        for (int j0 = 0; j0 < kGT; j0 += kGB) {
          U4 v[kGB];
          int dst[kGB];
#pragma unroll
          for (int j = 0; j < kGB; ++j) {
            const int row = rg + 64 * (j0 + j);
            dst[j] = -1;
            if (row < gtasks) {
              const int2 e = *reinterpret_cast<const int2*>(gtab + 2 * row);
              v[j] = *reinterpret_cast<const U4*>(gbu + (e.x + q4));
              dst[j] = e.y + q4;
            }
          }
#pragma unroll
          for (int j = 0; j < kGB; ++j) {
            if (dst[j] >= 0) {
              *reinterpret_cast<float2*>(gs + dst[j]) = make_float2(v[j].v[0], v[j].v[1]);
              *reinterpret_cast<float2*>(gs + dst[j] + 2) = make_float2(v[j].v[2], v[j].v[3]);
            }
          }
        }
      } else {
        const int q4 = (tid & 3) * 4, rg = tid >> 2;
        const float* gb = Gr + ((int64_t)n * g.Cg + cg0) * vol_o + (int64_t)od0 * plane_o + oh0 * g.Wo + ow0 + q4;
        const int gtasks = grows * R;
        const bool inside = false;
        for (int t0 = rg; t0 < gtasks; t0 += 64 * kG) {
          U4 v[kG];
#pragma unroll
          for (int u = 0; u < kG; ++u) {
            const int task = t0 + 64 * u;
            const int cgl = task >> g.lgR, r = task & (R - 1);
            const int d = r >> g.lgTH, h = r & (g.TH - 1);
            const float* src = gb + (int64_t)cgl * vol_o + d * plane_o + h * g.Wo;
            v[u] = U4{{0.0f, 0.0f, 0.0f, 0.0f}};
            if (task < gtasks) {
              if (inside) {
                v[u] = *reinterpret_cast<const U4*>(src);
              } else if (od0 + d < g.Do && oh0 + h < g.Ho) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (ow0 + q4 + j < g.Wo) v[u].v[j] = src[j];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < kG; ++u) {
            const int task = t0 + 64 * u;
            if (task < gtasks) {
              float* dst = gs + (task >> g.lgR) * g.GPLANE + (task & (R - 1)) * 16 + q4;
              *reinterpret_cast<float2*>(dst) = make_float2(v[u].v[0], v[u].v[1]);
              *reinterpret_cast<float2*>(dst + 2) = make_float2(v[u].v[2], v[u].v[3]);
            }
          }
        }
      }
      // X patch: 8 (16) lanes per patch row, 16 bytes each; rows come from the block's table
      {
        const int gsz = 1 << g.lgGS;
        const int q = tid & (gsz - 1), rg = tid >> g.lgGS, rp = 256 >> g.lgGS;
        const int q4 = q * 4;
        const float* xbu = X + ((int64_t)n * g.Cx + ci0) * vol_i + ((int64_t)id0 * plane_i + ih0 * g.Wi + iw0);   // uniform
        const float* xb = xbu + q4;
        const int stat = n / g.x_sps;
        const int xrows = creal * g.ID * g.IH;
        const bool lane_on = q < g.QX;
        const bool inside = id0 >= 0 && id0 + g.ID <= g.Di && ih0 >= 0 && ih0 + g.IH <= g.Hi && iw0 >= 0 &&
                            iw0 + g.IWP <= g.Wi;
        const bool affine = g.x_scale != nullptr;
        const float* scp = affine ? g.x_scale + (int64_t)stat * g.Cx + ci0 : nullptr;
        const float* shp = affine ? g.x_shift + (int64_t)stat * g.Cx + ci0 : nullptr;
        if (inside && !affine) {
          for (int t0 = rg; t0 < xrows; t0 += rp * kU) {
            U4 v[kU];
            int dst[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              const int row = t0 + rp * u;
              dst[u] = -1;
              if (row < xrows && lane_on) {
                const int2 e = *reinterpret_cast<const int2*>(tab + row * 4);
                v[u] = *reinterpret_cast<const U4*>(xbu + (e.x + q4));
                dst[u] = e.y + q4;
              }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              if (dst[u] >= 0) {
                *reinterpret_cast<float2*>(xs + dst[u]) = make_float2(v[u].v[0], v[u].v[1]);
                *reinterpret_cast<float2*>(xs + dst[u] + 2) = make_float2(v[u].v[2], v[u].v[3]);
              }
            }
          }
        } else {
          for (int t0 = rg; t0 < xrows; t0 += rp * kU) {
            U4 v[kU];
            int dst[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              const int row = t0 + rp * u;
              dst[u] = -1;
              v[u] = U4{{0.0f, 0.0f, 0.0f, 0.0f}};
              if (row < xrows && lane_on) {
                const int4 e = *reinterpret_cast<const int4*>(tab + row * 4);
                const int c = e.w >> 16, hy = e.w & 0xffff;
                const int id = id0 + e.z, ih = ih0 + hy, iw = iw0 + q4;
                dst[u] = e.y + q4;
                if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi) {
                  const float* src = xb + e.x;
                  const bool whole = iw >= 0 && iw + 4 <= g.Wi;
                  if (whole) {
                    v[u] = *reinterpret_cast<const U4*>(src);
                  } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                      if (iw + j >= 0 && iw + j < g.Wi) v[u].v[j] = src[j];
                  }
                  if (affine) {
                    const float sc = scp[c], sh = shp[c];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                      if (whole || (iw + j >= 0 && iw + j < g.Wi)) v[u].v[j] = fmaxf(fmaf(v[u].v[j], sc, sh), 0.0f);
                  }
                }
              }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              if (dst[u] >= 0) {
                *reinterpret_cast<float2*>(xs + dst[u]) = make_float2(v[u].v[0], v[u].v[1]);
                *reinterpret_cast<float2*>(xs + dst[u] + 2) = make_float2(v[u].v[2], v[u].v[3]);
              }
            }
          }
        }
      }
    }
    __syncthreads();

    for (int r = 0; r < ((g.dbg & 2) ? 0 : R); ++r) {
      const int d = r >> g.lgTH, h = r & (g.TH - 1);
      float a[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float* ap = gs + m * 16 * g.GPLANE + aoff + r * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) a[m][u] = ap[4 * u];
      }
      const int rowoff = ((d * STRIDE) * g.IH + h * STRIDE) * g.IWP;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const float* bp = xs + boff[t] + rowoff;
        float b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = bp[4 * u * STRIDE];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][u], b[u], acc[m][t], 0, 0, 0);
      }
    }
  }

  // partial dW of this block: C/D layout col = lane & 15 (column of the tile), row = (lane >> 4) * 4 + r (channel of Gr)
  float* pb = part + (int64_t)split * g.Cg * g.Cx * g.T;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int nt = wave + 4 * t;
    if (nt >= g.NTILES) continue;
    const int sub = nt / g.TPC, tg = nt - sub * g.TPC;
    const int cil = li & (g.CBP - 1), tl = li >> g.lgCBP;
    const int tap = tg * g.TPT + tl;
    const int ci = ci0 + sub * g.CBP + cil;
    if (tap >= g.T || ci >= g.Cx || (g.dbg & 4)) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cg = cg0 + m * 16 + lk * 4 + r;
        if (cg < g.Cg) pb[((int64_t)cg * g.T + tap) * g.Cx + ci] = acc[m][t][r];
      }
    }
  }
}

template <int MT, int STRIDE, int NTW>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ Gr, const float* __restrict__ X,
                                                    float* __restrict__ part, WgGeom g) {
  wgrad_body<MT, STRIDE, NTW>(Gr, X, part, g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several layers' weight gradients of ONE instantiation in one launch (round 6; pf_conv_wgrad_batch_f32).  VolumeConv's
// layers below 24x32x40 are 96-384 blocks of ~20 us each -- prologue, two staging round trips, 112 MFMAs per wave, the
// partial store -- one after the other in the backward chain, although nothing in the step waits for a weight gradient: a
// node now queues its layers and launches them together when its data-gradient chain is done, the small ones riding in
// the grid of the large one of the same instantiation.  Block b belongs to entry e with first_block[e] <= b.
constexpr int kWgBatch = 8;
struct WgBatchArgs {
  const float* Gr[kWgBatch];
  const float* X[kWgBatch];
  float* part[kWgBatch];
  WgGeom g[kWgBatch];
  int first_block[kWgBatch + 1];
  int gx[kWgBatch], gy[kWgBatch];
  int n;
};
static_assert(sizeof(WgBatchArgs) <= 3584, "kernel arguments");

template <int MT, int STRIDE, int NTW>
__global__ __launch_bounds__(256) void wgrad_batch_kernel(WgBatchArgs b) {
  int e = 0;
  while (e + 1 < b.n && (int)blockIdx.x >= b.first_block[e + 1]) ++e;          // block-uniform
  const int lb = (int)blockIdx.x - b.first_block[e];
  const int bx = lb % b.gx[e], r = lb / b.gx[e];
  wgrad_body<MT, STRIDE, NTW>(b.Gr[e], b.X[e], b.part[e], b.g[e], bx, r % b.gy[e], r / b.gy[e]);
}

// dw[e] = sum over the splits, in a fixed order: slice sl of 16 adds splits sl, sl + 16, ... and the 16 slice sums are added
// in slice order.  Block = 64 elements x 16 slices (round 6; it was 16 x 16: a wave then read four 64-byte pieces per load
// instruction -- the towers' ~95 MB of partials at ~2 TB/s; now a wave reads one 256-byte run.  Same sums, same order).
// The partials are (rows A, taps T, columns B), B fastest (what wgrad_kernel stores); element e' = (a * T + t) * B + b lands
// at (a * B + b) * T + t -- or, for a SWAPPED-OPERAND launch (rows = the layer's input channels, every tap axis reversed),
// at (b * A + a) * T + T - 1 - t: nn.ConvNd's (Cout, Cin, taps) order either way.
constexpr int kRedEl = 64, kRedThreads = 16 * kRedEl;
__device__ __forceinline__ void reduce_block(const float* __restrict__ part, float* __restrict__ dw, int64_t elems,
                                             int splits, int accumulate, int64_t block, float (*red)[kRedEl + 1],
                                             int rows, int taps, int swapped) {
  const int el = threadIdx.x & (kRedEl - 1), sl = threadIdx.x / kRedEl;
  const int64_t e = block * kRedEl + el;
  float s = 0.0f;
  if (e < elems) {
    int k = sl;
    for (; k + 48 < splits; k += 64) {           // four independent loads in flight
      const float a = part[(int64_t)k * elems + e], b = part[(int64_t)(k + 16) * elems + e];
      const float c = part[(int64_t)(k + 32) * elems + e], d = part[(int64_t)(k + 48) * elems + e];
      s += a;
      s += b;
      s += c;
      s += d;
    }
    for (; k < splits; k += 16) s += part[(int64_t)k * elems + e];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < elems) {
    const int64_t cols = elems / ((int64_t)rows * taps);
    const int64_t b = e % cols, at = e / cols;
    const int64_t a = at / taps, tp = at - a * taps;
    const int64_t eo = swapped ? (b * rows + a) * taps + (taps - 1 - tp) : (a * cols + b) * taps + tp;
    float t = accumulate ? dw[eo] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][el];
    dw[eo] = t;
  }
}

__global__ __launch_bounds__(kRedThreads) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ dw, int64_t elems, int splits,
                                                                   int accumulate, int rows, int taps) {
  __shared__ float red[16][kRedEl + 1];
  reduce_block(part, dw, elems, splits, accumulate, blockIdx.x, red, rows, taps, 0);
}

// Up to kRedBatch layers' reductions in one launch (the training step queues a node's weight gradients and reduces
// them together: 45 launches of ~5 us each in a dependency chain become 7).
constexpr int kRedBatch = 16;
struct RedBatch {
  const float* part[kRedBatch];
  float* dw[kRedBatch];
  int64_t elems[kRedBatch];
  int splits[kRedBatch];
  int rows[kRedBatch];
  int taps[kRedBatch];
  int swapped[kRedBatch];
  int first_block[kRedBatch + 1];
  int n, accumulate;
};

__global__ __launch_bounds__(kRedThreads) void wgrad_reduce_batch_kernel(RedBatch b) {
  __shared__ float red[16][kRedEl + 1];
  int d = 0;
  while (d + 1 < b.n && (int)blockIdx.x >= b.first_block[d + 1]) ++d;      // block-uniform
  reduce_block(b.part[d], b.dw[d], b.elems[d], b.splits[d], b.accumulate, (int64_t)blockIdx.x - b.first_block[d], red,
               b.rows[d], b.taps[d], b.swapped[d]);
}

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

constexpr int kCUs = 256;                          // MI355X (gfx950): 8 XCDs x 32 CUs
constexpr size_t kLdsPerCU = 160 * 1024;
constexpr size_t kLdsSoft = 64 * 1024;
constexpr size_t kLdsHard = 96 * 1024;
constexpr int64_t kWorkspaceCap = 32ll << 20;     // bytes of split partials a launch may use

struct WgPlan {
  WgGeom g;
  int MT, cblocks, mblocks, splits;
  size_t lds_bytes;
  bool ok;
};

template <int MT, int STRIDE>
const void* kernel_ptr(int ntw) {
  switch (ntw) {
    case 1: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 1>);
    case 2: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 2>);
    case 3: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 3>);
    case 4: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 4>);
    case 5: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 5>);
    case 6: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 6>);
    case 7: return reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, 7>);
    default: return nullptr;
  }
}

// Blocks of this instantiation one CU holds at `lds_bytes` of dynamic LDS: asked from the runtime once per
// (instantiation, LDS KB), with the register / LDS arithmetic of gfx950 as the fallback (no device, e.g. a build host).
int resident_blocks(int MT, int stride, int ntw, size_t lds_bytes) {
  const int kb = (int)((lds_bytes + 1023) / 1024);
  const int mi = MT == 1 ? 0 : (MT == 2 ? 1 : 2);
  static std::atomic<int> cache[3][2][8][161];                   // 0 = not asked yet
  if (ntw < 1 || ntw > 7 || kb > 160) return 1;
  std::atomic<int>& slot = cache[mi][stride - 1][ntw][kb];
  int occ = slot.load(std::memory_order_relaxed);
  if (occ > 0) return occ;
  const void* fn = stride == 1 ? (MT == 1 ? kernel_ptr<1, 1>(ntw) : (MT == 2 ? kernel_ptr<2, 1>(ntw) : kernel_ptr<4, 1>(ntw)))
                               : (MT == 1 ? kernel_ptr<1, 2>(ntw) : (MT == 2 ? kernel_ptr<2, 2>(ntw) : kernel_ptr<4, 2>(ntw)));
  int n = 0;
  if (fn == nullptr || hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds_bytes) != hipSuccess || n < 1) {
    (void)hipGetLastError();
    const int regs = 116 + (9 * MT * ntw) / 2;                   // VGPRs of the instantiations, within 10 %
    n = 512 / ((regs + 7) & ~7);
    const int by_lds = (int)(kLdsPerCU / ((size_t)kb * 1024));
    if (n > by_lds) n = by_lds;
    if (n < 1) n = 1;
  }
  if (n > 8) n = 8;
  slot.store(n, std::memory_order_relaxed);
  return n;
}

// The plan at MT row tiles per block (16 MT rows of Gr) and tile candidates from `first_tile` on.
WgPlan make_plan_mt(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi,
                    int64_t Wi, int KD, int KH, int KW, int stride, int pd, int ph, int pw, bool rows, int64_t P, int MT,
                    int first_tile) {
  WgPlan p;
  WgGeom& g = p.g;
  p.ok = false;
  g.N = (int)N;
  g.Cg = (int)Cg;
  g.Cx = (int)Cx;
  g.Do = (int)Do;
  g.Ho = (int)Ho;
  g.Wo = (int)Wo;
  g.Di = (int)Di;
  g.Hi = (int)Hi;
  g.Wi = (int)Wi;
  g.KD = KD;
  g.KH = KH;
  g.KW = KW;
  g.T = KD * KH * KW;
  g.pd = pd;
  g.ph = ph;
  g.pw = pw;
  g.x_scale = g.x_shift = nullptr;
  g.x_sps = 1;
  g.x_pps = 1;
  g.point_major = rows ? 1 : 0;
  static const int dbg = []() {
    const char* e = getenv("PF_WGRAD_DBG");
    return e == nullptr ? 0 : atoi(e);
  }();
  g.dbg = dbg;
  g.P = P;
  g.ldg = g.ldx = 0;
  g.CBP = Cx >= 16 ? 16 : (Cx > 4 ? 8 : 4);
  g.lgCBP = ilog2(g.CBP);
  g.TPT = 16 / g.CBP;
  g.TPC = (g.T + g.TPT - 1) / g.TPT;
  p.MT = MT;
  p.mblocks = (int)((Cg + 16 * MT - 1) / (16 * MT));
  const int cx_pad = (int)((Cx + g.CBP - 1) / g.CBP);          // channel sub-blocks in all
  // tile candidates: 3-D volumes 4 x 4 x 16 where it still leaves two blocks per CU (round 6: the patch halo costs
  // 1.5 x 1.5 x 1.25 instead of 2 x 1.5 x 1.25 staged floats per position -- conv0_1's gradient 201 -> 188 us, conv1_1's
  // 35 -> 30, profiles/r06j_wgrad_plans.md), then 2 x 4 x 16, 1 x 8 x 16, 1 x 4 x 16; 2-D layers try 1 x 16 x 16 first
  // when it fits the soft LDS budget: twice the matrix work between two barriers of a latency-bound loop
  const int cand[5][2] = {{1, 16}, {2, 4}, {1, 8}, {1, 4}, {4, 4}};
  static const bool big_tile = []() {
    const char* e = getenv("PF_WGRAD_BIG_TILE");
    return e == nullptr || e[0] != '0';
  }();
  static const bool big_rows = []() {
    const char* e = getenv("PF_WGRAD_BIG_ROWS");
    return e == nullptr || e[0] != '0';
  }();
  int seq[5], nseq = 0;
  if (rows) {
    if (big_rows) seq[nseq++] = 2;                             // 128 points per tile (E1, mlp3: +0.3 % on the step)
    seq[nseq++] = 3;
  } else if (first_tile == 2) {
    seq[nseq++] = 3;
  } else if (Do > 1) {
    if (Do >= 4 && Ho >= 4 && first_tile != 1) seq[nseq++] = 4;
    seq[nseq++] = 1;
    seq[nseq++] = 2;
    seq[nseq++] = 3;
  } else {
    if (big_tile && Ho >= 16) seq[nseq++] = 0;
    seq[nseq++] = 2;
    seq[nseq++] = 3;
  }
  for (int si = 0; si < nseq; ++si) {
    const int ci = seq[si];
    const size_t soft = ci == 4 ? 78 * 1024 : kLdsSoft;       // (4 x 4 x 16: 73-75 KB, still two blocks per CU)
    g.TD = cand[ci][0];
    g.TH = cand[ci][1];
    g.lgTH = ilog2(g.TH);
    g.lgR = ilog2(g.TD * g.TH);
    g.ID = (g.TD - 1) * stride + KD;
    g.IH = (g.TH - 1) * stride + KH;
    g.IW = 15 * stride + KW;
    g.IWP = rows ? g.IW : (g.IW + 3) & ~3;                     // planar: rows staged in 16-byte pieces
    g.QX = g.IWP / 4;
    g.lgGS = g.QX <= 8 ? 3 : 4;
    int xraw = g.ID * g.IH * g.IWP;
    g.XPLANE = ((xraw + 29) / 32) * 32 + 2;                    // = 2 (mod 32): lanes (channel li, step lk) of a half-wave
    g.GPLANE = g.TD * g.TH * 16 + 2;                           // read 2 li + lk = 32 different banks
    int cblk = kMaxNTW * 4 / g.TPC;
    if (cblk < 1) cblk = 1;                                    // (more than 28 tap groups: PF_ERR_UNSUPPORTED below)
    if (cblk > cx_pad) cblk = cx_pad;
    const int cblk_max = cblk;
    for (; cblk >= 1; --cblk) {
      g.CBLK = cblk;
      g.NTILES = cblk * g.TPC;
      g.gs_floats = p.MT * 16 * g.GPLANE;
      g.gs_floats = (g.gs_floats + 3) & ~3;
      g.xs_floats = cblk * g.CBP * g.XPLANE;
      g.xs_floats = (g.xs_floats + 3) & ~3;
      g.ntasks = rows ? 0 : cblk * g.CBP * g.ID * g.IH;
      p.lds_bytes = sizeof(float) * (size_t)(g.gs_floats + g.xs_floats) + sizeof(int) * 4 * (size_t)g.ntasks +
                    (rows ? 0 : sizeof(int) * 2 * (size_t)(p.MT * 16 * g.TD * g.TH));
      if (p.lds_bytes <= soft) break;
    }
    if ((ci == 0 || (rows && ci == 2)) && g.CBLK != cblk_max) continue;   // the big tile only where it costs no channel block
    if (g.CBLK >= 1 && p.lds_bytes <= soft && g.NTILES <= kMaxNTW * 4) {
      p.ok = true;
      break;
    }
    if (ci == 3 && g.NTILES <= kMaxNTW * 4 && p.lds_bytes <= kLdsHard) {   // last resort: the big-LDS opt-in
      p.ok = true;
      break;
    }
  }
  if (!p.ok) return p;
  p.cblocks = (cx_pad + g.CBLK - 1) / g.CBLK;
  if (rows) {
    g.tiles_d = g.tiles_h = 1;
    g.tiles_w = (int)((P + g.TD * g.TH * 16 - 1) / (g.TD * g.TH * 16));
    g.total_tiles = g.tiles_w;
  } else {
    g.tiles_d = (g.Do + g.TD - 1) / g.TD;
    g.tiles_h = (g.Ho + g.TH - 1) / g.TH;
    g.tiles_w = (g.Wo + 15) / 16;
    g.total_tiles = g.N * g.tiles_d * g.tiles_h * g.tiles_w;
  }
  const int64_t elems = Cg * Cx * g.T;
  // ONE round of resident blocks: every block walks the same number of tiles, so a second, partly filled round costs a
  // whole block time (the first version asked for ~1024 blocks: 960 blocks on 768 slots ran 2 rounds for 1.25 of work)
  const int occ = resident_blocks(p.MT, stride, (g.NTILES + 3) / 4, p.lds_bytes);
  int64_t splits = (int64_t)kCUs * occ / ((int64_t)p.cblocks * p.mblocks);
  if (splits < 1) splits = 1;
  const int64_t cap = kWorkspaceCap / (4 * elems);
  if (splits > cap) splits = cap < 1 ? 1 : cap;
  if (splits > g.total_tiles) splits = g.total_tiles;
  if (splits < 1) splits = 1;
  g.per_split = (int)((g.total_tiles + splits - 1) / splits);
  if (g.per_split < 1) g.per_split = 1;
  p.splits = (int)((g.total_tiles + g.per_split - 1) / g.per_split);
  if (p.splits < 1) p.splits = 1;
  return p;
}

// Row tiles per block: as many as the rows need (1, 2 or 4) -- unless that leaves the chip mostly empty (VolumeConv's
// 480- and 3 840-voxel layers: 6 tiles x 4 channel blocks = 24 blocks of 27 x 4 accumulator tiles each took 40-75 us):
// then fewer rows per block and the smallest tile, i.e. more blocks with less work each.
WgPlan make_plan(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi,
                 int64_t Wi, int KD, int KH, int KW, int stride, int pd, int ph, int pw, bool rows, int64_t P) {
  // Rows of Gr per block.  Point-major rows (one tap): as many as there are, up to 64 -- the operands are re-read per
  // row tile.  Planar layers: 16 (32 for the 64-row stride-1 layers): a block's partial dW is rows x channels x taps
  // floats per position slice, and with ~480 resident blocks the slices of the 25-tap and 32-row layers wrote and re-read
  // 18-25 MB -- 10-25 us of their 40-67 us (PF_WGRAD_DBG=4); fewer rows per block = more row blocks = fewer slices
  // (tower 16->32 / 32->64 5x5: 67 -> 55 us, 32->32: 41 -> 36, conv2_1: 36 -> 26; 64->64 3x3 is best at 32 rows: 40).
  int MT = Cg <= 16 ? 1 : (Cg <= 32 ? 2 : 4);
  if (!rows) MT = (Cg >= 64 && stride == 1) ? 2 : 1;
  WgPlan p = make_plan_mt(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, pd, ph, pw, rows, P, MT, -1);
  if (!p.ok || rows) return p;
  const auto blocks = [](const WgPlan& q) { return (int64_t)q.splits * q.cblocks * q.mblocks; };
  if (blocks(p) < kCUs / 2 && p.g.TD == 4) {                   // (a small volume: the 2 x 4 x 16 tile first)
    const WgPlan q = make_plan_mt(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, pd, ph, pw, rows, P, MT, 1);
    if (q.ok && blocks(q) > blocks(p)) p = q;
  }
  if (blocks(p) < kCUs / 2) {
    const WgPlan q = make_plan_mt(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, pd, ph, pw, rows, P, MT, 2);
    if (q.ok && blocks(q) > blocks(p)) p = q;
  }
  while (MT > 1 && blocks(p) < kCUs) {
    MT /= 2;
    const WgPlan q = make_plan_mt(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, pd, ph, pw, rows, P, MT,
                                  p.g.TD == 1 && p.g.TH == 4 ? 2 : -1);
    if (!q.ok) break;
    p = q;
  }
  return p;
}

template <int MT, int STRIDE, int NTW>
int launch_wgrad(const float* Gr, const float* X, float* part, const WgPlan& p, hipStream_t s) {
  if (p.lds_bytes > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};
    const int rc = pf_allow_big_lds(reinterpret_cast<const void*>(&wgrad_kernel<MT, STRIDE, NTW>), (int)kLdsHard, done);
    if (rc != PF_OK) return rc;
  }
  dim3 grid((unsigned)p.splits, (unsigned)p.cblocks, (unsigned)p.mblocks);
  hipLaunchKernelGGL((wgrad_kernel<MT, STRIDE, NTW>), grid, dim3(256), p.lds_bytes, s, Gr, X, part, p.g);
  return pf_launch_status();
}

template <int MT, int STRIDE>
int launch_wgrad_ntw(const float* Gr, const float* X, float* part, const WgPlan& p, hipStream_t s) {
  switch ((p.g.NTILES + 3) / 4) {
    case 1: return launch_wgrad<MT, STRIDE, 1>(Gr, X, part, p, s);
    case 2: return launch_wgrad<MT, STRIDE, 2>(Gr, X, part, p, s);
    case 3: return launch_wgrad<MT, STRIDE, 3>(Gr, X, part, p, s);
    case 4: return launch_wgrad<MT, STRIDE, 4>(Gr, X, part, p, s);
    case 5: return launch_wgrad<MT, STRIDE, 5>(Gr, X, part, p, s);
    case 6: return launch_wgrad<MT, STRIDE, 6>(Gr, X, part, p, s);
    case 7: return launch_wgrad<MT, STRIDE, 7>(Gr, X, part, p, s);
    default: return PF_ERR_UNSUPPORTED;
  }
}

int run_plan(const float* Gr, const float* X, float* dw, const WgPlan& p, int stride, void* workspace,
             int64_t workspace_bytes, int accumulate, hipStream_t s) {
  const int64_t elems = (int64_t)p.g.Cg * p.g.Cx * p.g.T;
  PF_REQUIRE(workspace != nullptr && workspace_bytes >= 4 * elems * p.splits);
  float* part = reinterpret_cast<float*>(workspace);
  int rc;
  if (stride == 1) {
    rc = p.MT == 1 ? launch_wgrad_ntw<1, 1>(Gr, X, part, p, s)
                   : (p.MT == 2 ? launch_wgrad_ntw<2, 1>(Gr, X, part, p, s) : launch_wgrad_ntw<4, 1>(Gr, X, part, p, s));
  } else {
    rc = p.MT == 1 ? launch_wgrad_ntw<1, 2>(Gr, X, part, p, s)
                   : (p.MT == 2 ? launch_wgrad_ntw<2, 2>(Gr, X, part, p, s) : launch_wgrad_ntw<4, 2>(Gr, X, part, p, s));
  }
  if (rc != PF_OK || dw == nullptr) return rc;          // dw == NULL: the partials only (pf_wgrad_reduce_batch_f32 later)
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)pf_cdiv(elems, kRedEl)), dim3(kRedThreads), 0, s, part, dw,
                     elems, p.splits, accumulate, p.g.Cg, p.g.T);
  return pf_launch_status();
}

template <int MT, int STRIDE, int NTW>
int launch_wgrad_batch(const WgBatchArgs& b, int blocks, size_t lds_bytes, hipStream_t s) {
  if (lds_bytes > kLdsSoft) {
    static std::atomic<unsigned long long> done{0};
    const int rc = pf_allow_big_lds(reinterpret_cast<const void*>(&wgrad_batch_kernel<MT, STRIDE, NTW>), (int)kLdsHard,
                                    done);
    if (rc != PF_OK) return rc;
  }
  hipLaunchKernelGGL((wgrad_batch_kernel<MT, STRIDE, NTW>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, b);
  return pf_launch_status();
}

template <int MT, int STRIDE>
int launch_wgrad_batch_ntw(const WgBatchArgs& b, int ntw, int blocks, size_t lds_bytes, hipStream_t s) {
  switch (ntw) {
    case 1: return launch_wgrad_batch<MT, STRIDE, 1>(b, blocks, lds_bytes, s);
    case 2: return launch_wgrad_batch<MT, STRIDE, 2>(b, blocks, lds_bytes, s);
    case 3: return launch_wgrad_batch<MT, STRIDE, 3>(b, blocks, lds_bytes, s);
    case 4: return launch_wgrad_batch<MT, STRIDE, 4>(b, blocks, lds_bytes, s);
    case 5: return launch_wgrad_batch<MT, STRIDE, 5>(b, blocks, lds_bytes, s);
    case 6: return launch_wgrad_batch<MT, STRIDE, 6>(b, blocks, lds_bytes, s);
    case 7: return launch_wgrad_batch<MT, STRIDE, 7>(b, blocks, lds_bytes, s);
    default: return PF_ERR_UNSUPPORTED;
  }
}

int launch_wgrad_batch_any(const WgBatchArgs& b, int MT, int stride, int ntw, int blocks, size_t lds_bytes, hipStream_t s) {
  if (stride == 1)
    return MT == 1 ? launch_wgrad_batch_ntw<1, 1>(b, ntw, blocks, lds_bytes, s)
                   : (MT == 2 ? launch_wgrad_batch_ntw<2, 1>(b, ntw, blocks, lds_bytes, s)
                              : launch_wgrad_batch_ntw<4, 1>(b, ntw, blocks, lds_bytes, s));
  return MT == 1 ? launch_wgrad_batch_ntw<1, 2>(b, ntw, blocks, lds_bytes, s)
                 : (MT == 2 ? launch_wgrad_batch_ntw<2, 2>(b, ntw, blocks, lds_bytes, s)
                            : launch_wgrad_batch_ntw<4, 2>(b, ntw, blocks, lds_bytes, s));
}

bool conv_args_ok(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi,
                  int64_t Wi, int KD, int KH, int KW, int stride) {
  return N >= 1 && Cg >= 1 && Cx >= 1 && Do >= 1 && Ho >= 1 && Wo >= 1 && Di >= 1 && Hi >= 1 && Wi >= 1 && KD >= 1 &&
         KH >= 1 && KW >= 1 && KD <= 7 && KH <= 7 && KW <= 7 && (stride == 1 || stride == 2) &&
         Cx * Di * Hi * Wi <= INT32_MAX && Cg * Do * Ho * Wo <= INT32_MAX && N <= 65535;
}

}  // namespace

extern "C" {

int64_t pf_conv_wgrad_workspace(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di,
                                int64_t Hi, int64_t Wi, int KD, int KH, int KW, int stride) {
  if (!conv_args_ok(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride)) return -1;
  const WgPlan p = make_plan(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, 0, 0, 0, false, 0);
  if (!p.ok) return -1;
  return 4 * Cg * Cx * (int64_t)p.g.T * p.splits;
}

namespace {
// The fields of a launch plan a test can pin (pf_conv_wgrad_plan / pf_rows_wgrad_plan): the plan is a pure function of
// the shape, so "the plan tested is the plan the step uses" is "the same shape gives the same twelve numbers".
void export_plan(const WgPlan& p, int stride, int* out) {
  out[0] = p.MT;                       // 16-row tiles of Gr per block
  out[1] = p.g.TD;                     // position tile: TD x TH x 16 (rows mode: TD * TH * 16 points)
  out[2] = p.g.TH;
  out[3] = p.g.CBLK;                   // input-channel sub-blocks per block
  out[4] = p.g.CBP;                    // channels per sub-block
  out[5] = (p.g.NTILES + 3) / 4;       // NTW: accumulator tiles per wave (template parameter)
  out[6] = p.splits;                   // position slices (partials added in this order)
  out[7] = p.cblocks;
  out[8] = p.mblocks;
  out[9] = (int)p.lds_bytes;
  out[10] = p.g.per_split;             // position tiles per block
  out[11] = stride;
}
}  // namespace

int pf_conv_wgrad_plan(int64_t N, int64_t Cg, int64_t Cx, int64_t Do, int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi,
                       int64_t Wi, int KD, int KH, int KW, int stride, int* plan12) {
  PF_REQUIRE(plan12 != nullptr);
  if (!conv_args_ok(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride)) return PF_ERR_UNSUPPORTED;
  const WgPlan p = make_plan(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, 0, 0, 0, false, 0);
  if (!p.ok) return PF_ERR_UNSUPPORTED;
  export_plan(p, stride, plan12);
  return PF_OK;
}

int pf_rows_wgrad_plan(int64_t P, int Cg, int Cx, int* plan12) {
  PF_REQUIRE(plan12 != nullptr);
  if (P < 1 || Cg < 1 || Cx < 1 || (Cg & 3) || (Cx & 3)) return PF_ERR_UNSUPPORTED;
  const WgPlan p = make_plan(1, Cg, Cx, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, true, P);
  if (!p.ok) return PF_ERR_UNSUPPORTED;
  export_plan(p, 1, plan12);
  return PF_OK;
}

int pf_conv_wgrad_f32(const float* gr, const float* x, float* dw, int64_t N, int64_t Cg, int64_t Cx, int64_t Do,
                      int64_t Ho, int64_t Wo, int64_t Di, int64_t Hi, int64_t Wi, int KD, int KH, int KW, int stride,
                      int pd, int ph, int pw, const float* x_scale, const float* x_shift, int x_samples_per_stat,
                      void* workspace, int64_t workspace_bytes, int accumulate, void* stream) {
  PF_REQUIRE(conv_args_ok(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride));
  PF_REQUIRE(pd >= 0 && ph >= 0 && pw >= 0 && x_samples_per_stat >= 1 && (x_scale == nullptr) == (x_shift == nullptr));
  PF_REQUIRE(gr && x);
  WgPlan p = make_plan(N, Cg, Cx, Do, Ho, Wo, Di, Hi, Wi, KD, KH, KW, stride, pd, ph, pw, false, 0);
  if (!p.ok) return PF_ERR_UNSUPPORTED;
  p.g.x_scale = x_scale;
  p.g.x_shift = x_shift;
  p.g.x_sps = x_samples_per_stat;
  return run_plan(gr, x, dw, p, stride, workspace, workspace_bytes, accumulate, (hipStream_t)stream);
}

int pf_conv_wgrad_batch_f32(const pf_wgrad_item* items, int n, void* stream) {
  PF_REQUIRE(n >= 0 && (n == 0 || items != nullptr) && n <= 64);
  WgPlan plans[64];
  bool done[64];
  for (int i = 0; i < n; ++i) {
    const pf_wgrad_item& it = items[i];
    PF_REQUIRE((it.x_scale == nullptr) == (it.x_shift == nullptr) && it.gr && it.x && it.workspace);
    if (it.rows_P > 0) {                                       // point-major rows: what pf_rows_wgrad_f32 takes
      PF_REQUIRE(it.Cg >= 1 && it.Cx >= 1 && it.ldg >= it.Cg && it.ldx >= it.Cx && it.x_rows_per_stat >= 1 && it.stride == 1);
      if ((it.Cg & 3) || (it.Cx & 3) || (it.ldg & 3) || (it.ldx & 3)) return PF_ERR_UNSUPPORTED;
      PF_REQUIRE((((uintptr_t)it.gr | (uintptr_t)it.x | (uintptr_t)it.x_scale | (uintptr_t)it.x_shift) & 15) == 0);
      plans[i] = make_plan(1, it.Cg, it.Cx, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, true, it.rows_P);
      if (!plans[i].ok) return PF_ERR_UNSUPPORTED;
      plans[i].g.x_scale = it.x_scale;
      plans[i].g.x_shift = it.x_shift;
      plans[i].g.x_pps = it.x_rows_per_stat;
      plans[i].g.ldg = it.ldg;
      plans[i].g.ldx = it.ldx;
      PF_REQUIRE(it.workspace_bytes >= 4 * (int64_t)it.Cg * it.Cx * plans[i].splits);
      done[i] = false;
      continue;
    }
    PF_REQUIRE(conv_args_ok(it.N, it.Cg, it.Cx, it.Do, it.Ho, it.Wo, it.Di, it.Hi, it.Wi, it.KD, it.KH, it.KW, it.stride));
    PF_REQUIRE(it.pd >= 0 && it.ph >= 0 && it.pw >= 0 && it.x_samples_per_stat >= 1);
    plans[i] = make_plan(it.N, it.Cg, it.Cx, it.Do, it.Ho, it.Wo, it.Di, it.Hi, it.Wi, it.KD, it.KH, it.KW, it.stride,
                         it.pd, it.ph, it.pw, false, 0);
    if (!plans[i].ok) return PF_ERR_UNSUPPORTED;
    plans[i].g.x_scale = it.x_scale;
    plans[i].g.x_shift = it.x_shift;
    plans[i].g.x_sps = it.x_samples_per_stat;
    PF_REQUIRE(it.workspace_bytes >= 4 * (int64_t)it.Cg * it.Cx * plans[i].g.T * plans[i].splits);
    done[i] = false;
  }
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const int MT = plans[i].MT, stride = items[i].stride, ntw = (plans[i].g.NTILES + 3) / 4;
    WgBatchArgs b;
    b.n = 0;
    int blocks = 0;
    size_t lds = 0;
    auto flush = [&]() {
      if (b.n == 0) return PF_OK;
      b.first_block[b.n] = blocks;
      const int rc = launch_wgrad_batch_any(b, MT, stride, ntw, blocks, lds, s);
      b.n = 0;
      blocks = 0;
      lds = 0;
      return rc;
    };
    for (int j = i; j < n; ++j) {
      if (done[j] || plans[j].MT != MT || items[j].stride != stride || (plans[j].g.NTILES + 3) / 4 != ntw) continue;
      const WgPlan& p = plans[j];
      b.Gr[b.n] = items[j].gr;
      b.X[b.n] = items[j].x;
      b.part[b.n] = reinterpret_cast<float*>(items[j].workspace);
      b.g[b.n] = p.g;
      b.first_block[b.n] = blocks;
      b.gx[b.n] = p.splits;
      b.gy[b.n] = p.cblocks;
      blocks += p.splits * p.cblocks * p.mblocks;
      lds = p.lds_bytes > lds ? p.lds_bytes : lds;
      done[j] = true;
      if (++b.n == kWgBatch) {
        const int rc = flush();
        if (rc != PF_OK) return rc;
      }
    }
    const int rc = flush();
    if (rc != PF_OK) return rc;
  }
  return PF_OK;
}

int64_t pf_rows_wgrad_workspace(int64_t P, int Cg, int Cx) {
  if (P < 1 || Cg < 1 || Cx < 1 || (Cg & 3) || (Cx & 3)) return -1;
  const WgPlan p = make_plan(1, Cg, Cx, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, true, P);
  if (!p.ok) return -1;
  return 4 * (int64_t)Cg * Cx * p.splits;
}

int pf_rows_wgrad_f32(const float* gr, int64_t ldg, const float* x, int64_t ldx, float* dw, int64_t P, int Cg, int Cx,
                      const float* x_scale, const float* x_shift, int64_t x_rows_per_stat, void* workspace,
                      int64_t workspace_bytes, int accumulate, void* stream) {
  PF_REQUIRE(P >= 1 && Cg >= 1 && Cx >= 1 && ldg >= Cg && ldx >= Cx && x_rows_per_stat >= 1);
  PF_REQUIRE((x_scale == nullptr) == (x_shift == nullptr) && gr && x);
  if ((Cg & 3) || (Cx & 3) || (ldg & 3) || (ldx & 3)) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE((((uintptr_t)gr | (uintptr_t)x | (uintptr_t)x_scale | (uintptr_t)x_shift) & 15) == 0);
  WgPlan p = make_plan(1, Cg, Cx, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, true, P);
  if (!p.ok) return PF_ERR_UNSUPPORTED;
  p.g.x_scale = x_scale;
  p.g.x_shift = x_shift;
  p.g.x_pps = x_rows_per_stat;
  p.g.ldg = ldg;
  p.g.ldx = ldx;
  return run_plan(gr, x, dw, p, 1, workspace, workspace_bytes, accumulate, (hipStream_t)stream);
}

int pf_wgrad_reduce_batch_f32(const float* const* parts, float* const* dws, const int64_t* elems, const int* splits,
                              const int* rows, const int* taps, const int* swapped, int n, int accumulate, void* stream) {
  PF_REQUIRE(n >= 0 && (n == 0 || (parts && dws && elems && splits && rows && taps && swapped)));
  for (int base = 0; base < n; base += kRedBatch) {
    RedBatch b;
    b.n = n - base < kRedBatch ? n - base : kRedBatch;
    b.accumulate = accumulate;
    int64_t blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      PF_REQUIRE(parts[base + i] && dws[base + i] && elems[base + i] >= 1 && splits[base + i] >= 1);
      b.part[i] = parts[base + i];
      b.dw[i] = dws[base + i];
      b.elems[i] = elems[base + i];
      b.splits[i] = splits[base + i];
      b.rows[i] = rows[base + i];
      b.taps[i] = taps[base + i];
      b.swapped[i] = swapped[base + i] ? 1 : 0;
      PF_REQUIRE(b.rows[i] >= 1 && b.taps[i] >= 1 && elems[base + i] % ((int64_t)b.rows[i] * b.taps[i]) == 0);
      b.first_block[i] = (int)blocks;
      blocks += pf_cdiv(elems[base + i], kRedEl);
      PF_REQUIRE(blocks <= INT32_MAX);
    }
    b.first_block[b.n] = (int)blocks;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(kRedThreads), 0, (hipStream_t)stream, b);
    const int rc = pf_launch_status();
    if (rc != PF_OK) return rc;
  }
  return PF_OK;
}

}  // extern "C"
