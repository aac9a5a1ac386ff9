// Rows E0/E1/E2, M, H (+T): the EdgeConv chain, the flow MLP and the flow head.
//
// The reference runs, per EdgeConv, two 1x1 convs, a gather that materialises (B,C,N,16), expand, cat,
// BatchNorm2d (batch statistics), ReLU and mean -- five passes over 105-210 MB tensors per 25 600
// points (reference networks.py:18-45, :56-81).  Here the (N,16,C) edge tensor never exists:
//
//   pointwise_gemm   [l | e] = x . [W1;W2]^T on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
//                    an fmaf chain in k order), 128-point tiles, K streamed through double-buffered
//                    LDS, optional BatchNorm+ReLU of the previous layer fused into the A staging,
//                    optional per-block float64 column sums (statistics of the "central" half and of
//                    the MLP layers).
//   edge_stats       pass A: per-channel sum / sum-of-squares of d = e[idx] - l over all (point,
//                    neighbour) pairs -> per-block float64 partials.
//   bn_finalize      partials -> scale/shift per stat group, running-stat update (train-mode BN runs
//                    at test time too: reference test.py:58).  Fixed summation order:
//                    results are bit-reproducible run to run (no float atomics anywhere).
//   edge_apply       pass B: y = mean_j relu(scale*d + shift) (+ the central half), written straight
//                    into its slice of the (N,224) concat buffer that feeds the MLP.
//   flow_head        last BN+ReLU, the 16->1 conv, softmax over the 5 hypotheses, expected offset, add
//                    to the prior depth, un-tiling of the sub-grid order.
//
// Activations are point-major (N, C): a neighbour's C channels are one or two cache lines, and a wave
// reads them as 16-byte lanes (C/4 lanes per neighbour row), so the irregular gather is line-granular
// and mostly L2-resident (e of 25 600 points = 3.3-6.5 MB).  BatchNorm statistics are per stat group
// (= per sub-grid in test mode, pooled over the batch for the nn.Module API).
//
// Algorithmic HBM bytes per point: EdgeConv(C_in->C_out, out C'):
// 4*C_in + 8*16 + 16*4*C_out + 4*C'.
#include <stdlib.h>

#include "pf_common.h"
#include "pf_bn_resolve.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE = PF_GEMM_TILE;

// ------------------------------------------------------------------------------------------------
// pointwise GEMM on f32 MFMA
//
// Block = 256 threads = 4 waves, tile = 128 points x Nc columns; wave w owns points [32w, 32w+32) and all
// NT = Nc/32 column tiles (NT accumulators of 16 registers).  K is streamed in chunks of KC through two
// LDS buffers: chunk c+1 travels global -> registers while the matrix cores work on chunk c, then
// registers -> LDS, one barrier per chunk.  A is kept k-major in LDS ([k][point], row stride 130) so the
// MFMA A operand (lane = point, k = lane>>5) and the B operand (lane = column) are both conflict-free
// 32-lane rows.  v_mfma_f32_32x32x2_f32 is an exact float32 fmaf chain in k order (64 cycles/SIMD);
// the bound is the fp32 matrix peak (157 TF), not HBM: 2*K*Nc flop per point vs 4*(K+Nc) bytes.
// ------------------------------------------------------------------------------------------------
constexpr int GT = 128;        // points per GEMM tile
constexpr int LDA = GT + 2;    // 130: 4*LDA = 8 (mod 32) -> transposed staging writes are <= 2-way (free)

template <bool POINT_MAJOR, int NT>
__global__ __launch_bounds__(256, 2) void pointwise_gemm_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ Wt, float* __restrict__ Y, int64_t ldy,
    int Ng, int K, int Nc_store, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int groups_per_stat, double* __restrict__ partials, int T) {
  constexpr int NC = NT * 32;
  constexpr int KC = NT == 4 ? 16 : 32;
  constexpr int NA = KC * GT / 256;   // A floats staged per thread per chunk (16 or 8)
  constexpr int NW = KC * NC / 256;   // W floats staged per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[2][KC][LDA];
  __shared__ __attribute__((aligned(16))) float Ws[2][KC][NC];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int hi = lane >> 5, lo = lane & 31;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + GT - 1) / GT;
  const int chunks = (K + KC - 1) / KC;
  const float* sc = in_scale ? in_scale + (int64_t)(g / groups_per_stat) * K : nullptr;
  const float* sh = in_scale ? in_shift + (int64_t)(g / groups_per_stat) * K : nullptr;

  double csum[NT], csq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) csum[t] = csq[t] = 0.0;

  for (int tile = tb; tile < tiles; tile += T) {
    const int n0 = tile * GT;
    const int rows = min(GT, Ng - n0);
    float ra[NA], rw[NW];

    auto load_chunk = [&](int c) {
      const int k0 = c * KC;
      if (!POINT_MAJOR) {
        const int p = tid & (GT - 1), kr = tid >> 7;                // 2 k-rows per pass, 512 B per row
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          const int k = k0 + 2 * r + kr;
          ra[r] = (k < K && p < rows) ? X[((int64_t)g * K + k) * Ng + n0 + p] : 0.0f;
        }
      } else {
        const int kk = tid & (KC - 1), pr = tid / KC;               // KC consecutive k of 256/KC rows per pass
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          const int p = r * (256 / KC) + pr;
          const int k = k0 + kk;
          ra[r] = (k < K && p < rows) ? X[((int64_t)g * Ng + n0 + p) * ldx + k] : 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const int e = tid + 256 * r;
        const int k = k0 + e / NC, j = e % NC;
        rw[r] = k < K ? Wt[(int64_t)k * NC + j] : 0.0f;
      }
    };
    auto store_chunk = [&](int c, int buf) {
      const int k0 = c * KC;
      if (!POINT_MAJOR) {
        const int p = tid & (GT - 1), kr = tid >> 7;
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          const int kl = 2 * r + kr;
          float v = ra[r];
          if (sc && k0 + kl < K && p < rows) v = fmaxf(fmaf(v, sc[k0 + kl], sh[k0 + kl]), 0.0f);
          As[buf][kl][p] = v;
        }
      } else {
        const int kk = tid & (KC - 1), pr = tid / KC;
        float s1 = 1.0f, s0 = 0.0f;
        const bool aff = sc && (k0 + kk < K);
        if (aff) {
          s1 = sc[k0 + kk];
          s0 = sh[k0 + kk];
        }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
          const int p = r * (256 / KC) + pr;
          float v = ra[r];
          if (aff && p < rows) v = fmaxf(fmaf(v, s1, s0), 0.0f);
          As[buf][kk][p] = v;
        }
      }
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const int e = tid + 256 * r;
        Ws[buf][e / NC][e % NC] = rw[r];
      }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    __syncthreads();                    // previous tile's last chunk has been consumed
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < chunks) load_chunk(c + 1);
#pragma unroll 4
      for (int kp = 0; kp < KC / 2; ++kp) {
        const float a = As[buf][2 * kp + hi][32 * wave + lo];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float b = Ws[buf][2 * kp + hi][32 * t + lo];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
      if (c + 1 < chunks) store_chunk(c + 1, buf ^ 1);
      __syncthreads();
    }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = 32 * t + lo;
      float cs = 0.0f, cq = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float v = acc[t][r];
        if (row < rows) {
          if (col < Nc_store) Y[((int64_t)g * Ng + n0 + row) * ldy + col] = v;
          cs += v;
          cq += v * v;
        }
      }
      csum[t] += (double)cs;
      csq[t] += (double)cq;
    }
  }

  if (partials != nullptr) {
    __syncthreads();
    double* red = reinterpret_cast<double*>(&As[0][0][0]);  // [wave][NC][2] doubles <= 8 KB
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double s = csum[t], q = csq[t];
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lane < 32) {
        red[((wave * NC) + 32 * t + lane) * 2 + 0] = s;
        red[((wave * NC) + 32 * t + lane) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < NC) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[((w * NC) + tid) * 2 + 0];
        q += red[((w * NC) + tid) * 2 + 1];
      }
      double* o = partials + (((int64_t)g * T + tb) * NC + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pointwise GEMM, "direct A" form for point-major rows (the PointFlow chain's six shapes)
//
// The kernel above stages BOTH operands through LDS in K chunks with a barrier per chunk; on these skinny
// GEMMs (K <= 224, Nc <= 128) a chunk holds 0.4 us of MFMA work against ~2 us of load latency, so the matrix
// cores idle (18-35 % of the f32 peak whatever the staging variant, profiles/r01l_microbench_gemm_*.log).
// Here the A operand never touches LDS: the reduction index k may be assigned to (MFMA step, lane half) in any
// order as long as A and B agree, so lane (row = lane & 31, half h = lane >> 5) loads its point's row straight
// from global memory as 16-byte pieces  k = 8 j + 4 h + {0..3}  and feeds element i of piece j to step
// (j, i) of v_mfma_f32_32x32x2_f32 -- every loaded byte is used, all of a tile's loads are in flight at once
// (KJ independent dwordx4 per lane, the waits fall between the MFMA groups), waves never synchronise inside a
// tile, and the whole W (K x Nc <= 16 384 floats) plus the fused BatchNorm affine of the previous layer sit in
// LDS for the block's lifetime (one barrier per block instead of one per chunk).  The summation order differs
// from the chunked kernel's (still one exact f32 fmaf chain per output), so results agree to rounding, not bits.
// ------------------------------------------------------------------------------------------------
// AFFINE: 0 = rows as they are; 1 = relu(x * in_scale + in_shift); 2 = the same, the rows computed by every block
// from the producer's statistics (pf_bn_resolve, pf_bn_resolve.h: the pending BatchNorm gets no launch of its own)
// (EdgeConv 64 -> [64 | 64], KJ = 8, NT = 4: the default allocation is 138 VGPRs + 64 AGPRs = 2 waves per SIMD; asked for 3
// the compiler fits it in 168 registers without scratch -- stand-alone 12.4 -> 11.5 us at 25 600 points, 29.7 -> 28.6 at
// 102 400; headline unchanged, profiles/r06n_resolve_batch.md)
template <int KJ, int NT, int AFFINE>
__global__ __launch_bounds__(256, (KJ == 8 && NT == 4) ? 3 : 1) void pointwise_gemm_direct_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ Wt, float* __restrict__ Y, int64_t ldy,
    int Ng, int K, int Nc_store, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int groups_per_stat, double* __restrict__ partials, int T, pf_bn_job in_bn) {
  constexpr int NC = NT * 32;
  constexpr int KP = KJ * 8;                       // K rounded up to whole pieces
  constexpr int J0 = (KJ + 1) / 2, J1 = KJ - J0;   // the two halves of a row's pieces (software pipeline below)
  __shared__ __attribute__((aligned(16))) float Ws[KP * NC];
  __shared__ __attribute__((aligned(16))) float Aff[2][AFFINE ? KP : 4];
  constexpr bool kAlias = sizeof(float) * KP * NC >= sizeof(double) * (4 * NC * 2);
  __shared__ double red_own[kAlias ? 1 : 4 * NC * 2];
  double* red = kAlias ? reinterpret_cast<double*>(Ws) : red_own;      // W is dead once the tiles are done
  __shared__ double bn_red[AFFINE == 2 ? 512 : 1];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int hi = lane >> 5, lo = lane & 31;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + GT - 1) / GT;

  {
    // W -> LDS in 16-byte pieces, eight loads in flight per thread (a scalar copy loop is one dependent L2 round
    // trip per element and costs more than the tile's matrix work)
    const float4* W4 = reinterpret_cast<const float4*>(Wt);
    float4* Ws4 = reinterpret_cast<float4*>(Ws);
    constexpr int N4 = KP * NC / 4;
    const int valid4 = K * NC / 4;                 // K*NC is a multiple of 4 (NC % 32 == 0)
    for (int e0 = 0; e0 < N4; e0 += 256 * 8) {
      float4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + tid + 256 * u;
        r[u] = e < valid4 ? W4[e] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + tid + 256 * u;
        if (e < N4) Ws4[e] = r[u];
      }
    }
  }
  if (AFFINE == 1) {
    const float* sc = in_scale + (int64_t)(g / groups_per_stat) * K;
    const float* sh = in_shift + (int64_t)(g / groups_per_stat) * K;
    for (int e = tid; e < KP; e += 256) {
      Aff[0][e] = e < K ? sc[e] : 0.0f;
      Aff[1][e] = e < K ? sh[e] : 0.0f;
    }
  }
  if (AFFINE == 2) {
    if (tid < KP - K) Aff[0][K + tid] = Aff[1][K + tid] = 0.0f;
    pf_bn_resolve<256>(in_bn, g / groups_per_stat, Aff[0], Aff[1], bn_red);   // (its W loads are in flight meanwhile)
  }
  __syncthreads();

  double csum[NT], csq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) csum[t] = csq[t] = 0.0;

  // this lane's row of tile `tile` (lanes past the end of the group shadow its last point; never stored)
  auto row_ptr = [&](int tile) -> const float* {
    int n = tile * GT + 32 * wave + lo;
    n = n < Ng ? n : Ng - 1;
    return X + ((int64_t)g * Ng + n) * ldx + 4 * hi;
  };
  // pieces [JB, JB+JN) of a row: 16-byte loads, all in flight together.  The last piece of the upper half may lie
  // beyond K (K % 8 == 4): it re-reads the previous piece, its weights are zero.
  auto load_half = [&](const float* xrow, float4* a, int jb, int jn) {
#pragma unroll
    for (int j = 0; j < (J0 > J1 ? J0 : J1); ++j)
      if (j < jn) {
        const int jj = jb + j;
        int kb = 8 * jj;                                          // an immediate offset, except for the last piece
        if (jj == KJ - 1 && 8 * jj + 4 * hi >= K) kb -= 4;
        a[j] = *reinterpret_cast<const float4*>(xrow + kb);
      }
  };
  const float* wbase = &Ws[(4 * hi) * NC + lo];
  auto mma_half = [&](const float4* a, int jb, int jn, f32x16* acc) {
    // B operands one piece ahead, and no further: left alone, the scheduler hoists a whole tile's LDS reads
    // (4*NT*KJ registers) above the MFMAs and the kernel drops to one wave per SIMD
    float bcur[4][NT], bnxt[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) bcur[i][t] = wbase[(8 * jb + i) * NC + 32 * t];
#pragma unroll
    for (int j = 0; j < (J0 > J1 ? J0 : J1); ++j) {
      if (j < jn) {
        const int jj = jb + j;
        if (j + 1 < jn) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) bnxt[i][t] = wbase[(8 * (jj + 1) + i) * NC + 32 * t];
        }
        float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
        if (AFFINE) {
          const float4 s1 = *reinterpret_cast<const float4*>(&Aff[0][8 * jj + 4 * hi]);
          const float4 s0 = *reinterpret_cast<const float4*>(&Aff[1][8 * jj + 4 * hi]);
          av[0] = fmaxf(fmaf(av[0], s1.x, s0.x), 0.0f);
          av[1] = fmaxf(fmaf(av[1], s1.y, s0.y), 0.0f);
          av[2] = fmaxf(fmaf(av[2], s1.z, s0.z), 0.0f);
          av[3] = fmaxf(fmaf(av[3], s1.w, s0.w), 0.0f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bcur[i][t], acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) bcur[i][t] = bnxt[i][t];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // Software pipeline over (tile, half): while the matrix cores work on one half of a row, the other half -- of
  // this tile, then of the block's next tile -- is already in flight.
  float4 a0[J0], a1[J1 > 0 ? J1 : 1];
  if (tb < tiles) load_half(row_ptr(tb), a0, 0, J0);
  for (int tile = tb; tile < tiles; tile += T) {
    const int n0 = tile * GT + 32 * wave;
    const int rows = min(32, Ng - n0);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
    if (J1 > 0) load_half(row_ptr(tile), a1, J0, J1);
    mma_half(a0, 0, J0, acc);
    if (tile + T < tiles) load_half(row_ptr(tile + T), a0, 0, J0);
    if (J1 > 0) mma_half(a1, J0, J1, acc);
    // epilogue: C/D layout col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
    if (rows > 0) {
      float* yrow = Y + ((int64_t)g * Ng + n0 + 4 * hi) * ldy + lo;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 32 * t + lo;
        float cs = 0.0f, cq = 0.0f;
        if (rows == 32) {                                // full tile: unconditional rows
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float v = acc[t][q];
            if (col < Nc_store) yrow[(int64_t)((q & 3) + 8 * (q >> 2)) * ldy + 32 * t] = v;
            cs += v;
            cq += v * v;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * hi;
            const float v = acc[t][q];
            if (row < rows) {
              if (col < Nc_store) yrow[(int64_t)((q & 3) + 8 * (q >> 2)) * ldy + 32 * t] = v;
              cs += v;
              cq += v * v;
            }
          }
        }
        csum[t] += (double)cs;
        csq[t] += (double)cq;
      }
    }
  }

  if (partials != nullptr) {
    __syncthreads();                                   // every wave is done reading W (red may alias it)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double s = csum[t], q = csq[t];
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lane < 32) {
        red[((wave * NC) + 32 * t + lane) * 2 + 0] = s;
        red[((wave * NC) + 32 * t + lane) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < NC) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += red[((w * NC) + tid) * 2 + 0];
        q += red[((w * NC) + tid) * 2 + 1];
      }
      double* o = partials + (((int64_t)g * T + tb) * NC + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// EdgeConv pass A / pass B.  C/4 lanes share one point (16-byte lanes), 256/(C/4) points per pass.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Neighbour rows of one point, all in flight at once.  With k known at compile time the 16 indices come
// in as 8 sixteen-byte loads and the 16 row loads are issued back to back; the runtime-k loop below would
// otherwise serialise index load -> address -> row load once per neighbour (two dependent global
// latencies x 16 per point: the kernels were latency-bound at ~1/6 of their L2 rate).
template <int C, int K>
__device__ __forceinline__ void gather_rows(const float* __restrict__ LE, int64_t ldle, const int64_t* __restrict__ ip,
                                            int64_t gbase, int Ng, int q, float4 (&e)[K], bool& bad) {
  long long ii[K];
  static_assert(K % 2 == 0, "K must be even");
#pragma unroll
  for (int j = 0; j < K / 2; ++j) {
    const longlong2 v = *reinterpret_cast<const longlong2*>(ip + 2 * j);
    ii[2 * j] = v.x;
    ii[2 * j + 1] = v.y;
  }
#pragma unroll
  for (int j = 0; j < K; ++j) {
    long long i = ii[j];
    if (i < 0 || i >= Ng) {
      bad = true;
      i = i < 0 ? 0 : Ng - 1;
    }
    e[j] = ld4(LE + (gbase + i) * ldle + C + 4 * q);
  }
}

// The lattice form of the same gather: the neighbours of point n are n + offset(code_j) for the 16 window
// codes the lattice kNN wrote (16 bytes per point instead of 128 bytes of int64 indices, read by six passes
// per PointFlow iteration), clamped to the group like get_knn_3d clamps (reference utils/torch_utils.py:55-59).
// `lut` (LDS) maps a code to its offset (pd-hk)*H*W + (ph-hk)*W + (pw-hk).
struct Lattice {
  int ks, H, W;     // window size and lattice plane shape; ks == 0: neighbours come from the int64 index tensor
  int band;         // > 0: XCD-aware tile order (xcd_tile below), tiles of a plane per XCD; 0: tile = block
};

// Workgroups go to the 8 XCDs round-robin (block b -> XCD b % 8) and every XCD has its own 4 MB L2.  With tile = block,
// the eight neighbours of a tile -- which gather the same rows -- sit on eight different L2s, so every L2 ends up holding
// the whole lattice's rows.  With a band, XCD x owns the x-th eighth of EVERY plane (the window reaches all planes at
// the same pixels, +-2 rows): its L2 holds an eighth of the rows plus a halo.  Needs whole tiles per plane, a multiple
// of 8 of them, and one tile per block (the host side checks); the partial rows stay in tile order (same bits).
__device__ __forceinline__ int xcd_tile(int tb, const Lattice& lat) {
  if (lat.band == 0) return tb;
  const int x = tb & 7, j = tb >> 3;
  const int d = j / lat.band, o = j - d * lat.band;
  return d * (lat.band * 8) + x * lat.band + o;
}

__device__ __forceinline__ void build_code_lut(int* lut, const Lattice& lat) {
  const int hk = lat.ks >> 1, k2 = lat.ks * lat.ks;
  for (int c = threadIdx.x; c < 256; c += blockDim.x) {
    int off = 0;
    if (c < k2 * lat.ks) {
      const int pd = c / k2, rem = c - pd * k2, ph = rem / lat.ks, pw = rem - ph * lat.ks;
      off = (pd - hk) * lat.H * lat.W + (ph - hk) * lat.W + (pw - hk);
    }
    lut[c] = off;
  }
  __syncthreads();
}

template <int C>
__device__ __forceinline__ void gather_rows_codes(const float* __restrict__ LE, int64_t ldle,
                                                  const uint8_t* __restrict__ cp, const int* lut, int64_t gbase, int n,
                                                  int Ng, int q, float4 (&e)[16]) {
  const uint4 cw = *reinterpret_cast<const uint4*>(cp);
  const unsigned w[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int code = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
    int i = n + lut[code];
    i = i < 0 ? 0 : (i > Ng - 1 ? Ng - 1 : i);
    e[j] = ld4(LE + (gbase + i) * ldle + C + 4 * q);
  }
}

template <int C, int K>   // K == 0: neighbour count known only at run time
__global__ __launch_bounds__(256) void edge_stats_kernel(const float* __restrict__ LE, int64_t ldle,
                                                         const int64_t* __restrict__ idx, int k, int Ng,
                                                         double* __restrict__ partials, int T,
                                                         unsigned* __restrict__ status,
                                                         const uint8_t* __restrict__ codes, Lattice lat) {
  constexpr int Q = C / 4;         // lanes per point
  constexpr int PPB = 256 / Q;     // points per pass
  __shared__ double red[256 * 8];
  __shared__ int lut[256];
  if (K == 16 && codes != nullptr) build_code_lut(lut, lat);
  const int tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + TILE - 1) / TILE;
  const int64_t gbase = (int64_t)g * Ng;
  double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
  bool bad = false;
  const int tile0 = xcd_tile(tb, lat);
  for (int tile = tile0; tile < tiles; tile += T) {
    const int n0 = tile * TILE;
    for (int p = pl; p < TILE; p += PPB) {
      const int n = n0 + p;
      if (n >= Ng) break;
      const int64_t row = gbase + n;
      const float4 l = ld4(LE + row * ldle + 4 * q);
      const int64_t* ip = idx + row * k;
      float4 s = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
      if constexpr (K > 0) {
        float4 e[K > 0 ? K : 1];
        if constexpr (K == 16) {
          if (codes != nullptr) gather_rows_codes<C>(LE, ldle, codes + row * 16, lut, gbase, n, Ng, q, e);
          else gather_rows<C, 16>(LE, ldle, ip, gbase, Ng, q, e, bad);
        } else {
          gather_rows<C, (K > 0 ? K : 2)>(LE, ldle, ip, gbase, Ng, q, e, bad);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float dx = e[j].x - l.x, dy = e[j].y - l.y, dz = e[j].z - l.z, dw = e[j].w - l.w;
          s.x += dx; s.y += dy; s.z += dz; s.w += dw;
          s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
      } else {
        for (int j = 0; j < k; ++j) {
          int64_t i = ip[j];
          if (i < 0 || i >= Ng) {
            bad = true;
            i = i < 0 ? 0 : Ng - 1;
          }
          const float4 e = ld4(LE + (gbase + i) * ldle + C + 4 * q);
          const float dx = e.x - l.x, dy = e.y - l.y, dz = e.z - l.z, dw = e.w - l.w;
          s.x += dx; s.y += dy; s.z += dz; s.w += dw;
          s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
      }
      ds[0] += (double)s.x; ds[1] += (double)s.y; ds[2] += (double)s.z; ds[3] += (double)s.w;
      dq[0] += (double)s2.x; dq[1] += (double)s2.y; dq[2] += (double)s2.z; dq[3] += (double)s2.w;
    }
  }
  if (bad) atomicOr(status, PF_STATUS_BAD_INDEX);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    red[tid * 8 + c] = ds[c];
    red[tid * 8 + 4 + c] = dq[c];
  }
  __syncthreads();
  if (tid < 2 * C) {
    const int qq = tid / 8, comp = tid % 8;
    double acc = 0.0;
    for (int s = 0; s < PPB; ++s) acc += red[(s * Q + qq) * 8 + comp];
    double* o = partials + (((int64_t)g * T + tile0) * C + 4 * qq + (comp & 3)) * 2 + (comp >> 2);
    *o = acc;
  }
}

template <int C, int K>
__global__ __launch_bounds__(256) void edge_apply_kernel(const float* __restrict__ LE, int64_t ldle,
                                                         const int64_t* __restrict__ idx, int k, int Ng,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int ld_affine,
                                                         int groups_per_stat, int concat, float* __restrict__ Y,
                                                         int64_t ldy, int T, const uint8_t* __restrict__ codes,
                                                         Lattice lat) {
  constexpr int Q = C / 4;
  constexpr int PPB = 256 / Q;
  __shared__ int lut[256];
  if (K == 16 && codes != nullptr) build_code_lut(lut, lat);
  const int tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + TILE - 1) / TILE;
  const int64_t gbase = (int64_t)g * Ng;
  const int64_t so = (int64_t)(g / groups_per_stat) * ld_affine;
  const int doff = concat ? C : 0;
  const float4 dsc = ld4(scale + so + doff + 4 * q), dsh = ld4(shift + so + doff + 4 * q);
  float4 csc = {0, 0, 0, 0}, csh = {0, 0, 0, 0};
  if (concat) {
    csc = ld4(scale + so + 4 * q);
    csh = ld4(shift + so + 4 * q);
  }
  const float kf = (float)k;
  bool bad = false;
  for (int tile = xcd_tile(tb, lat); tile < tiles; tile += T) {
    const int n0 = tile * TILE;
    for (int p = pl; p < TILE; p += PPB) {
      const int n = n0 + p;
      if (n >= Ng) break;
      const int64_t row = gbase + n;
      const float4 l = ld4(LE + row * ldle + 4 * q);
      const int64_t* ip = idx + row * k;
      float4 a = {0, 0, 0, 0};
      if constexpr (K > 0) {
        float4 e[K > 0 ? K : 1];
        if constexpr (K == 16) {
          if (codes != nullptr) gather_rows_codes<C>(LE, ldle, codes + row * 16, lut, gbase, n, Ng, q, e);
          else gather_rows<C, 16>(LE, ldle, ip, gbase, Ng, q, e, bad);
        } else {
          gather_rows<C, (K > 0 ? K : 2)>(LE, ldle, ip, gbase, Ng, q, e, bad);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          a.x += fmaxf(fmaf(e[j].x - l.x, dsc.x, dsh.x), 0.0f);
          a.y += fmaxf(fmaf(e[j].y - l.y, dsc.y, dsh.y), 0.0f);
          a.z += fmaxf(fmaf(e[j].z - l.z, dsc.z, dsh.z), 0.0f);
          a.w += fmaxf(fmaf(e[j].w - l.w, dsc.w, dsh.w), 0.0f);
        }
      } else {
        for (int j = 0; j < k; ++j) {
          int64_t i = ip[j];
          i = i < 0 ? 0 : (i >= Ng ? Ng - 1 : i);
          const float4 e = ld4(LE + (gbase + i) * ldle + C + 4 * q);
          a.x += fmaxf(fmaf(e.x - l.x, dsc.x, dsh.x), 0.0f);
          a.y += fmaxf(fmaf(e.y - l.y, dsc.y, dsh.y), 0.0f);
          a.z += fmaxf(fmaf(e.z - l.z, dsc.z, dsh.z), 0.0f);
          a.w += fmaxf(fmaf(e.w - l.w, dsc.w, dsh.w), 0.0f);
        }
      }
      float4 yd = {a.x / kf, a.y / kf, a.z / kf, a.w / kf};
      float* yo = Y + row * ldy + 4 * q;
      if (concat) {
        float4 yc = {fmaxf(fmaf(l.x, csc.x, csh.x), 0.0f), fmaxf(fmaf(l.y, csc.y, csh.y), 0.0f),
                     fmaxf(fmaf(l.z, csc.z, csh.z), 0.0f), fmaxf(fmaf(l.w, csc.w, csh.w), 0.0f)};
        *reinterpret_cast<float4*>(yo) = yc;
        *reinterpret_cast<float4*>(yo + C) = yd;
      } else {
        *reinterpret_cast<float4*>(yo) = yd;
      }
    }
  }
  (void)bad;
}

// ------------------------------------------------------------------------------------------------
// EdgeConv backward (BASELINE config 4, training).  The reference differentiates the composition
// conv1d / gather_knn / cat / BatchNorm2d / ReLU / mean (networks.py:18-45), which keeps the (B,2C,N,16)
// edge tensor -- 839 MB at the 102 400-point lattice -- alive for backward and scatters through it with
// atomicAdd (functions/csrc/gather_knn_kernel.cu:50-89).  Here nothing of size N*k is stored: both passes
// recompute d = e[idx] - l from the saved (N, 2C) rows [l | e].
//
// With u = a*d + b (a = gamma*invstd, b = beta - mean*a), xhat = (d - mean)*invstd, y = mean_j relu(u_j)
// and the upstream gradient G: g_j = [u_j > 0] * G / k;  dbeta = sum g;  dgamma = sum g*xhat;
// dd_j = a * (g_j - dbeta/M - xhat_j * dgamma/M)  (M = elements behind the batch statistics);
// dl = -sum_j dd_j,  de[idx_j] += dd_j.  The central half of EdgeConv (k identical copies of l through the
// same BatchNorm) reduces to  dl += a_c * (g_c - dbeta_c/N - xhat_c * dgamma_c/N),  g_c = [u_c > 0] * G_c.
//
//   pass 1 (edge_bwd_reduce)  per-block float64 partial sums of (g, g*xhat) per channel -> dbeta, dgamma
//   pass 2 (edge_bwd_apply)   dl rows (plain stores) and, SCATTER, de rows by float atomics like the reference
//   pass 3 (edge_bwd_inverse) de rows WITHOUT atomics: point m sums dd over the pairs (n, j) that gathered it, in
//                             ascending pair order, from the inverted index tensor (knn_inverse.hip) -- the same
//                             dd expression as pass 2, so dl and de are bit-reproducible from run to run
// Round 6, the default with inverted lists: TWO walks.  dl = -sum_j dd_j is linear in (dbeta, dgamma):
//   dl = -a * ((sum_j g_j - k * dbeta/M) - (dgamma/M) * sum_j xhat_j),
// so pass 1 (edge_bwd_reduce<.., SUMS>) leaves each point's (sum_j g_j | sum_j xhat_j) in its dLE row and pass 3
// (edge_bwd_inverse<.., FINISH>) turns them into dl while it gathers de: pass 2's walk over the forward lists (186 us of
// the config-4 step) is gone; dl differs from the three-pass form in float32 rounding only.
// The two GEMMs that follow (dX = [dl|de] W, dW = [dl|de]^T X) are pointwise_gemm / conv_wgrad launches.
// ------------------------------------------------------------------------------------------------
struct EdgeBwdAffine {
  const float* scale;      // (S, ld) [central C | diff C] (concat) or [diff C]
  const float* shift;
  const float* mean;
  const float* invstd;
  const float* c1;         // pass 2: dbeta / M  (central columns: / N)
  const float* c2;         // pass 2: dgamma / M
  int ld, groups_per_stat, concat;
};

template <int C, int K, bool SUMS>
__global__ __launch_bounds__(256) void edge_bwd_reduce_kernel(const float* __restrict__ LE, int64_t ldle,
                                                              const int64_t* __restrict__ idx, int k, int Ng,
                                                              const float* __restrict__ Gy, int64_t ldg,
                                                              EdgeBwdAffine A, double* __restrict__ partials, int T,
                                                              unsigned* __restrict__ status, float* __restrict__ sums,
                                                              float* __restrict__ gacc, int64_t lda, int band) {
  constexpr int Q = C / 4;
  constexpr int PPB = 256 / Q;
  __shared__ double red[256 * 16];
  const int tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + TILE - 1) / TILE;
  const int64_t gbase = (int64_t)g * Ng;
  const int64_t so = (int64_t)(g / A.groups_per_stat) * A.ld;
  const int doff = A.concat ? C : 0;
  const float4 a = ld4(A.scale + so + doff + 4 * q), b = ld4(A.shift + so + doff + 4 * q);
  const float4 mu = ld4(A.mean + so + doff + 4 * q), is = ld4(A.invstd + so + doff + 4 * q);
  float4 ac = {0, 0, 0, 0}, bc = {0, 0, 0, 0}, muc = {0, 0, 0, 0}, isc = {0, 0, 0, 0};
  if (A.concat) {
    ac = ld4(A.scale + so + 4 * q);
    bc = ld4(A.shift + so + 4 * q);
    muc = ld4(A.mean + so + 4 * q);
    isc = ld4(A.invstd + so + 4 * q);
  }
  const float kf = (float)k;
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  bool bad = false;
  const Lattice xo{0, 1, 1, band};                    // XCD-aware tile order (xcd_tile); the partial row stays the tile's
  const int tile0 = xcd_tile(tb, xo);
  for (int tile = tile0; tile < tiles; tile += T) {
    const int n0 = tile * TILE;
    for (int p = pl; p < TILE; p += PPB) {
      const int n = n0 + p;
      if (n >= Ng) break;
      const int64_t row = gbase + n;
      const float4 l = ld4(LE + row * ldle + 4 * q);
      float4 gy = ld4(Gy + row * ldg + doff + 4 * q);
      if (SUMS && gacc != nullptr) {     // a second gradient meets this one here (the next layer's data gradient):
        float* ga = gacc + row * lda + doff + 4 * q;           // the sum replaces it, the later passes read gacc
        const float4 h = ld4(ga);
        gy = make_float4(gy.x + h.x, gy.y + h.y, gy.z + h.z, gy.w + h.w);
        *reinterpret_cast<float4*>(ga) = gy;
      }
      const float gd[4] = {gy.x / kf, gy.y / kf, gy.z / kf, gy.w / kf};
      const float lv[4] = {l.x, l.y, l.z, l.w};
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
      const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
      float sg[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0}, sd[4] = {0, 0, 0, 0};
      const int64_t* ip = idx + row * k;
      auto pair = [&](const float4& e) {
        const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = ev[c] - lv[c];
          const float u = fmaf(d, av[c], bv[c]);
          const float gg = u > 0.0f ? gd[c] : 0.0f;
          sg[c] += gg;
          sx[c] += gg * ((d - mv[c]) * iv[c]);
          if (SUMS) sd[c] += d;
        }
      };
      if constexpr (K > 0) {
        float4 e[K > 0 ? K : 1];
        gather_rows<C, (K > 0 ? K : 2)>(LE, ldle, ip, gbase, Ng, q, e, bad);
#pragma unroll
        for (int j = 0; j < K; ++j) pair(e[j]);
      } else {
        for (int j = 0; j < k; ++j) {
          int64_t i = ip[j];
          if (i < 0 || i >= Ng) {
            bad = true;
            i = i < 0 ? 0 : Ng - 1;
          }
          pair(ld4(LE + (gbase + i) * ldle + C + 4 * q));
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] += (double)sg[c];
        acc[4 + c] += (double)sx[c];
      }
      if (SUMS) {   // the point's own sums over its k pairs: [sum g | sum xhat] in the row the finish pass turns into [dl | de]
        float* o = sums + row * ldle + 4 * q;
        *reinterpret_cast<float4*>(o) = make_float4(sg[0], sg[1], sg[2], sg[3]);
        *reinterpret_cast<float4*>(o + C) = make_float4((sd[0] - kf * mv[0]) * iv[0], (sd[1] - kf * mv[1]) * iv[1],
                                                        (sd[2] - kf * mv[2]) * iv[2], (sd[3] - kf * mv[3]) * iv[3]);
      }
      if (A.concat) {
        float4 gc4 = ld4(Gy + row * ldg + 4 * q);
        if (SUMS && gacc != nullptr) {
          float* ga = gacc + row * lda + 4 * q;
          const float4 h = ld4(ga);
          gc4 = make_float4(gc4.x + h.x, gc4.y + h.y, gc4.z + h.z, gc4.w + h.w);
          *reinterpret_cast<float4*>(ga) = gc4;
        }
        const float gcv[4] = {gc4.x, gc4.y, gc4.z, gc4.w};
        const float acv[4] = {ac.x, ac.y, ac.z, ac.w}, bcv[4] = {bc.x, bc.y, bc.z, bc.w};
        const float mcv[4] = {muc.x, muc.y, muc.z, muc.w}, icv[4] = {isc.x, isc.y, isc.z, isc.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float u = fmaf(lv[c], acv[c], bcv[c]);
          const float gg = u > 0.0f ? gcv[c] : 0.0f;
          acc[8 + c] += (double)gg;
          acc[12 + c] += (double)(gg * ((lv[c] - mcv[c]) * icv[c]));
        }
      }
    }
  }
  if (bad) atomicOr(status, PF_STATUS_BAD_INDEX);
#pragma unroll
  for (int i = 0; i < 16; ++i) red[tid * 16 + i] = acc[i];
  __syncthreads();
  // partial row (cols = C or 2C, 2 doubles each): column order follows the BatchNorm channels
  const int cols = A.concat ? 2 * C : C;
  for (int o = tid; o < cols * 2; o += 256) {
    const int col = o >> 1, comp = o & 1;                 // comp 0: sum g, 1: sum g*xhat
    const bool central = A.concat && col < C;
    const int ch = central ? col : col - doff;
    const int qq = ch >> 2, cc = ch & 3;
    const int slot = (central ? 8 : 0) + 4 * comp + cc;
    double v = 0.0;
    for (int s = 0; s < PPB; ++s) v += red[(s * Q + qq) * 16 + slot];
    partials[(((int64_t)g * T + tile0) * cols + col) * 2 + comp] = v;
  }
}

template <int C, int K, bool SCATTER>
__global__ __launch_bounds__(256) void edge_bwd_apply_kernel(const float* __restrict__ LE, int64_t ldle,
                                                             const int64_t* __restrict__ idx, int k, int Ng,
                                                             const float* __restrict__ Gy, int64_t ldg,
                                                             EdgeBwdAffine A, float* __restrict__ dLE, int T) {
  constexpr int Q = C / 4;
  constexpr int PPB = 256 / Q;
  const int tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q;
  const int g = blockIdx.y, tb = blockIdx.x;
  const int tiles = (Ng + TILE - 1) / TILE;
  const int64_t gbase = (int64_t)g * Ng;
  const int64_t so = (int64_t)(g / A.groups_per_stat) * A.ld;
  const int doff = A.concat ? C : 0;
  const float4 a = ld4(A.scale + so + doff + 4 * q), b = ld4(A.shift + so + doff + 4 * q);
  const float4 mu = ld4(A.mean + so + doff + 4 * q), is = ld4(A.invstd + so + doff + 4 * q);
  const float4 k1 = ld4(A.c1 + so + doff + 4 * q), k2 = ld4(A.c2 + so + doff + 4 * q);
  const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
  const float c1v[4] = {k1.x, k1.y, k1.z, k1.w}, c2v[4] = {k2.x, k2.y, k2.z, k2.w};
  const float kf = (float)k;
  for (int tile = tb; tile < tiles; tile += T) {
    const int n0 = tile * TILE;
    for (int p = pl; p < TILE; p += PPB) {
      const int n = n0 + p;
      if (n >= Ng) break;
      const int64_t row = gbase + n;
      const float4 l = ld4(LE + row * ldle + 4 * q);
      const float4 gy = ld4(Gy + row * ldg + doff + 4 * q);
      const float gd[4] = {gy.x / kf, gy.y / kf, gy.z / kf, gy.w / kf};
      const float lv[4] = {l.x, l.y, l.z, l.w};
      float dl[4] = {0, 0, 0, 0};
      const int64_t* ip = idx + row * k;
      auto pair = [&](const float4& e, int64_t i) {
        const float ev[4] = {e.x, e.y, e.z, e.w};
        float* de = dLE + (gbase + i) * ldle + C + 4 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d = ev[c] - lv[c];
          const float u = fmaf(d, av[c], bv[c]);
          const float gg = u > 0.0f ? gd[c] : 0.0f;
          const float dd = av[c] * ((gg - c1v[c]) - ((d - mv[c]) * iv[c]) * c2v[c]);
          dl[c] -= dd;
          if (SCATTER) unsafeAtomicAdd(de + c, dd);
        }
      };
      if constexpr (K > 0) {
        float4 e[K > 0 ? K : 1];
        bool bad = false;
        gather_rows<C, (K > 0 ? K : 2)>(LE, ldle, ip, gbase, Ng, q, e, bad);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          int64_t i = ip[j];
          i = i < 0 ? 0 : (i >= Ng ? Ng - 1 : i);
          pair(e[j], i);
        }
      } else {
        for (int j = 0; j < k; ++j) {
          int64_t i = ip[j];
          i = i < 0 ? 0 : (i >= Ng ? Ng - 1 : i);
          pair(ld4(LE + (gbase + i) * ldle + C + 4 * q), i);
        }
      }
      if (A.concat) {
        const float4 gc4 = ld4(Gy + row * ldg + 4 * q);
        const float gcv[4] = {gc4.x, gc4.y, gc4.z, gc4.w};
        const float4 ac = ld4(A.scale + so + 4 * q), bc = ld4(A.shift + so + 4 * q);
        const float4 muc = ld4(A.mean + so + 4 * q), isc = ld4(A.invstd + so + 4 * q);
        const float4 k1c = ld4(A.c1 + so + 4 * q), k2c = ld4(A.c2 + so + 4 * q);
        const float acv[4] = {ac.x, ac.y, ac.z, ac.w}, bcv[4] = {bc.x, bc.y, bc.z, bc.w};
        const float mcv[4] = {muc.x, muc.y, muc.z, muc.w}, icv[4] = {isc.x, isc.y, isc.z, isc.w};
        const float c1c[4] = {k1c.x, k1c.y, k1c.z, k1c.w}, c2c[4] = {k2c.x, k2c.y, k2c.z, k2c.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float u = fmaf(lv[c], acv[c], bcv[c]);
          const float gg = u > 0.0f ? gcv[c] : 0.0f;
          dl[c] += acv[c] * ((gg - c1c[c]) - ((lv[c] - mcv[c]) * icv[c]) * c2c[c]);
        }
      }
      *reinterpret_cast<float4*>(dLE + row * ldle + 4 * q) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    }
  }
}

template <int C, bool FINISH>
__global__ __launch_bounds__(256, 3) void edge_bwd_inverse_kernel(const float* __restrict__ LE, int64_t ldle, int k, int Ng,
                                                               const float* __restrict__ Gy, int64_t ldg,
                                                               EdgeBwdAffine A, const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ start,
                                                               float* __restrict__ dLE, int64_t rows, int dbg,
                                                               int band) {
  constexpr int Q = C / 4;
  constexpr int PPB = 256 / Q;
  const int tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q;
  const int doff = A.concat ? C : 0;
  const float kf = (float)k;
  const Lattice xo{0, 1, 1, band};                    // XCD-aware order of the blocks' row groups (band > 0: one group per block)
  constexpr int NG = 64 / Q;                          // row groups of a wave
  constexpr uint32_t HEAD = 24;                       // pairs a row's own lanes walk; what is left is shared by the wave
  const int lane = (int)(tid & 63);
  const int gl = lane & ~(Q - 1), gi = lane / Q;
  // every lane stays in the loop (the tail below is a wave-level collective); `live` = the lane has a row
  for (int64_t row0 = (int64_t)xcd_tile((int)blockIdx.x, xo) * PPB; row0 < rows; row0 += (int64_t)gridDim.x * PPB) {
    const bool live = row0 + pl < rows;
    const int64_t row = live ? row0 + pl : rows - 1;
    const int g = (int)(row / Ng);
    const int64_t so = (int64_t)(g / A.groups_per_stat) * A.ld + doff + 4 * q;
    const float4 a = ld4(A.scale + so), b = ld4(A.shift + so), mu = ld4(A.mean + so), is = ld4(A.invstd + so);
    const float4 k1 = ld4(A.c1 + so), k2 = ld4(A.c2 + so);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
    const float c1v[4] = {k1.x, k1.y, k1.z, k1.w}, c2v[4] = {k2.x, k2.y, k2.z, k2.w};
    const float4 e = ld4(LE + row * ldle + C + 4 * q);
    const float ev[4] = {e.x, e.y, e.z, e.w};
    const uint32_t t0 = start[row], t1 = live ? start[row + 1] : t0;
    float de[4] = {0, 0, 0, 0};
    float4 sg4 = {0, 0, 0, 0}, sx4 = {0, 0, 0, 0}, l4 = {0, 0, 0, 0}, gc4 = {0, 0, 0, 0};
    if (FINISH && !(dbg & 1)) {       // issued before the list walk: their latency hides behind it
      sg4 = ld4(dLE + row * ldle + 4 * q);
      sx4 = ld4(dLE + row * ldle + C + 4 * q);
      if (A.concat && !(dbg & 4)) {
        l4 = ld4(LE + row * ldle + 4 * q);
        gc4 = ld4(Gy + row * ldg + 4 * q);
      }
    }
    auto pair_term = [&](const float4& lq, const float4& gq, const float* evv, float* acc) {
      const float lv[4] = {lq.x, lq.y, lq.z, lq.w};
      const float gv[4] = {gq.x / kf, gq.y / kf, gq.z / kf, gq.w / kf};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = evv[c] - lv[c];
        const float uu = fmaf(d, av[c], bv[c]);
        const float gg = uu > 0.0f ? gv[c] : 0.0f;
        acc[c] += av[c] * ((gg - c1v[c]) - ((d - mv[c]) * iv[c]) * c2v[c]);
      }
    };
    // The list's pair ids arrive Q at a time -- ONE coalesced load by the row's Q lanes, a chunk ahead of its use, handed
    // round by wave shuffles -- and the rows of 8 pairs are in flight together: ~n / 8 dependent round trips for n pairs
    // (round 4's walk: 4 ids, then 4 row pairs -- 2 dependent round trips per 4 pairs).
    const uint32_t len = t1 - t0;
    const uint32_t head = len < HEAD ? len : HEAD;
    // (wave-uniform trip counts -- the shuffles are wave collectives; a group past its own head idles through the rest)
    uint32_t idn = (uint32_t)q < head ? order[t0 + q] : 0u;
    for (uint32_t c0 = 0; __any(c0 < head); c0 += Q) {
      const uint32_t idc = idn;
      idn = c0 + Q + q < head ? order[t0 + c0 + Q + q] : 0u;
#pragma unroll
      for (int sub = 0; sub < Q; sub += 8) {
        if (!__any(c0 + sub < head)) break;
        uint32_t nrow[8];
        float4 l[8], gy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t p = (uint32_t)__shfl((int)idc, gl + sub + u);
          nrow[u] = c0 + sub + u < head ? (k == 16 ? (p >> 4) : (p / (uint32_t)k)) : (uint32_t)row;
        }
        if (c0 + sub < head) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            l[u] = ld4(LE + (int64_t)nrow[u] * ldle + 4 * q);
            gy[u] = ld4(Gy + (int64_t)nrow[u] * ldg + doff + 4 * q);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (c0 + sub + u < head) pair_term(l[u], gy[u], ev, de);
        }
      }
    }
    // Hub rows.  The kernel lasts as long as its longest list, and the config-4 step has lists of 79-87 pairs against a
    // mean of 16 (tools/knn_indegree.py: p99 = 28) -- a 25 600-point pass took 62-72 us there against 22-35 us on a lattice
    // without hubs.  What a list holds beyond HEAD pairs is shared by the wave's NG row groups: group i takes pairs
    // HEAD + i, HEAD + i + NG, ... of the owner's list (its lanes hold the same channel quads as the owner's), and the
    // partial sums join the owner's in group order -- a fixed order, so the result stays bit-reproducible.  Needs one set of
    // BatchNorm rows per wave (the lanes use their own): checked, else the owner walks its tail alone.
    {
      const bool same = __all(g == __shfl(g, 0)) != 0;
      unsigned long long todo = __ballot(live && len > HEAD);
      while (todo != 0ull) {
        const int og = (__ffsll((long long)todo) - 1) & ~(Q - 1);
        todo &= ~(((1ull << Q) - 1ull) << og);
        const uint32_t ot0 = (uint32_t)__shfl((int)t0, og), olen = (uint32_t)__shfl((int)len, og);
        float oe[4], part[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; ++c) oe[c] = __shfl(ev[c], og + q);
        const uint32_t step = same ? (uint32_t)NG : 1u;
        uint32_t t = HEAD + (same ? (uint32_t)gi : 0u);
        const bool mine = same || gl == og;
        for (; mine && t < olen; t += 4 * step) {
          uint32_t nr[4];
          float4 l[4], gy[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t tt = t + u * step;
            const uint32_t p = order[ot0 + (tt < olen ? tt : olen - 1)];
            nr[u] = k == 16 ? (p >> 4) : (p / (uint32_t)k);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            l[u] = ld4(LE + (int64_t)nr[u] * ldle + 4 * q);
            gy[u] = ld4(Gy + (int64_t)nr[u] * ldg + doff + 4 * q);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (t + u * step < olen) pair_term(l[u], gy[u], oe, part);
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = __shfl(part[c], i * Q + q);
            if (gl == og) de[c] += v;
          }
        }
      }
    }
    if (FINISH) {
      // dl from the row's own sums (edge_bwd_reduce_kernel<.., true>): dl = -sum_j dd_j is linear in (c1, c2), so
      //   dl = -a * ((sum_j g_j - k * c1) - c2 * sum_j xhat_j)  -- no second walk over the neighbour rows.
      const float sgv[4] = {sg4.x, sg4.y, sg4.z, sg4.w}, sxv[4] = {sx4.x, sx4.y, sx4.z, sx4.w};
      float dl[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) dl[c] = -av[c] * ((sgv[c] - kf * c1v[c]) - c2v[c] * sxv[c]);
      if (A.concat && !(dbg & 4)) {
        const int64_t sc = (int64_t)(g / A.groups_per_stat) * A.ld + 4 * q;
        const float4 ac = ld4(A.scale + sc), bc = ld4(A.shift + sc), muc = ld4(A.mean + sc), isc = ld4(A.invstd + sc);
        const float4 k1c = ld4(A.c1 + sc), k2c = ld4(A.c2 + sc);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, gcv[4] = {gc4.x, gc4.y, gc4.z, gc4.w};
        const float acv[4] = {ac.x, ac.y, ac.z, ac.w}, bcv[4] = {bc.x, bc.y, bc.z, bc.w};
        const float mcv[4] = {muc.x, muc.y, muc.z, muc.w}, icv[4] = {isc.x, isc.y, isc.z, isc.w};
        const float c1c[4] = {k1c.x, k1c.y, k1c.z, k1c.w}, c2c[4] = {k2c.x, k2c.y, k2c.z, k2c.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float u = fmaf(lv[c], acv[c], bcv[c]);
          const float gg = u > 0.0f ? gcv[c] : 0.0f;
          dl[c] += acv[c] * ((gg - c1c[c]) - ((lv[c] - mcv[c]) * icv[c]) * c2c[c]);
        }
      }
      if (!(dbg & 2) && live) *reinterpret_cast<float4*>(dLE + row * ldle + 4 * q) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    }
    if (live) *reinterpret_cast<float4*>(dLE + row * ldle + C + 4 * q) = make_float4(de[0], de[1], de[2], de[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize: one workgroup, stat groups in order (running statistics are sequential state)
// ------------------------------------------------------------------------------------------------
// One launch serves up to kBnJobs independent finalize jobs (blockIdx.y) and splits the channels of a job
// over blocks of kBnCB channels (blockIdx.x): the partial sums of one layer are up to ~1 MB, which a single
// workgroup pulls through one CU's L1 in 5-13 us; channels are independent, so 8-16 CUs share the read.
constexpr int kBnJobs = 32;   // (3.3 KB of kernel arguments)
constexpr int kBnThreads = 256;
// Channels per block: the kernel is a chain of dependent L2 round trips (rows -> LDS -> statistics), so the more
// blocks share the rows of a job the fewer batches each thread walks through -- in principle; measured
// (profiles/archive/r02/r02aj_small_ab.txt) 4 channels per block beat 2 and 1 (646 / 645 / 642 depth maps/s): the fixed
// launch + round-trip latency dominates, not the row batches.
static int bn_channels_per_block() { return 4; }
struct BnJobs {
  pf_bn_job j[kBnJobs];
};

__global__ __launch_bounds__(kBnThreads) void bn_finalize_kernel(BnJobs jobs, int kBnCB) {
  const pf_bn_job& J = jobs.j[blockIdx.y];
  const int c_base = blockIdx.x * kBnCB;
  if (c_base >= J.C) return;
  const int Cl = min(kBnCB, J.C - c_base);
  const int S = J.G / J.groups_per_stat;
  // Phase 1 (parallel): all (stat group, channel) pairs of a round are reduced at once -- one batch of
  // independent loads per round instead of one global round trip per stat group.
  // Phase 2 (sequential in s, from LDS): affine parameters + the running-statistics recurrence.
  __shared__ double2 red[kBnThreads];
  __shared__ double2 stat[kBnThreads];
  const int tid = threadIdx.x;
  const int entries = J.groups_per_stat * J.T;
  const int64_t estride = (int64_t)J.pcols * 2;
  const int Sc = 64 / kBnCB;        // stat groups per round: at most 64 pairs, at least 4 slices each
  float rm = 0.0f, rv = 0.0f;
  const bool track = J.running_mean != nullptr;
  if (track && tid < Cl) {
    rm = J.running_mean[c_base + tid];
    rv = J.running_var[c_base + tid];
  }
  for (int s0 = 0; s0 < S; s0 += Sc) {
    const int ns = min(Sc, S - s0);
    const int P = ns * Cl;
    const int slices = kBnThreads / P;
    double a = 0.0, b = 0.0;
    if (tid < P * slices) {
      const int pair = tid % P, sl = tid / P;
      const int s = s0 + pair / Cl, c = c_base + pair % Cl;
      const double* base = J.partials + ((int64_t)s * entries * J.pcols + J.col0 + c) * 2;
      int e = sl;
      for (; e + 7 * slices < entries; e += 8 * slices) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(base + (int64_t)(e + u * slices) * estride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a += v[u].x;
          b += v[u].y;
        }
      }
      for (; e < entries; e += slices) {
        const double2 v = *reinterpret_cast<const double2*>(base + (int64_t)e * estride);
        a += v.x;
        b += v.y;
      }
    }
    __syncthreads();
    red[tid] = make_double2(a, b);
    __syncthreads();
    if (tid < P) {
      double sum = 0.0, sq = 0.0;
      for (int i = 0; i < slices; ++i) {
        sum += red[i * P + tid].x;
        sq += red[i * P + tid].y;
      }
      const double mean = sum / J.count;
      double var = sq / J.count - mean * mean;
      stat[tid] = make_double2(mean, var < 0.0 ? 0.0 : var);
    }
    __syncthreads();
    if (tid < Cl) {
      const float g_ = J.gamma[c_base + tid], b_ = J.beta[c_base + tid];
      for (int si = 0; si < ns; ++si) {
        const double2 mv = stat[si * Cl + tid];
        const float invstd = (float)(1.0 / sqrt(mv.y + (double)J.eps));
        const float a_ = invstd * g_;
        J.scale[(int64_t)(s0 + si) * J.ld_affine + c_base + tid] = a_;
        J.shift[(int64_t)(s0 + si) * J.ld_affine + c_base + tid] = b_ - (float)mv.x * a_;
        if (J.rows4) {                                     // rows 2 / 3 of the (4, S, ld) rows tensor: mean, invstd
          const int64_t row = J.shift - J.scale;
          J.scale[2 * row + (int64_t)(s0 + si) * J.ld_affine + c_base + tid] = (float)mv.x;
          J.scale[3 * row + (int64_t)(s0 + si) * J.ld_affine + c_base + tid] = invstd;
        }
        if (track) {
          const double unbiased = J.unbias_n > 1.0 ? mv.y * (J.unbias_n / (J.unbias_n - 1.0)) : mv.y;
          rm = (1.0f - J.momentum) * rm + J.momentum * (float)mv.x;
          rv = (1.0f - J.momentum) * rv + J.momentum * (float)unbiased;
        }
      }
    }
  }
  if (track && tid < Cl) {
    J.running_mean[c_base + tid] = rm;
    J.running_var[c_base + tid] = rv;
  }
}

// ------------------------------------------------------------------------------------------------
// flow head
// ------------------------------------------------------------------------------------------------
// LAZY: the last BatchNorm of the MLP is resolved here (pf_bn_resolve.h, consumer side): a block's pixels belong to
// all ratio^2 sub-grids, so every block reduces the statistics rows of all S = G groups x 16 channels (one
// (group, channel) pair per thread, S * 16 <= 256) -- the rows of the persistent GEMM blocks, 32-128 per group.
template <bool LAZY>
__global__ __launch_bounds__(256) void flow_head_kernel(const float* __restrict__ Z, int64_t ldz,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int ld_affine,
                                                        const float* __restrict__ w_out,
                                                        const float* __restrict__ depth_in, int dh, int dw,
                                                        const float* __restrict__ interval_p, int h, int w,
                                                        int ratio,
                                                        float* __restrict__ flow_prob,
                                                        float* __restrict__ depth_out, pf_bn_job J) {
  __shared__ float aff_s[2][LAZY ? 256 : 1];
  __shared__ double red_s[LAZY ? 512 : 1];
  if (LAZY) {
    const int tid = threadIdx.x;
    const int P = J.G * 16;                                  // (group, channel) pairs; groups_per_stat == 1
    int slices = 256 / P;
    if (slices > J.T) slices = J.T;
    const int per = (J.T + slices - 1) / slices;
    const int pair = tid % P, sl = tid / P;
    double a = 0.0, b = 0.0;
    if (sl < slices) {
      const int sg = pair >> 4, c = pair & 15;
      const double2* base = reinterpret_cast<const double2*>(J.partials) + (int64_t)sg * J.T * J.pcols + J.col0 + c;
      const int r0 = sl * per, r1 = min(J.T, r0 + per);
      for (int r = r0; r < r1; r += 16) {
        double2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = base[(int64_t)min(r + u, r1 - 1) * J.pcols];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          a += (r + u < r1) ? v[u].x : 0.0;
          b += (r + u < r1) ? v[u].y : 0.0;
        }
      }
    }
    red_s[2 * tid + 0] = a;
    red_s[2 * tid + 1] = b;
    __syncthreads();
    if (tid < P) {
      double sum = 0.0, sq = 0.0;
      for (int q = 0; q < slices; ++q) {
        sum += red_s[2 * (q * P + tid) + 0];
        sq += red_s[2 * (q * P + tid) + 1];
      }
      const double mean = sum / J.count;
      double var = sq / J.count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float invstd = (float)(1.0 / sqrt(var + (double)J.eps));
      const float a_ = invstd * J.gamma[tid & 15];
      aff_s[0][tid] = a_;
      aff_s[1][tid] = J.beta[tid & 15] - (float)mean * a_;
    }
    __syncthreads();
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  const int hs = h / ratio, ws = w / ratio;
  const int g = (y % ratio) * ratio + (x % ratio);
  const int64_t Ng = (int64_t)5 * hs * ws;
  const int64_t loc0 = (int64_t)(y / ratio) * ws + (x / ratio);
  const float* sc = LAZY ? &aff_s[0][g * 16] : scale + (int64_t)g * ld_affine;
  const float* sh = LAZY ? &aff_s[1][g * 16] : shift + (int64_t)g * ld_affine;
  float f[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const float* z = Z + ((int64_t)g * Ng + (int64_t)d * hs * ws + loc0) * ldz;
    float acc = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 v = ld4(z + 4 * c4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 4 * c4 + u;
        const float a = fmaxf(fmaf(vv[u], sc[c], sh[c]), 0.0f);
        acc = fmaf(w_out[c], a, acc);
      }
    }
    f[d] = -acc;
  }
  float mx = f[0];
#pragma unroll
  for (int d = 1; d < 5; ++d) mx = fmaxf(mx, f[d]);
  float e[5], den = 0.0f;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    e[d] = expf(f[d] - mx);
    den += e[d];
  }
  float flow = 0.0f;
  const float interval = interval_p[0];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const float p = e[d] / den;
    flow_prob[(int64_t)d * h * w + i] = p;
    flow += p * ((float)(d - 2) * interval);
  }
  const float scy = (float)dh / (float)h, scx = (float)dw / (float)w;
  int sy = (int)floorf((float)y * scy);
  int sx = (int)floorf((float)x * scx);
  sy = sy > dh - 1 ? dh - 1 : sy;
  sx = sx > dw - 1 ? dw - 1 : sx;
  depth_out[i] = depth_in[sy * dw + sx] + flow;
}

}  // namespace

extern "C" {

// Blocks per group of the EdgeConv gather passes: one 64-point tile per block while that stays below ~4096
// blocks in total (measured: 1 600 single-tile blocks on the 4 x 25 600-point lattice run the passes 10 % faster
// than 1 024 blocks with 1-2 tiles each -- 552 -> 565 depth maps/s, profiles/archive/r01/r01p_stat_blocks_ab.log), beyond
// that every block the same number of tiles.
int pf_stat_blocks(int G, int Ng) {
  if (G <= 0 || Ng <= 0) return 0;
  const int tiles = (Ng + TILE - 1) / TILE;
#ifndef PF_STAT_CAP
#define PF_STAT_CAP 4096
#endif
  int cap = PF_STAT_CAP / G;
  cap = cap < 32 ? 32 : cap;
  if (tiles <= cap) return tiles;
  const int per = (tiles + cap - 1) / cap;
  return (tiles + per - 1) / per;
}

// Blocks per group of the pointwise GEMM: about two persistent blocks per CU over the whole launch (the direct-A
// kernel keeps W in LDS for a block's lifetime, so a block should own several tiles), every block the same number
// of tiles; a launch with fewer tiles than that gives each tile its own block.
int pf_gemm_blocks(int G, int Ng) {
  if (G <= 0 || Ng <= 0) return 0;
  const int tiles = (Ng + GT - 1) / GT;
#ifndef PF_GEMM_CAP
#define PF_GEMM_CAP 512
#endif
  int cap = PF_GEMM_CAP / G;                          // persistent GEMM blocks over all groups (measured: fewer is slower)
  cap = cap < 16 ? 16 : cap;
  if (tiles <= cap) return tiles;
  const int per = (tiles + cap - 1) / cap;
  return (tiles + per - 1) / per;
}

// A pending BatchNorm whose consumer cannot resolve it in its prologue: the ordinary finalize launch, minus the
// running statistics (those belong to the one pf_bn_finalize_jobs_f32 call the owner of the job makes anyway).
static int bn_materialize_rows(const pf_bn_job* in_bn, hipStream_t s) {
  PF_REQUIRE(in_bn->scale != nullptr && in_bn->shift != nullptr && in_bn->ld_affine >= in_bn->C);
  pf_bn_job j = *in_bn;
  j.running_mean = j.running_var = nullptr;
  return pf_bn_finalize_jobs_f32(&j, 1, s);
}

int pf_pointwise_gemm_f32(const float* X, int x_point_major, int64_t ldx, const float* Wt, float* Y, int64_t ldy,
                          int G, int Ng, int K, int Nc, int Nc_store, const float* in_scale,
                          const float* in_shift, const pf_bn_job* in_bn, int groups_per_stat, double* col_partials,
                          void* stream) {
  PF_REQUIRE(in_bn == nullptr || in_scale == nullptr);
  if (in_bn != nullptr && G > 0 && Ng > 0) {
    PF_REQUIRE(groups_per_stat >= 1 && G % groups_per_stat == 0);
    const int rc = pf_bn_in_check(in_bn, K, G / groups_per_stat);
    if (rc != PF_OK && rc != PF_ERR_UNSUPPORTED) return rc;
    const int kj = (K + 7) / 8, nt = Nc / 32;
    const bool shape = (kj == 17 && nt == 2) || (kj == 4 && nt == 2) || (kj == 8 && nt == 4) || (kj == 28 && nt == 2) ||
                       (kj == 8 && nt == 2) || (kj == 8 && nt == 1);
    const bool direct = rc == PF_OK && shape && x_point_major && (K % 4) == 0 && (ldx % 4) == 0 &&
                        kj * 8 * Nc <= 16384 && (reinterpret_cast<uintptr_t>(X) % 16) == 0 && K <= 256;
    if (!direct) {
      const int rc2 = bn_materialize_rows(in_bn, (hipStream_t)stream);
      if (rc2 != PF_OK) return rc2;
      PF_REQUIRE(in_bn->ld_affine == K);
      return pf_pointwise_gemm_f32(X, x_point_major, ldx, Wt, Y, ldy, G, Ng, K, Nc, Nc_store, in_bn->scale,
                                   in_bn->shift, nullptr, groups_per_stat, col_partials, stream);
    }
  }
  PF_REQUIRE(G >= 0 && Ng >= 0 && K >= 1 && Nc >= 32 && Nc_store >= 1 && Nc_store <= Nc);
  PF_REQUIRE(Nc % 32 == 0 && groups_per_stat >= 1);
  if (Nc != 32 && Nc != 64 && Nc != 128) return PF_ERR_UNSUPPORTED;
  PF_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
  PF_REQUIRE(G <= 65535);
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(X && Wt && Y && ldy >= Nc_store);
  if (x_point_major) PF_REQUIRE(ldx >= K);
  const int T = pf_gemm_blocks(G, Ng);
  dim3 grid((unsigned)T, (unsigned)G);
  hipStream_t s = (hipStream_t)stream;
  // the PointFlow chain's shapes (point-major rows, K a multiple of 4 that fits LDS with W): direct-A kernel
  {
    const pf_bn_job no_bn = {};
    const int kj = (K + 7) / 8, nt = Nc / 32;
    const bool direct = x_point_major && (K % 4) == 0 && (ldx % 4) == 0 && kj * 8 * Nc <= 16384 &&
                        (reinterpret_cast<uintptr_t>(X) % 16) == 0;
#define PF_GEMM_DIRECT(KJV, NTV)                                                                                   \
  if (direct && kj == KJV && nt == NTV) {                                                                          \
    if (in_bn != nullptr)                                                                                          \
      hipLaunchKernelGGL((pointwise_gemm_direct_kernel<KJV, NTV, 2>), grid, dim3(256), 0, s, X, ldx, Wt, Y, ldy,   \
                         Ng, K, Nc_store, in_scale, in_shift, groups_per_stat, col_partials, T, *in_bn);           \
    else if (in_scale != nullptr)                                                                                  \
      hipLaunchKernelGGL((pointwise_gemm_direct_kernel<KJV, NTV, 1>), grid, dim3(256), 0, s, X, ldx, Wt, Y, ldy,   \
                         Ng, K, Nc_store, in_scale, in_shift, groups_per_stat, col_partials, T, no_bn);            \
    else                                                                                                           \
      hipLaunchKernelGGL((pointwise_gemm_direct_kernel<KJV, NTV, 0>), grid, dim3(256), 0, s, X, ldx, Wt, Y, ldy,   \
                         Ng, K, Nc_store, in_scale, in_shift, groups_per_stat, col_partials, T, no_bn);            \
    return pf_launch_status();                                                                                     \
  }
    PF_GEMM_DIRECT(17, 2)   // EdgeConvNoC 136 -> [32 | 32]
    PF_GEMM_DIRECT(4, 2)    // EdgeConv 32 -> [32 | 32]
    PF_GEMM_DIRECT(8, 4)    // EdgeConv 64 -> [64 | 64]
    PF_GEMM_DIRECT(28, 2)   // MLP 224 -> 64
    PF_GEMM_DIRECT(8, 2)    // MLP 64 -> 64
    PF_GEMM_DIRECT(8, 1)    // MLP 64 -> 16 (padded to 32 columns)
#undef PF_GEMM_DIRECT
  }
#define PF_GEMM_LAUNCH(PM, NTV)                                                                                  \
  hipLaunchKernelGGL((pointwise_gemm_kernel<PM, NTV>), grid, dim3(256), 0, s, X, ldx, Wt, Y, ldy, Ng, K, Nc_store, \
                     in_scale, in_shift, groups_per_stat, col_partials, T)
  if (x_point_major) {
    if (Nc == 32) PF_GEMM_LAUNCH(true, 1);
    else if (Nc == 64) PF_GEMM_LAUNCH(true, 2);
    else PF_GEMM_LAUNCH(true, 4);
  } else {
    if (Nc == 32) PF_GEMM_LAUNCH(false, 1);
    else if (Nc == 64) PF_GEMM_LAUNCH(false, 2);
    else PF_GEMM_LAUNCH(false, 4);
  }
#undef PF_GEMM_LAUNCH
  return pf_launch_status();
}

// `codes` non-NULL selects the lattice form of the neighbourhood (see pf_edge_stats_f32 in the header).
static int check_lattice(const int64_t* idx, const uint8_t* codes, int k, int Ng, int lat_ks, int lat_h, int lat_w,
                         Lattice& lat) {
  lat.ks = 0;
  lat.H = lat.W = 1;
  lat.band = 0;
  if (codes == nullptr) {
    PF_REQUIRE(idx != nullptr);
    // no window codes, but the caller may still name the lattice plane (lat_ks 0, lat_h x lat_w): a hint for the
    // XCD-aware tile order below, ignored when it does not divide the group
    if (lat_ks == 0 && lat_h >= 1 && lat_w >= 1 && (int64_t)lat_h * lat_w > 1 && Ng % ((int64_t)lat_h * lat_w) == 0) {
      lat.H = lat_h;
      lat.W = lat_w;
    }
    return PF_OK;
  }
  PF_REQUIRE(k == 16 && (lat_ks == 3 || lat_ks == 5) && lat_h >= 1 && lat_w >= 1);
  PF_REQUIRE(Ng % (lat_h * lat_w) == 0);
  lat.ks = lat_ks;
  lat.H = lat_h;
  lat.W = lat_w;
  return PF_OK;
}

// XCD-aware tile order (Lattice::band) where the launch allows it: whole tiles per plane, a multiple of 8 of them, one
// tile per block.  PF_EDGE_XCD=0 keeps tile = block (tools: the A/B arm).
static int xcd_band(int64_t plane, int unit, int64_t Ng, int64_t blocks) {      // units of `unit` points per block
  static const bool on = !(getenv("PF_EDGE_XCD") && atoi(getenv("PF_EDGE_XCD")) == 0);
  if (!on || plane <= 1 || plane % unit != 0 || (plane / unit) % 8 != 0 || Ng % plane != 0 || blocks != Ng / unit) return 0;
  return (int)(plane / unit / 8);
}
static void lattice_band(Lattice& lat, int Ng, int T) { lat.band = xcd_band((int64_t)lat.H * lat.W, TILE, Ng, T); }

int pf_edge_stats_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                      double* partials, const uint8_t* codes, int lat_ks, int lat_h, int lat_w, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1 && ldle >= 2 * (int64_t)C && (ldle % 4) == 0 && G <= 65535);
  if (C != 32 && C != 64 && C != 128) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(LE && partials);
  Lattice lat;
  {
    const int rc = check_lattice(idx, codes, k, Ng, lat_ks, lat_h, lat_w, lat);
    if (rc != PF_OK) return rc;
  }
  unsigned* status = pf_status_ptr();
  PF_REQUIRE(status != nullptr);
  const int T = pf_stat_blocks(G, Ng);
  lattice_band(lat, Ng, T);
  dim3 grid((unsigned)T, (unsigned)G);
  hipStream_t s = (hipStream_t)stream;
#define PF_ES(CV, KV) hipLaunchKernelGGL((edge_stats_kernel<CV, KV>), grid, dim3(256), 0, s, LE, ldle, idx, k, Ng, partials, T, status, codes, lat)
  if (k == 16) {
    if (C == 32) PF_ES(32, 16); else if (C == 64) PF_ES(64, 16); else PF_ES(128, 16);
  } else {
    if (C == 32) PF_ES(32, 0); else if (C == 64) PF_ES(64, 0); else PF_ES(128, 0);
  }
#undef PF_ES
  return pf_launch_status();
}

int pf_bn_finalize_jobs_f32(const pf_bn_job* jobs, int njobs, void* stream) {
  PF_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= kBnJobs);
  BnJobs packed;
  int cmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const pf_bn_job& j = jobs[i];
    PF_REQUIRE(j.T >= 1 && j.pcols >= 1 && j.col0 >= 0 && j.C >= 1 && j.col0 + j.C <= j.pcols && j.count > 0.0);
    PF_REQUIRE(j.G >= 1 && j.groups_per_stat >= 1 && (j.G % j.groups_per_stat) == 0 && j.ld_affine >= j.C);
    PF_REQUIRE(j.partials && j.gamma && j.beta && j.scale && j.shift);
    PF_REQUIRE((j.running_mean == nullptr) == (j.running_var == nullptr));
    packed.j[i] = j;
    cmax = j.C > cmax ? j.C : cmax;
  }
  for (int i = njobs; i < kBnJobs; ++i) packed.j[i] = jobs[0];   // never addressed (gridDim.y == njobs)
  const int cb = bn_channels_per_block();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pf_cdiv(cmax, cb), (unsigned)njobs), dim3(kBnThreads), 0,
                     (hipStream_t)stream, packed, cb);
  return pf_launch_status();
}

int pf_bn_finalize_f32(const double* partials, int T, int pcols, int col0, int C, double count, double unbias_n,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int G, int groups_per_stat, float* scale, float* shift,
                       int ld_affine, void* stream) {
  pf_bn_job j;
  j.partials = partials;
  j.T = T;
  j.pcols = pcols;
  j.col0 = col0;
  j.C = C;
  j.count = count;
  j.unbias_n = unbias_n;
  j.gamma = gamma;
  j.beta = beta;
  j.running_mean = running_mean;
  j.running_var = running_var;
  j.momentum = momentum;
  j.eps = eps;
  j.G = G;
  j.groups_per_stat = groups_per_stat;
  j.scale = scale;
  j.shift = shift;
  j.ld_affine = ld_affine;
  j.rows4 = 0;
  return pf_bn_finalize_jobs_f32(&j, 1, stream);
}

int pf_edge_apply_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                      const float* scale, const float* shift, int ld_affine, int groups_per_stat, int concat,
                      float* Y, int64_t ldy, const uint8_t* codes, int lat_ks, int lat_h, int lat_w, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1 && ldle >= 2 * (int64_t)C && (ldle % 4) == 0 && G <= 65535);
  PF_REQUIRE(groups_per_stat >= 1 && (ldy % 4) == 0 && ldy >= (concat ? 2 : 1) * (int64_t)C);
  PF_REQUIRE((ld_affine % 4) == 0 && ld_affine >= (concat ? 2 : 1) * C);
  if (C != 32 && C != 64 && C != 128) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(LE && scale && shift && Y);
  Lattice lat;
  {
    const int rc = check_lattice(idx, codes, k, Ng, lat_ks, lat_h, lat_w, lat);
    if (rc != PF_OK) return rc;
  }
  const int T = pf_stat_blocks(G, Ng);
  lattice_band(lat, Ng, T);
  dim3 grid((unsigned)T, (unsigned)G);
  hipStream_t s = (hipStream_t)stream;
#define PF_EA(CV, KV)                                                                                         \
  hipLaunchKernelGGL((edge_apply_kernel<CV, KV>), grid, dim3(256), 0, s, LE, ldle, idx, k, Ng, scale, shift, \
                     ld_affine, groups_per_stat, concat, Y, ldy, T, codes, lat)
  if (k == 16) {
    if (C == 32) PF_EA(32, 16); else if (C == 64) PF_EA(64, 16); else PF_EA(128, 16);
  } else {
    if (C == 32) PF_EA(32, 0); else if (C == 64) PF_EA(64, 0); else PF_EA(128, 0);
  }
#undef PF_EA
  return pf_launch_status();
}

extern "C++" {
static int edge_backward_reduce(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                                const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int ld_affine, int groups_per_stat,
                                int concat, double* partials, float* sums, float* gacc, int64_t lda, int plane,
                                void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1 && ldle >= 2 * (int64_t)C && (ldle % 4) == 0 && G <= 65535);
  PF_REQUIRE(gacc == nullptr || (sums != nullptr && (lda % 4) == 0 && lda >= (concat ? 2 : 1) * (int64_t)C));
  PF_REQUIRE(groups_per_stat >= 1 && (ldg % 4) == 0 && ldg >= (concat ? 2 : 1) * (int64_t)C);
  PF_REQUIRE((ld_affine % 4) == 0 && ld_affine >= (concat ? 2 : 1) * C);
  if (C != 32 && C != 64) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(LE && idx && grad_y && scale && shift && mean && invstd && partials);
  unsigned* status = pf_status_ptr();
  PF_REQUIRE(status != nullptr);
  const EdgeBwdAffine A{scale, shift, mean, invstd, nullptr, nullptr, ld_affine, groups_per_stat, concat};
  const int T = pf_stat_blocks(G, Ng);
  const int band = xcd_band(plane, TILE, Ng, T);
  dim3 grid((unsigned)T, (unsigned)G);
  hipStream_t s = (hipStream_t)stream;
#define PF_EBR(CV, KV, SV) hipLaunchKernelGGL((edge_bwd_reduce_kernel<CV, KV, SV>), grid, dim3(256), 0, s, LE, ldle, idx, k, Ng, grad_y, ldg, A, partials, T, status, sums, gacc, lda, band)
  if (sums != nullptr) {
    if (k == 16) {
      if (C == 32) PF_EBR(32, 16, true); else PF_EBR(64, 16, true);
    } else {
      if (C == 32) PF_EBR(32, 0, true); else PF_EBR(64, 0, true);
    }
  } else if (k == 16) {
    if (C == 32) PF_EBR(32, 16, false); else PF_EBR(64, 16, false);
  } else {
    if (C == 32) PF_EBR(32, 0, false); else PF_EBR(64, 0, false);
  }
#undef PF_EBR
  return pf_launch_status();
}
}  // extern "C++"

int pf_edge_backward_reduce_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                                const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int ld_affine, int groups_per_stat,
                                int concat, double* partials, void* stream) {
  return edge_backward_reduce(LE, ldle, C, idx, k, G, Ng, grad_y, ldg, scale, shift, mean, invstd, ld_affine,
                              groups_per_stat, concat, partials, nullptr, nullptr, 0, 0, stream);
}

int pf_edge_backward_sums_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                              const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                              const float* mean, const float* invstd, int ld_affine, int groups_per_stat, int concat,
                              double* partials, float* grad_le, float* grad_acc, int64_t ld_acc, int plane,
                              void* stream) {
  PF_REQUIRE(grad_le != nullptr || G == 0 || Ng == 0);
  PF_REQUIRE(plane >= 0);
  return edge_backward_reduce(LE, ldle, C, idx, k, G, Ng, grad_y, ldg, scale, shift, mean, invstd, ld_affine,
                              groups_per_stat, concat, partials, grad_le, grad_acc, ld_acc, plane, stream);
}

extern "C++" {
template <bool FINISH>
static void edge_backward_inverse_launch(const float* LE, int64_t ldle, int C, int k, int G, int Ng, const float* grad_y,
                                         int64_t ldg, const EdgeBwdAffine& A, const uint32_t* inv_order,
                                         const uint32_t* inv_start, float* grad_le, int plane, hipStream_t s) {
  const int64_t rows = (int64_t)G * Ng;
  const int ppb = 256 / (C / 4);
  int64_t blocks = pf_cdiv(rows, ppb);
  if (blocks > 16384) blocks = 16384;
  const int band = G == 1 ? xcd_band(plane, ppb, Ng, blocks) : 0;
  static const int dbg = getenv("PF_EDGE_FINISH_DBG") ? atoi(getenv("PF_EDGE_FINISH_DBG")) : 0;   // tools only: wrong results
  if (C == 32)
    hipLaunchKernelGGL((edge_bwd_inverse_kernel<32, FINISH>), dim3((unsigned)blocks), dim3(256), 0, s, LE, ldle, k, Ng,
                       grad_y, ldg, A, inv_order, inv_start, grad_le, rows, dbg, band);
  else
    hipLaunchKernelGGL((edge_bwd_inverse_kernel<64, FINISH>), dim3((unsigned)blocks), dim3(256), 0, s, LE, ldle, k, Ng,
                       grad_y, ldg, A, inv_order, inv_start, grad_le, rows, dbg, band);
}
}  // extern "C++"

int pf_edge_backward_apply_f32(const float* LE, int64_t ldle, int C, const int64_t* idx, int k, int G, int Ng,
                               const float* grad_y, int64_t ldg, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* c1, const float* c2,
                               int ld_affine, int groups_per_stat, int concat, float* grad_le,
                               const uint32_t* inv_order, const uint32_t* inv_start, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1 && ldle >= 2 * (int64_t)C && (ldle % 4) == 0 && G <= 65535);
  PF_REQUIRE((inv_order == nullptr) == (inv_start == nullptr));
  PF_REQUIRE(groups_per_stat >= 1 && (ldg % 4) == 0 && ldg >= (concat ? 2 : 1) * (int64_t)C);
  PF_REQUIRE((ld_affine % 4) == 0 && ld_affine >= (concat ? 2 : 1) * C);
  if (C != 32 && C != 64) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(LE && idx && grad_y && scale && shift && mean && invstd && c1 && c2 && grad_le);
  hipStream_t s = (hipStream_t)stream;
  const bool scatter = inv_order == nullptr;
  if (scatter) {     // the atomics accumulate into de: clear it first (the inverse gather writes every row itself)
    const int zrc = pf_zero_async(grad_le, sizeof(float) * (size_t)G * Ng * ldle, s);
    if (zrc != PF_OK) return zrc;
  }
  const EdgeBwdAffine A{scale, shift, mean, invstd, c1, c2, ld_affine, groups_per_stat, concat};
  const int T = pf_stat_blocks(G, Ng);
  dim3 grid((unsigned)T, (unsigned)G);
#define PF_EBA(CV, KV, SV) hipLaunchKernelGGL((edge_bwd_apply_kernel<CV, KV, SV>), grid, dim3(256), 0, s, LE, ldle, idx, k, Ng, grad_y, ldg, A, grad_le, T)
  if (scatter) {
    if (k == 16) {
      if (C == 32) PF_EBA(32, 16, true); else PF_EBA(64, 16, true);
    } else {
      if (C == 32) PF_EBA(32, 0, true); else PF_EBA(64, 0, true);
    }
    return pf_launch_status();
  }
  if (k == 16) {
    if (C == 32) PF_EBA(32, 16, false); else PF_EBA(64, 16, false);
  } else {
    if (C == 32) PF_EBA(32, 0, false); else PF_EBA(64, 0, false);
  }
#undef PF_EBA
  edge_backward_inverse_launch<false>(LE, ldle, C, k, G, Ng, grad_y, ldg, A, inv_order, inv_start, grad_le, 0, s);
  return pf_launch_status();
}

int pf_edge_backward_finish_f32(const float* LE, int64_t ldle, int C, int k, int G, int Ng, const float* grad_y,
                                int64_t ldg, const float* scale, const float* shift, const float* mean,
                                const float* invstd, const float* c1, const float* c2, int ld_affine,
                                int groups_per_stat, int concat, float* grad_le, const uint32_t* inv_order,
                                const uint32_t* inv_start, int plane, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1 && ldle >= 2 * (int64_t)C && (ldle % 4) == 0 && G <= 65535 && plane >= 0);
  PF_REQUIRE(groups_per_stat >= 1 && (ldg % 4) == 0 && ldg >= (concat ? 2 : 1) * (int64_t)C);
  PF_REQUIRE((ld_affine % 4) == 0 && ld_affine >= (concat ? 2 : 1) * C);
  if (C != 32 && C != 64) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(LE && grad_y && scale && shift && mean && invstd && c1 && c2 && grad_le && inv_order && inv_start);
  const EdgeBwdAffine A{scale, shift, mean, invstd, c1, c2, ld_affine, groups_per_stat, concat};
  edge_backward_inverse_launch<true>(LE, ldle, C, k, G, Ng, grad_y, ldg, A, inv_order, inv_start, grad_le, plane,
                                     (hipStream_t)stream);
  return pf_launch_status();
}

int pf_flow_head_f32(const float* Z, int64_t ldz, const float* scale, const float* shift, int ld_affine,
                     const pf_bn_job* in_bn, const float* w_out, const float* depth_in, int dh, int dw,
                     const float* interval, int h, int w, int ratio, float* flow_prob, float* depth_out,
                     void* stream) {
  PF_REQUIRE(h >= 1 && w >= 1 && dh >= 1 && dw >= 1 && ratio >= 1 && h % ratio == 0 && w % ratio == 0);
  PF_REQUIRE(ldz >= 16 && (ldz % 4) == 0 && (int64_t)h * w <= INT32_MAX);
  PF_REQUIRE(Z && w_out && depth_in && interval && flow_prob && depth_out);
  PF_REQUIRE((in_bn != nullptr) != (scale != nullptr) && (scale == nullptr) == (shift == nullptr));
  dim3 grid((unsigned)pf_cdiv((int64_t)h * w, 256));
  hipStream_t s = (hipStream_t)stream;
  if (in_bn != nullptr) {
    const int rc = pf_bn_in_check(in_bn, 16, in_bn->G / (in_bn->groups_per_stat > 0 ? in_bn->groups_per_stat : 1));
    if (rc != PF_OK && rc != PF_ERR_UNSUPPORTED) return rc;
    PF_REQUIRE(in_bn->G == ratio * ratio);
    if (rc == PF_OK && in_bn->groups_per_stat == 1 && in_bn->G * 16 <= 256) {
      hipLaunchKernelGGL(flow_head_kernel<true>, grid, dim3(256), 0, s, Z, ldz, nullptr, nullptr, 0, w_out, depth_in,
                         dh, dw, interval, h, w, ratio, flow_prob, depth_out, *in_bn);
      return pf_launch_status();
    }
    PF_REQUIRE(in_bn->groups_per_stat == 1);     // (the kernel indexes its affine rows by sub-grid)
    const int rc2 = bn_materialize_rows(in_bn, s);
    if (rc2 != PF_OK) return rc2;
    scale = in_bn->scale;
    shift = in_bn->shift;
    ld_affine = in_bn->ld_affine;
  }
  PF_REQUIRE(ld_affine >= 16);
  const pf_bn_job no_bn = {};
  hipLaunchKernelGGL(flow_head_kernel<false>, grid, dim3(256), 0, s, Z, ldz, scale, shift, ld_affine, w_out, depth_in,
                     dh, dw, interval, h, w, ratio, flow_prob, depth_out, no_bn);
  return pf_launch_status();
}

}  // extern "C"
