// Row Z (training step, BASELINE config 4): train-mode BatchNorm (+ReLU) BACKWARD for the conv stacks and the flow
// MLP, and the forward finalize that keeps what the backward needs.
//
// The reference's step (train.py:72-82) differentiates  conv -> BatchNorm(batch statistics) -> ReLU  blocks
// (nn/conv.py:24-35, :62-77, :108-121, :197-210) through ATen's batch_norm backward.  Here, per block with raw
// convolution output y, z = relu(y * scale + shift), upstream gradient g = dL/dz:
//     g'     = [y * scale + shift > 0] * g                      (g' = g for a block without ReLU)
//     dbeta  = sum g'            dgamma = sum g' * xhat         xhat = (y - mean) * invstd
//     dy     = scale * (g' - dbeta / M - xhat * dgamma / M)     M = elements behind the statistic
// as three launches: `reduce` streams (g, y) once into float64 per-block partials (fixed order: bit-reproducible),
// `coeffs` folds them into per-(statistic group, channel) constants and dgamma / dbeta, `apply` streams (g, y) -> dy.
// Planar tensors are (N, C, S) like norm.hip; the MLP's are point-major rows (P, ld).  All HBM-bound streams:
// reduce 8 bytes per element, apply 12.
#include "pf_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// forward finalize that also keeps (mean, invstd): rows (4, S, C) = [scale | shift | mean | invstd] (each an (S, C) block, the layout the
// forward kernels' in_scale / in_shift rows have)
// ------------------------------------------------------------------------------------------------
// One block per 4 channels; the S statistic groups in order (the running statistics are sequential state, one update
// per group like S successive nn.BatchNorm calls).  Thread (channel c = tid & 3, slice tid >> 2) sums every 64th
// statistics row; the 64 slices are added in slice order.  Same arithmetic as bn_finalize_kernel (edgeconv.hip).
__global__ __launch_bounds__(256) void bn_train_rows_kernel(const double* __restrict__ partials, int T, int pcols,
                                                            int col0, int C, double count, double unbias_n,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float momentum, float eps,
                                                            int S, int gps, float* __restrict__ rows, int ld,
                                                            int col_out, int ch0) {
  __shared__ double2 red[256];
  const int tid = threadIdx.x;
  const int cl = tid & 3, sl = tid >> 2;
  const int c = blockIdx.x * 4 + cl;
  const bool live = c < C;
  const int entries = gps * T;
  const bool track = running_mean != nullptr;
  float rm = 0.0f, rv = 0.0f;
  if (track && sl == 0 && live) {
    rm = running_mean[ch0 + c];
    rv = running_var[ch0 + c];
  }
  for (int s = 0; s < S; ++s) {
    double a = 0.0, b = 0.0;
    if (live) {
      const double* base = partials + ((int64_t)s * entries * pcols + col0 + c) * 2;
      for (int e = sl; e < entries; e += 64) {
        const double2 v = *reinterpret_cast<const double2*>(base + (int64_t)e * pcols * 2);
        a += v.x;
        b += v.y;
      }
    }
    red[tid] = make_double2(a, b);
    __syncthreads();
    if (sl == 0 && live) {
      double sum = 0.0, sq = 0.0;
      for (int i = 0; i < 64; ++i) {
        sum += red[i * 4 + cl].x;
        sq += red[i * 4 + cl].y;
      }
      const double mean = sum / count;
      double var = sq / count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = invstd * gamma[ch0 + c];
      const int64_t SC = (int64_t)S * ld;
      float* r = rows + (int64_t)s * ld + col_out + c;
      r[0] = sc;
      r[SC] = beta[ch0 + c] - (float)mean * sc;
      r[2 * SC] = (float)mean;
      r[3 * SC] = invstd;
      if (track) {
        const double unbiased = unbias_n > 1.0 ? var * (unbias_n / (unbias_n - 1.0)) : var;
        rm = (1.0f - momentum) * rm + momentum * (float)mean;
        rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
      }
    }
    __syncthreads();
  }
  if (track && sl == 0 && live) {
    running_mean[ch0 + c] = rm;
    running_var[ch0 + c] = rv;
  }
}

// ------------------------------------------------------------------------------------------------
// planar (N, C, S): reduce / apply
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float masked(float g, float y, float sc, float sh, int relu) {
  return (!relu || fmaf(y, sc, sh) > 0.0f) ? g : 0.0f;
}

// grid = (T, C, N); block t of (n, c) reduces elements [t*chunk, (t+1)*chunk) -> partials (N, T, C, 2).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ rows, int C, int64_t S,
                                                            int64_t chunk, int sps, int relu,
                                                            double* __restrict__ partials, int T) {
  __shared__ double red[2 * 4];
  const int t = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int64_t SC = (int64_t)(gridDim.z / sps) * C;
  const float* r = rows + (int64_t)(n / sps) * C + c;
  const float sc = r[0], sh = r[SC], mean = r[2 * SC], invstd = r[3 * SC];
  const float* pg = g + ((int64_t)n * C + c) * S;
  const float* py = y + ((int64_t)n * C + c) * S;
  const int64_t lo = (int64_t)t * chunk;
  const int64_t hi = min(S, lo + chunk);
  float s0 = 0.0f, s1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
  auto one = [&](float gv, float yv, float& s, float& q) {
    const float m = masked(gv, yv, sc, sh, relu);
    s += m;
    q += m * ((yv - mean) * invstd);
  };
  if ((((uintptr_t)(pg + lo) | (uintptr_t)(py + lo)) & 15) == 0) {
    const int64_t n4 = (hi - lo) >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(pg + lo);
    const float4* y4 = reinterpret_cast<const float4*>(py + lo);
    int64_t i = threadIdx.x;
    for (; i + 256 < n4; i += 512) {
      const float4 a = g4[i], b = y4[i], a2 = g4[i + 256], b2 = y4[i + 256];
      one(a.x, b.x, s0, q0);
      one(a.y, b.y, s0, q0);
      one(a.z, b.z, s0, q0);
      one(a.w, b.w, s0, q0);
      one(a2.x, b2.x, s1, q1);
      one(a2.y, b2.y, s1, q1);
      one(a2.z, b2.z, s1, q1);
      one(a2.w, b2.w, s1, q1);
    }
    for (; i < n4; i += 256) {
      const float4 a = g4[i], b = y4[i];
      one(a.x, b.x, s0, q0);
      one(a.y, b.y, s0, q0);
      one(a.z, b.z, s0, q0);
      one(a.w, b.w, s0, q0);
    }
    for (int64_t j = lo + (n4 << 2) + threadIdx.x; j < hi; j += 256) one(pg[j], py[j], s0, q0);
  } else {
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) one(pg[j], py[j], s0, q0);
  }
  double ds = (double)s0 + (double)s1;
  double dq = (double)q0 + (double)q1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ds += __shfl_xor(ds, off);
    dq += __shfl_xor(dq, off);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave * 2 + 0] = ds;
    red[wave * 2 + 1] = dq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partials + (((int64_t)n * T + t) * C + c) * 2;
    o[0] = (red[0] + red[2]) + (red[4] + red[6]);
    o[1] = (red[1] + red[3]) + (red[5] + red[7]);
  }
}

// coef rows (2, S, C): k1 = scale * dbeta_s / M, k2 = scale * invstd * dgamma_s / M, so that
//   dy = scale * g' - k1 - k2 * (y - mean);  dgamma[c] / dbeta[c] = sums over the S statistic groups, in order.
// One block per 4 channels, as bn_train_rows_kernel.
__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(const double* __restrict__ partials, int T, int pcols,
                                                            int col0, int C, double count, int S, int gps,
                                                            const float* __restrict__ rows, float* __restrict__ coef,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int accumulate) {
  __shared__ double2 red[256];
  const int tid = threadIdx.x;
  const int cl = tid & 3, sl = tid >> 2;
  const int c = blockIdx.x * 4 + cl;
  const bool live = c < C;
  const int entries = gps * T;
  double tg = 0.0, tb = 0.0;
  for (int s = 0; s < S; ++s) {
    double a = 0.0, b = 0.0;
    if (live) {
      const double* base = partials + ((int64_t)s * entries * pcols + col0 + c) * 2;
      for (int e = sl; e < entries; e += 64) {
        const double2 v = *reinterpret_cast<const double2*>(base + (int64_t)e * pcols * 2);
        a += v.x;
        b += v.y;
      }
    }
    red[tid] = make_double2(a, b);
    __syncthreads();
    if (sl == 0 && live) {
      double sb = 0.0, sg = 0.0;
      for (int i = 0; i < 64; ++i) {
        sb += red[i * 4 + cl].x;
        sg += red[i * 4 + cl].y;
      }
      const int64_t SC = (int64_t)S * C;
      const float* r = rows + (int64_t)s * C + c;
      const double sc = (double)r[0], invstd = (double)r[3 * SC];
      coef[(int64_t)s * C + c] = (float)(sc * sb / count);
      coef[SC + (int64_t)s * C + c] = (float)(sc * invstd * sg / count);
      tb += sb;
      tg += sg;
    }
    __syncthreads();
  }
  if (sl == 0 && live) {
    if (dgamma != nullptr) dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)tg;
    if (dbeta != nullptr) dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)tb;
  }
}

// grid = (blocks, C, N): dy = scale * g' - k1 - k2 * (y - mean)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ rows,
                                                           const float* __restrict__ coef, float* __restrict__ dy,
                                                           int C, int64_t S, int sps, int relu) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int s = n / sps;
  const int64_t SC = (int64_t)(gridDim.z / sps) * C;
  const float* r = rows + (int64_t)s * C + c;
  const float sc = r[0], sh = r[SC], mean = r[2 * SC];
  const float k1 = coef[(int64_t)s * C + c], k2 = coef[SC + (int64_t)s * C + c];
  const float* pg = g + ((int64_t)n * C + c) * S;
  const float* py = y + ((int64_t)n * C + c) * S;
  float* po = dy + ((int64_t)n * C + c) * S;
  auto one = [&](float gv, float yv) {
    const float m = masked(gv, yv, sc, sh, relu);
    return fmaf(-k2, yv - mean, fmaf(sc, m, -k1));
  };
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)pg | (uintptr_t)py | (uintptr_t)po) & 15) == 0 && (S & 3) == 0) {
    const int64_t n4 = S >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(pg);
    const float4* y4 = reinterpret_cast<const float4*>(py);
    float4* o4 = reinterpret_cast<float4*>(po);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 a = g4[i], b = y4[i];
      float4 v;
      v.x = one(a.x, b.x);
      v.y = one(a.y, b.y);
      v.z = one(a.z, b.z);
      v.w = one(a.w, b.w);
      o4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += stride) po[i] = one(pg[i], py[i]);
  }
}

// The coefficients and the apply pass in ONE launch: every block of (n, c) first adds its statistic group's partial
// rows exactly as bn_bwd_coeffs_kernel does (64 slices, slice order: the same bits), then streams its elements.  Block
// (0, c, 0) also writes dgamma[c] / dbeta[c] (the sums over all S groups).  entries = sps * T partial rows per group is
// a few dozen doubles: cheaper than a launch in a chain of ~700.
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                 const float* __restrict__ rows,
                                                                 const double* __restrict__ partials, int T,
                                                                 double count, float* __restrict__ dy, int C, int64_t S,
                                                                 int sps, int relu, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, int accumulate) {
  __shared__ double2 red[64];
  __shared__ double2 tot;
  const int c = blockIdx.y, n = blockIdx.z;
  const int s = n / sps;
  const int groups = gridDim.z / sps;
  const int64_t SC = (int64_t)groups * C;
  const int entries = sps * T;
  const bool writer = blockIdx.x == 0 && n == 0 && (dgamma != nullptr || dbeta != nullptr);
  double tb = 0.0, tg = 0.0, sb_own = 0.0, sg_own = 0.0;
  for (int q = writer ? 0 : s; q < (writer ? groups : s + 1); ++q) {
    if (threadIdx.x < 64) {
      double a = 0.0, b = 0.0;
      const double* base = partials + ((int64_t)q * entries * C + c) * 2;
      for (int e = threadIdx.x; e < entries; e += 64) {
        const double2 v = *reinterpret_cast<const double2*>(base + (int64_t)e * C * 2);
        a += v.x;
        b += v.y;
      }
      red[threadIdx.x] = make_double2(a, b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sb = 0.0, sg = 0.0;
      for (int i = 0; i < 64; ++i) {
        sb += red[i].x;
        sg += red[i].y;
      }
      tb += sb;
      tg += sg;
      if (q == s) {
        sb_own = sb;
        sg_own = sg;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tot = make_double2(sb_own, sg_own);
    if (writer) {
      if (dgamma != nullptr) dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)tg;
      if (dbeta != nullptr) dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)tb;
    }
  }
  __syncthreads();
  const float* r = rows + (int64_t)s * C + c;
  const float sc = r[0], sh = r[SC], mean = r[2 * SC];
  const double scd = (double)sc, invstd = (double)r[3 * SC];
  const float k1 = (float)(scd * tot.x / count), k2 = (float)(scd * invstd * tot.y / count);
  const float* pg = g + ((int64_t)n * C + c) * S;
  const float* py = y + ((int64_t)n * C + c) * S;
  float* po = dy + ((int64_t)n * C + c) * S;
  auto one = [&](float gv, float yv) {
    const float m = masked(gv, yv, sc, sh, relu);
    return fmaf(-k2, yv - mean, fmaf(sc, m, -k1));
  };
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)pg | (uintptr_t)py | (uintptr_t)po) & 15) == 0 && (S & 3) == 0) {
    const int64_t n4 = S >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(pg);
    const float4* y4 = reinterpret_cast<const float4*>(py);
    float4* o4 = reinterpret_cast<float4*>(po);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 a = g4[i], b = y4[i];
      float4 v;
      v.x = one(a.x, b.x);
      v.y = one(a.y, b.y);
      v.z = one(a.z, b.z);
      v.w = one(a.w, b.w);
      o4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += stride) po[i] = one(pg[i], py[i]);
  }
}

// ------------------------------------------------------------------------------------------------
// planar, ONE pass (round 6): a plane that fits the registers of one block
// ------------------------------------------------------------------------------------------------
// The config-4 step runs 18 BatchNorm backwards over planes of 480 .. 30 720 elements (the towers' 32- and 64-channel
// layers, VolumeConv below 48x64x80): `reduce` + `apply_fused` are two launches of 5-8 us each that read (g, y) twice.  Here
// one block of 1024 threads owns a CHANNEL and walks its N samples (each sample its own statistic group: samples_per_stat
// 1): a plane's (g, y) is loaded ONCE into registers (KQ 16-byte pieces of each per thread), summed (float per thread,
// float64 across threads, fixed order), turned into dy from the registers and stored; dgamma / dbeta are the sums over the
// samples in sample order.  12 bytes per element instead of 20, one launch instead of two.
template <int KQ>
__global__ __launch_bounds__(1024) void bn_bwd_plane_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                            const float* __restrict__ rows, int N, int C, int S,
                                                            int relu, float* __restrict__ dy,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int accumulate) {
  __shared__ double2 red[16];
  __shared__ double2 tot;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int n4 = S >> 2;
  const int64_t SC = (int64_t)N * C;
  double tb = 0.0, tg = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* r = rows + (int64_t)n * C + c;
    const float sc = r[0], sh = r[SC], mean = r[2 * SC], invstd = r[3 * SC];
    const float4* g4 = reinterpret_cast<const float4*>(g + ((int64_t)n * C + c) * S);
    const float4* y4 = reinterpret_cast<const float4*>(y + ((int64_t)n * C + c) * S);
    float4* o4 = reinterpret_cast<float4*>(dy + ((int64_t)n * C + c) * S);
    float4 gv[KQ], yv[KQ];
#pragma unroll
    for (int u = 0; u < KQ; ++u) {
      const int i = tid + 1024 * u;
      gv[u] = i < n4 ? g4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      yv[u] = i < n4 ? y4[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float s = 0.0f, q = 0.0f;
#pragma unroll
    for (int u = 0; u < KQ; ++u) {
      if (tid + 1024 * u < n4) {
        gv[u].x = masked(gv[u].x, yv[u].x, sc, sh, relu);
        gv[u].y = masked(gv[u].y, yv[u].y, sc, sh, relu);
        gv[u].z = masked(gv[u].z, yv[u].z, sc, sh, relu);
        gv[u].w = masked(gv[u].w, yv[u].w, sc, sh, relu);
        s += gv[u].x;
        q += gv[u].x * ((yv[u].x - mean) * invstd);
        s += gv[u].y;
        q += gv[u].y * ((yv[u].y - mean) * invstd);
        s += gv[u].z;
        q += gv[u].z * ((yv[u].z - mean) * invstd);
        s += gv[u].w;
        q += gv[u].w * ((yv[u].w - mean) * invstd);
      }
    }
    double ds = (double)s, dq = (double)q;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ds += __shfl_xor(ds, off);
      dq += __shfl_xor(dq, off);
    }
    __syncthreads();                                   // (the previous sample's reads of red / tot are done)
    if (lane == 0) red[wave] = make_double2(ds, dq);
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < 16; ++w) {
        a += red[w].x;
        b += red[w].y;
      }
      tot = make_double2(a, b);
    }
    __syncthreads();
    const double sb = tot.x, sg = tot.y;
    tb += sb;
    tg += sg;
    const double scd = (double)sc;
    const float k1 = (float)(scd * sb / (double)S), k2 = (float)(scd * (double)invstd * sg / (double)S);
#pragma unroll
    for (int u = 0; u < KQ; ++u) {
      const int i = tid + 1024 * u;
      if (i < n4) {
        float4 v;
        v.x = fmaf(-k2, yv[u].x - mean, fmaf(sc, gv[u].x, -k1));
        v.y = fmaf(-k2, yv[u].y - mean, fmaf(sc, gv[u].y, -k1));
        v.z = fmaf(-k2, yv[u].z - mean, fmaf(sc, gv[u].z, -k1));
        v.w = fmaf(-k2, yv[u].w - mean, fmaf(sc, gv[u].w, -k1));
        o4[i] = v;
      }
    }
  }
  if (tid == 0) {
    if (dgamma != nullptr) dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)tg;
    if (dbeta != nullptr) dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)tb;
  }
}

// ------------------------------------------------------------------------------------------------
// point-major rows (P, ld): the flow MLP's BatchNorm1d (reference nn/conv.py:24-35 via nn/mlp.py:45-81)
// ------------------------------------------------------------------------------------------------
// G groups of Ng rows; block (t, g) reduces rows [t*chunk, (t+1)*chunk) of group g.  partials (G, T, C, 2).
// Round 6: 64 rows per block (was 256: 100 blocks of serial 4-byte loads took 22 us for the 13 MB of a 25 600-point
// layer) and, where the rows allow 16-byte loads, thread = (4 columns, row phase); else thread = (column, row phase).
constexpr int kRowT = 2048;    // blocks per group (at most)
constexpr int kRowsPerBlock = 64;

__global__ __launch_bounds__(256) void rows_bn_bwd_reduce_kernel(const float* __restrict__ g, int64_t ldg,
                                                                 const float* __restrict__ y, int64_t ldy,
                                                                 const float* __restrict__ rows, int C, int Ng,
                                                                 int chunk, int gps, int relu,
                                                                 double* __restrict__ partials, int T, int vec) {
  __shared__ double2 red[256 * 4];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, grp = blockIdx.y;
  const int64_t SC = (int64_t)(gridDim.y / gps) * C;
  const int lo = t * chunk, hi = min(Ng, lo + chunk);
  if (vec) {
    const int C4 = C >> 2;
    const int cq = tid % C4, ph = tid / C4, nph = 256 / C4;
    const float* r = rows + (int64_t)(grp / gps) * C + 4 * cq;
    const float4 sc = *reinterpret_cast<const float4*>(r), sh = *reinterpret_cast<const float4*>(r + SC);
    const float4 mean = *reinterpret_cast<const float4*>(r + 2 * SC), invstd = *reinterpret_cast<const float4*>(r + 3 * SC);
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    double ds[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
    int cnt = 0;
    for (int m = lo + ph; m < hi; m += nph) {
      const int64_t row = (int64_t)grp * Ng + m;
      const float4 gv = *reinterpret_cast<const float4*>(g + row * ldg + 4 * cq);
      const float4 yv = *reinterpret_cast<const float4*>(y + row * ldy + 4 * cq);
      const float mk0 = masked(gv.x, yv.x, sc.x, sh.x, relu), mk1 = masked(gv.y, yv.y, sc.y, sh.y, relu);
      const float mk2 = masked(gv.z, yv.z, sc.z, sh.z, relu), mk3 = masked(gv.w, yv.w, sc.w, sh.w, relu);
      s[0] += mk0;
      s[1] += mk1;
      s[2] += mk2;
      s[3] += mk3;
      q[0] += mk0 * ((yv.x - mean.x) * invstd.x);
      q[1] += mk1 * ((yv.y - mean.y) * invstd.y);
      q[2] += mk2 * ((yv.z - mean.z) * invstd.z);
      q[3] += mk3 * ((yv.w - mean.w) * invstd.w);
      if (++cnt == 64) {            // float32 over short runs, float64 across them
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ds[j] += (double)s[j];
          dq[j] += (double)q[j];
          s[j] = q[j] = 0.0f;
        }
        cnt = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(ph * C4 + cq) * 4 + j] = make_double2(ds[j] + (double)s[j], dq[j] + (double)q[j]);
    __syncthreads();
    if (tid < C) {                  // column tid: its quad's phases in order
      const int qd = tid >> 2, j = tid & 3;
      double a = 0.0, b = 0.0;
      for (int i = 0; i < nph; ++i) {
        a += red[(i * C4 + qd) * 4 + j].x;
        b += red[(i * C4 + qd) * 4 + j].y;
      }
      double* o = partials + (((int64_t)grp * T + t) * C + tid) * 2;
      o[0] = a;
      o[1] = b;
    }
    return;
  }
  const int c = tid % C, ph = tid / C, nph = 256 / C;
  const float* r = rows + (int64_t)(grp / gps) * C + c;
  const float sc = r[0], sh = r[SC], mean = r[2 * SC], invstd = r[3 * SC];
  double ds = 0.0, dq = 0.0;
  float s = 0.0f, q = 0.0f;
  int cnt = 0;
  for (int m = lo + ph; m < hi; m += nph) {
    const int64_t row = (int64_t)grp * Ng + m;
    const float gv = g[row * ldg + c], yv = y[row * ldy + c];
    const float mk = masked(gv, yv, sc, sh, relu);
    s += mk;
    q += mk * ((yv - mean) * invstd);
    if (++cnt == 64) {            // float32 over short runs, float64 across them
      ds += (double)s;
      dq += (double)q;
      s = q = 0.0f;
      cnt = 0;
    }
  }
  ds += (double)s;
  dq += (double)q;
  red[tid] = make_double2(ds, dq);
  __syncthreads();
  if (ph == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < nph; ++i) {
      a += red[i * C + c].x;
      b += red[i * C + c].y;
    }
    double* o = partials + (((int64_t)grp * T + t) * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// thread = (row, 4 columns): dy rows (P, ldo)
__global__ __launch_bounds__(256) void rows_bn_bwd_apply_kernel(const float* __restrict__ g, int64_t ldg,
                                                                const float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ rows,
                                                                const float* __restrict__ coef,
                                                                float* __restrict__ dy, int64_t ldo, int C, int Ng,
                                                                int64_t P, int gps, int relu) {
  const int q4 = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / q4;
  const int c0 = (int)(i - row * q4) * 4;
  if (row >= P) return;
  const int s = (int)(row / Ng) / gps;
  const int64_t SC = (P / Ng / gps) * C;
  const float* r = rows + (int64_t)s * C + c0;
  const float* k = coef + (int64_t)s * C + c0;
  const float4 gv = *reinterpret_cast<const float4*>(g + row * ldg + c0);
  const float4 yv = *reinterpret_cast<const float4*>(y + row * ldy + c0);
  const float4 sc = *reinterpret_cast<const float4*>(r), sh = *reinterpret_cast<const float4*>(r + SC);
  const float4 mean = *reinterpret_cast<const float4*>(r + 2 * SC);
  const float4 k1 = *reinterpret_cast<const float4*>(k), k2 = *reinterpret_cast<const float4*>(k + SC);
  float4 o;
  o.x = fmaf(-k2.x, yv.x - mean.x, fmaf(sc.x, masked(gv.x, yv.x, sc.x, sh.x, relu), -k1.x));
  o.y = fmaf(-k2.y, yv.y - mean.y, fmaf(sc.y, masked(gv.y, yv.y, sc.y, sh.y, relu), -k1.y));
  o.z = fmaf(-k2.z, yv.z - mean.z, fmaf(sc.z, masked(gv.z, yv.z, sc.z, sh.z, relu), -k1.z));
  o.w = fmaf(-k2.w, yv.w - mean.w, fmaf(sc.w, masked(gv.w, yv.w, sc.w, sh.w, relu), -k1.w));
  *reinterpret_cast<float4*>(dy + row * ldo + c0) = o;
}

// z rows = relu(y * scale + shift) (materialises a normalised activation the training step hands to ATen)
__global__ __launch_bounds__(256) void rows_affine_kernel(const float* __restrict__ y, int64_t ldy,
                                                          const float* __restrict__ rows, float* __restrict__ z,
                                                          int64_t ldz, int C, int Ng, int64_t P, int gps, int relu) {
  const int q4 = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = i / q4;
  const int c0 = (int)(i - row * q4) * 4;
  if (row >= P) return;
  const int s = (int)(row / Ng) / gps;
  const int64_t SC = (P / Ng / gps) * C;
  const float* r = rows + (int64_t)s * C + c0;
  const float4 yv = *reinterpret_cast<const float4*>(y + row * ldy + c0);
  const float4 sc = *reinterpret_cast<const float4*>(r), sh = *reinterpret_cast<const float4*>(r + SC);
  float4 o;
  o.x = fmaf(yv.x, sc.x, sh.x);
  o.y = fmaf(yv.y, sc.y, sh.y);
  o.z = fmaf(yv.z, sc.z, sh.z);
  o.w = fmaf(yv.w, sc.w, sh.w);
  if (relu) {
    o.x = fmaxf(o.x, 0.0f);
    o.y = fmaxf(o.y, 0.0f);
    o.z = fmaxf(o.z, 0.0f);
    o.w = fmaxf(o.w, 0.0f);
  }
  *reinterpret_cast<float4*>(z + row * ldz + c0) = o;
}

// Finalize of pf_edge_backward_reduce_f32's partials (G, T, cbn, 2) = per block (sum g', sum g' * xhat): per statistic
// set s and channel c the sums over the set's groups and blocks in a fixed order (64 slices of the rows, then the slices
// in order), c1 = sum0 / m, c2 = sum1 / m (m = the values per statistic: points for the central half of a concat layer,
// pairs else), dbeta / dgamma = the sums over the sets.  Block = 4 channels x 64 slices.
__global__ __launch_bounds__(256) void edge_bwd_coeffs_kernel(const double* __restrict__ partials, int S, int gps, int T,
                                                              int cbn, int ncentral, double m_points, double m_pairs,
                                                              float* __restrict__ c1, float* __restrict__ c2,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int accumulate) {
  __shared__ double2 red[256];
  const int tid = threadIdx.x;
  const int cl = tid & 3, sl = tid >> 2;
  const int c = blockIdx.x * 4 + cl;
  const bool live = c < cbn;
  const int rows = gps * T;
  const double m = c < ncentral ? m_points : m_pairs;
  double tb = 0.0, tg = 0.0;
  for (int s = 0; s < S; ++s) {
    double a = 0.0, b = 0.0;
    if (live) {
      const double2* p = reinterpret_cast<const double2*>(partials) + (int64_t)s * rows * cbn + c;
      for (int e = sl; e < rows; e += 64) {
        const double2 v = p[(int64_t)e * cbn];
        a += v.x;
        b += v.y;
      }
    }
    red[tid] = make_double2(a, b);
    __syncthreads();
    if (sl == 0 && live) {
      double r0 = 0.0, r1 = 0.0;
      for (int i = 0; i < 64; ++i) {
        r0 += red[i * 4 + cl].x;
        r1 += red[i * 4 + cl].y;
      }
      c1[(int64_t)s * cbn + c] = (float)(r0 / m);
      c2[(int64_t)s * cbn + c] = (float)(r1 / m);
      tb += r0;
      tg += r1;
    }
    __syncthreads();
  }
  if (sl == 0 && live) {
    if (dgamma != nullptr) dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)tg;
    if (dbeta != nullptr) dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)tb;
  }
}

}  // namespace

extern "C" {

int pf_bn_train_rows_f32(const double* partials, int T, int pcols, int col0, int C, double count, double unbias_n,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int G, int groups_per_stat, float* rows, int ld_rows, int col_out,
                         int ch0, void* stream) {
  PF_REQUIRE(T >= 1 && pcols >= 1 && col0 >= 0 && C >= 1 && col0 + C <= pcols && count > 0.0);
  PF_REQUIRE(col_out >= 0 && ld_rows >= col_out + C && ch0 >= 0);
  PF_REQUIRE(G >= 1 && groups_per_stat >= 1 && (G % groups_per_stat) == 0);
  PF_REQUIRE(partials && gamma && beta && rows && (running_mean == nullptr) == (running_var == nullptr));
  hipLaunchKernelGGL(bn_train_rows_kernel, dim3((unsigned)pf_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partials, T,
                     pcols, col0, C, count, unbias_n, gamma, beta, running_mean, running_var, momentum, eps,
                     G / groups_per_stat, groups_per_stat, rows, ld_rows, col_out, ch0);
  return pf_launch_status();
}

int pf_bn_bwd_reduce_f32(const float* g, const float* y, const float* rows, int64_t N, int64_t C, int64_t S,
                         int samples_per_stat, int relu, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && partials && (N % samples_per_stat) == 0);
  const int T = pf_norm_blocks(S);
  int64_t chunk = (S + T - 1) / T;
  chunk = (chunk + 3) & ~(int64_t)3;
  dim3 grid((unsigned)T, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, y, rows, (int)C, S, chunk,
                     samples_per_stat, relu, partials, T);
  return pf_launch_status();
}

int pf_bn_bwd_plane_supported(int64_t S, int samples_per_stat) {
  return samples_per_stat == 1 && S >= 4 && (S & 3) == 0 && S <= 4 * 1024 * 8;
}

int pf_bn_bwd_plane_f32(const float* g, const float* y, const float* rows, int64_t N, int64_t C, int64_t S, int relu,
                        float* dy, float* dgamma, float* dbeta, int accumulate, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && N <= 65535 && C <= 65535);
  if (!pf_bn_bwd_plane_supported(S, 1)) return PF_ERR_UNSUPPORTED;
  if (N == 0 || C == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && dy && (((uintptr_t)g | (uintptr_t)y | (uintptr_t)dy) & 15) == 0);
  const int kq = (int)pf_cdiv(S >> 2, 1024);
  hipStream_t st = (hipStream_t)stream;
#define PF_PLANE(K)                                                                                                  \
  hipLaunchKernelGGL(bn_bwd_plane_kernel<K>, dim3((unsigned)C), dim3(1024), 0, st, g, y, rows, (int)N, (int)C, (int)S, \
                     relu, dy, dgamma, dbeta, accumulate)
  if (kq <= 1) PF_PLANE(1);
  else if (kq <= 2) PF_PLANE(2);
  else if (kq <= 4) PF_PLANE(4);
  else if (kq <= 6) PF_PLANE(6);
  else PF_PLANE(8);
#undef PF_PLANE
  return pf_launch_status();
}

int pf_bn_bwd_coeffs_f32(const double* partials, int T, int pcols, int col0, int C, double count, int G,
                         int groups_per_stat, const float* rows, float* coef, float* dgamma, float* dbeta,
                         int accumulate, void* stream) {
  PF_REQUIRE(T >= 1 && pcols >= 1 && col0 >= 0 && C >= 1 && col0 + C <= pcols && count > 0.0);
  PF_REQUIRE(G >= 1 && groups_per_stat >= 1 && (G % groups_per_stat) == 0 && partials && rows && coef);
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((unsigned)pf_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partials, T,
                     pcols, col0, C, count, G / groups_per_stat, groups_per_stat, rows, coef, dgamma, dbeta, accumulate);
  return pf_launch_status();
}

int pf_bn_bwd_apply_fused_f32(const float* g, const float* y, const float* rows, const double* partials, int T,
                              double count, float* dy, int64_t N, int64_t C, int64_t S, int samples_per_stat, int relu,
                              float* dgamma, float* dbeta, int accumulate, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1 && T >= 1 && count > 0.0);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && partials && dy && (N % samples_per_stat) == 0);
  int64_t blocks = (S / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  dim3 grid((unsigned)blocks, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, y, rows, partials, T, count,
                     dy, (int)C, S, samples_per_stat, relu, dgamma, dbeta, accumulate);
  return pf_launch_status();
}

int pf_edge_backward_coeffs_f32(const double* partials, int G, int T, int cbn, int C, int concat, int groups_per_stat,
                                int Ng, int k, float* c1, float* c2, float* dgamma, float* dbeta, int accumulate,
                                void* stream) {
  PF_REQUIRE(G >= 1 && T >= 1 && cbn >= 1 && C >= 1 && groups_per_stat >= 1 && (G % groups_per_stat) == 0);
  PF_REQUIRE(Ng >= 1 && k >= 1 && partials && c1 && c2 && (concat ? cbn == 2 * C : cbn == C));
  const double points = (double)groups_per_stat * Ng;
  hipLaunchKernelGGL(edge_bwd_coeffs_kernel, dim3((unsigned)pf_cdiv(cbn, 4)), dim3(256), 0, (hipStream_t)stream,
                     partials, G / groups_per_stat, groups_per_stat, T, cbn, concat ? C : 0, points, points * k, c1, c2,
                     dgamma, dbeta, accumulate);
  return pf_launch_status();
}

int pf_bn_bwd_apply_f32(const float* g, const float* y, const float* rows, const float* coef, float* dy, int64_t N,
                        int64_t C, int64_t S, int samples_per_stat, int relu, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && coef && dy);
  int64_t blocks = (S / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  dim3 grid((unsigned)blocks, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, y, rows, coef, dy, (int)C, S,
                     samples_per_stat, relu);
  return pf_launch_status();
}

int pf_rows_bn_blocks(int G, int Ng) {
  if (G <= 0 || Ng <= 0) return 0;
  const int t = (Ng + kRowsPerBlock - 1) / kRowsPerBlock;
  return t > kRowT ? kRowT : t;
}

int pf_rows_bn_bwd_reduce_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* rows, int C, int G,
                              int Ng, int groups_per_stat, int relu, double* partials, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && groups_per_stat >= 1 && ldg >= C && ldy >= C);
  if (C != 16 && C != 32 && C != 64 && C != 128) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && partials && (G % groups_per_stat) == 0);
  const int T = pf_rows_bn_blocks(G, Ng);
  const int chunk = (Ng + T - 1) / T;
  const int vec = ((ldg | ldy) & 3) == 0 && (((uintptr_t)g | (uintptr_t)y | (uintptr_t)rows) & 15) == 0 &&
                  ((int64_t)(G / groups_per_stat) * C & 3) == 0;
  hipLaunchKernelGGL(rows_bn_bwd_reduce_kernel, dim3((unsigned)T, (unsigned)G), dim3(256), 0, (hipStream_t)stream, g, ldg,
                     y, ldy, rows, C, Ng, chunk, groups_per_stat, relu, partials, T, vec);
  return pf_launch_status();
}

int pf_rows_bn_bwd_apply_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, const float* rows,
                             const float* coef, float* dy, int64_t ldo, int C, int G, int Ng, int groups_per_stat,
                             int relu, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && groups_per_stat >= 1 && ldg >= C && ldy >= C && ldo >= C);
  if ((C & 3) != 0 || ((ldg | ldy | ldo) & 3) != 0) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(g && y && rows && coef && dy);
  PF_REQUIRE((((uintptr_t)g | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)rows | (uintptr_t)coef) & 15) == 0);
  const int64_t P = (int64_t)G * Ng;
  const int64_t items = P * (C >> 2);
  hipLaunchKernelGGL(rows_bn_bwd_apply_kernel, dim3((unsigned)pf_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     ldg, y, ldy, rows, coef, dy, ldo, C, Ng, P, groups_per_stat, relu);
  return pf_launch_status();
}

int pf_rows_affine_f32(const float* y, int64_t ldy, const float* rows, float* z, int64_t ldz, int C, int G, int Ng,
                       int groups_per_stat, int relu, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && groups_per_stat >= 1 && ldy >= C && ldz >= C);
  if ((C & 3) != 0 || ((ldy | ldz) & 3) != 0) return PF_ERR_UNSUPPORTED;
  if (G == 0 || Ng == 0) return PF_OK;
  PF_REQUIRE(y && rows && z && (((uintptr_t)y | (uintptr_t)z | (uintptr_t)rows) & 15) == 0);
  const int64_t P = (int64_t)G * Ng;
  const int64_t items = P * (C >> 2);
  hipLaunchKernelGGL(rows_affine_kernel, dim3((unsigned)pf_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy,
                     rows, z, ldz, C, Ng, P, groups_per_stat, relu);
  return pf_launch_status();
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// weight packing for a whole training step in ONE launch
// ------------------------------------------------------------------------------------------------
// Every kernel of the step reads its weights in a layout of its own (MFMA operand order, zero padded; the data
// gradients read them flipped and transposed).  Eagerly that is 3-5 tiny ATen launches per layer and direction
// (flip, permute, zeros, copy): ~350 launches of ~5 us per step.  Each layout is an AFFINE GATHER of the parameter:
//     dst[d_0 .. d_{n-1}] = src[sum_k a_k * sstride_k],  a_k = off_k + sum_i M[k][i] * d_i,  zero unless 0 <= a_k < lim_k
// so a table of descriptors in device memory, built once per model (parameter and buffer addresses are stable across
// optimizer steps), re-packs everything with one kernel at the start of each step -- inside the captured graph.
namespace {
constexpr int kPackDims = 7, kPackSrc = 5;
struct PackDesc {                 // 8-byte aligned, mirrored by pointmvsnet_amd/train_packs.py (numpy structured dtype)
  const float* src;
  float* dst;
  long long total;                // product of dshape
  int ndim, nsrc;
  int dshape[kPackDims];
  int dstride[kPackDims];         // element strides of dst (a pack may fill a column slice of a padded matrix)
  int off[kPackSrc];
  int lim[kPackSrc];
  int sstride[kPackSrc];
  int M[kPackSrc][kPackDims];
  int pad_;
};

__global__ __launch_bounds__(256) void pack_gather_kernel(const PackDesc* __restrict__ table) {
  const PackDesc& P = table[blockIdx.y];
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P.total; i += stride) {
    long long rest = i;
    int d[kPackDims];
#pragma unroll
    for (int k = kPackDims - 1; k >= 0; --k) {
      d[k] = 0;
      if (k < P.ndim) {
        d[k] = (int)(rest % P.dshape[k]);
        rest /= P.dshape[k];
      }
    }
    long long so = 0, doff = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < kPackDims; ++k) doff += (long long)d[k] * P.dstride[k];
#pragma unroll
    for (int s = 0; s < kPackSrc; ++s) {
      if (s < P.nsrc) {
        int a = P.off[s];
#pragma unroll
        for (int k = 0; k < kPackDims; ++k) a += P.M[s][k] * d[k];
        ok = ok && a >= 0 && a < P.lim[s];
        so += (long long)a * P.sstride[s];
      }
    }
    P.dst[doff] = ok ? P.src[so] : 0.0f;
  }
}
}  // namespace

extern "C" {

int pf_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }

int pf_pack_gather_f32(const void* table, int npacks, long long max_total, void* stream) {
  PF_REQUIRE(npacks >= 0 && max_total >= 0);
  if (npacks == 0 || max_total == 0) return PF_OK;
  PF_REQUIRE(table != nullptr && npacks <= 65535);
  long long blocks = (max_total + 255) / 256;
  blocks = blocks > 64 ? 64 : blocks;
  hipLaunchKernelGGL(pack_gather_kernel, dim3((unsigned)blocks, (unsigned)npacks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackDesc*>(table));
  return pf_launch_status();
}

}  // extern "C"
