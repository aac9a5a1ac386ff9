// Train-mode BatchNorm for the conv stacks on either side of the path (ImageConv, VolumeConv: SURVEY.md
// section 8(f) items 1-2): per-(sample group, channel) statistics and the fused affine + ReLU epilogue.
//
// The library BatchNorm the reference ends up in costs ~26 us per call on MI355X regardless of size
// (rocprofv3, profiles/r01_*): 70 calls per depth map.  Here: one streaming reduction (float4 loads,
// float64 block partials in the same (G, T, C, 2) layout pf_bn_finalize_f32 consumes), the shared
// finalize kernel, and one streaming normalise+ReLU pass in place.  Both passes are HBM-bound:
// 4*N*C*S bytes read, then read + written.
#include "pf_common.h"

namespace {

// x is (N, C, S) contiguous.  grid = (T, C, N); block t of (n, c) reduces elements [t*chunk, (t+1)*chunk).
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, int C, int64_t S,
                                                            int64_t chunk, double* __restrict__ partials, int T) {
  __shared__ double red[2 * 4];
  const int t = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float* p = x + ((int64_t)n * C + c) * S;
  const int64_t lo = (int64_t)t * chunk;
  const int64_t hi = min(S, lo + chunk);
  float s0 = 0.0f, s1 = 0.0f, q0 = 0.0f, q1 = 0.0f;
  double ds = 0.0, dq = 0.0;
  if (((uintptr_t)(p + lo) & 15) == 0) {
    const int64_t n4 = (hi - lo) >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p + lo);
    int64_t i = threadIdx.x;
    for (; i + 256 < n4; i += 512) {           // two independent 16-byte loads in flight
      const float4 a = p4[i], b = p4[i + 256];
      s0 += (a.x + a.y) + (a.z + a.w);
      q0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
      s1 += (b.x + b.y) + (b.z + b.w);
      q1 += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
    }
    for (; i < n4; i += 256) {
      const float4 a = p4[i];
      s0 += (a.x + a.y) + (a.z + a.w);
      q0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    for (int64_t j = lo + (n4 << 2) + threadIdx.x; j < hi; j += 256) {
      const float v = p[j];
      s0 += v;
      q0 += v * v;
    }
  } else {
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) {
      const float v = p[j];
      s0 += v;
      q0 += v * v;
    }
  }
  ds = (double)s0 + (double)s1;
  dq = (double)q0 + (double)q1;
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

// y = act(x * scale[s, c] + shift[s, c]), s = n / samples_per_stat; grid = (blocks, C, N)
__device__ __forceinline__ void affine_stream(const float* __restrict__ p, float* __restrict__ o, int64_t S,
                                              float a, float b, int relu, const float* __restrict__ add = nullptr) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)p | (uintptr_t)o | (uintptr_t)add) & 15) == 0 && (S & 3) == 0) {
    const int64_t n4 = S >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* a4 = reinterpret_cast<const float4*>(add);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 v = p4[i];
      v.x = fmaf(v.x, a, b);
      v.y = fmaf(v.y, a, b);
      v.z = fmaf(v.z, a, b);
      v.w = fmaf(v.w, a, b);
      if (relu) {
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        v.z = fmaxf(v.z, 0.0f);
        v.w = fmaxf(v.w, 0.0f);
      }
      if (add != nullptr) {                      // other + act(bn(x)): the operand order of the reference's add
        const float4 u = a4[i];
        v.x = u.x + v.x;
        v.y = u.y + v.y;
        v.z = u.z + v.z;
        v.w = u.w + v.w;
      }
      o4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += stride) {
      float v = fmaf(p[i], a, b);
      v = relu ? fmaxf(v, 0.0f) : v;
      o[i] = add != nullptr ? add[i] + v : v;
    }
  }
}

__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int C, int64_t S,
                                                             int samples_per_stat, int relu) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t so = (int64_t)(n / samples_per_stat) * C + c;
  affine_stream(x + ((int64_t)n * C + c) * S, y + ((int64_t)n * C + c) * S, S, scale[so], shift[so], relu);
}

// Block-wide, fixed-order reduction of the `entries` float64 partial pairs of (stat group s, channel c).
__device__ __forceinline__ double2 reduce_partials(const double* __restrict__ partials, int64_t first_entry,
                                                   int entries, int C, int c, double2* red) {
  double a = 0.0, b = 0.0;
  for (int e = threadIdx.x; e < entries; e += 256) {
    const double2 v = *reinterpret_cast<const double2*>(partials + ((first_entry + e) * C + c) * 2);
    a += v.x;
    b += v.y;
  }
  red[threadIdx.x] = make_double2(a, b);
  __syncthreads();
#pragma unroll
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[threadIdx.x].x += red[threadIdx.x + off].x;
      red[threadIdx.x].y += red[threadIdx.x + off].y;
    }
    __syncthreads();
  }
  const double2 r = red[0];
  __syncthreads();
  return r;
}

// BatchNorm finalize fused into the normalise pass: every block reduces the (<= a few hundred) float64
// partials of ITS (stat group, channel) itself -- a 1-2 us prologue instead of a separate launch -- and
// block (x=0, n=0) of each channel additionally applies the running-statistics recurrence over all
// stat groups in order (one update per reference module call).
__global__ __launch_bounds__(256) void channel_bn_apply_kernel(
    const float* __restrict__ x, float* __restrict__ y, const double* __restrict__ partials, int T, int C, int64_t S,
    int samples_per_stat, int G, double count, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps, int relu,
    const float* __restrict__ addend) {
  __shared__ double2 red[256];
  const int c = blockIdx.y, n = blockIdx.z;
  const int s = n / samples_per_stat;
  const int entries = samples_per_stat * T;
  const double2 sums = reduce_partials(partials, (int64_t)s * entries, entries, C, c, red);
  const double mean = sums.x / count;
  double var = sums.y / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float a = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
  const float b = beta[c] - (float)mean * a;
  if (running_mean != nullptr && blockIdx.x == 0 && n == 0) {
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < G; ++g) {
      const double2 sg = reduce_partials(partials, (int64_t)g * entries, entries, C, c, red);
      const double m = sg.x / count;
      double v = sg.y / count - m * m;
      v = v < 0.0 ? 0.0 : v;
      const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
      rm = (1.0f - momentum) * rm + momentum * (float)m;
      rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
    }
    if (threadIdx.x == 0) {
      running_mean[c] = rm;
      running_var[c] = rv;
    }
  }
  affine_stream(x + ((int64_t)n * C + c) * S, y + ((int64_t)n * C + c) * S, S, a, b, relu,
                addend ? addend + ((int64_t)n * C + c) * S : nullptr);
}

// Two BatchNorms in one pass: y = relu(bn2(x2)) + relu(bn1(x1)) -- the last skip add of VolumeConv's decoder
// (reference networks.py:166: conv6_0's output + conv0_1's output, both straight out of their convolutions): one
// launch and 3 streams instead of two launches and 5.  Same prologue as above, once per side.
struct BnSide {
  const double* partials;
  int T;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float momentum, eps;
};

__device__ __forceinline__ void bn_side_affine(const BnSide& B, int C, int c, int n, int samples_per_stat, int G,
                                               double count, double2* red, float& a, float& b) {
  const int s = n / samples_per_stat;
  const int entries = samples_per_stat * B.T;
  const double2 sums = reduce_partials(B.partials, (int64_t)s * entries, entries, C, c, red);
  const double mean = sums.x / count;
  double var = sums.y / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  a = (float)(1.0 / sqrt(var + (double)B.eps)) * B.gamma[c];
  b = B.beta[c] - (float)mean * a;
  if (B.running_mean != nullptr && blockIdx.x == 0 && n == 0) {
    float rm = B.running_mean[c], rv = B.running_var[c];
    for (int g = 0; g < G; ++g) {
      const double2 sg = reduce_partials(B.partials, (int64_t)g * entries, entries, C, c, red);
      const double m = sg.x / count;
      double v = sg.y / count - m * m;
      v = v < 0.0 ? 0.0 : v;
      const double unbiased = count > 1.0 ? v * (count / (count - 1.0)) : v;
      rm = (1.0f - B.momentum) * rm + B.momentum * (float)m;
      rv = (1.0f - B.momentum) * rv + B.momentum * (float)unbiased;
    }
    if (threadIdx.x == 0) {
      B.running_mean[c] = rm;
      B.running_var[c] = rv;
    }
  }
}

__global__ __launch_bounds__(256) void channel_bn_apply2_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                                float* __restrict__ y, BnSide B1, BnSide B2, int C,
                                                                int64_t S, int samples_per_stat, int G, double count) {
  __shared__ double2 red[256];
  const int c = blockIdx.y, n = blockIdx.z;
  float a1, b1, a2, b2;
  bn_side_affine(B1, C, c, n, samples_per_stat, G, count, red, a1, b1);
  bn_side_affine(B2, C, c, n, samples_per_stat, G, count, red, a2, b2);
  const float* p1 = x1 + ((int64_t)n * C + c) * S;
  const float* p2 = x2 + ((int64_t)n * C + c) * S;
  float* o = y + ((int64_t)n * C + c) * S;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)o) & 15) == 0 && (S & 3) == 0) {
    const float4* q1 = reinterpret_cast<const float4*>(p1);
    const float4* q2 = reinterpret_cast<const float4*>(p2);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (S >> 2); i += stride) {
      const float4 v = q1[i], u = q2[i];
      float4 r;
      r.x = fmaxf(fmaf(u.x, a2, b2), 0.0f) + fmaxf(fmaf(v.x, a1, b1), 0.0f);
      r.y = fmaxf(fmaf(u.y, a2, b2), 0.0f) + fmaxf(fmaf(v.y, a1, b1), 0.0f);
      r.z = fmaxf(fmaf(u.z, a2, b2), 0.0f) + fmaxf(fmaf(v.z, a1, b1), 0.0f);
      r.w = fmaxf(fmaf(u.w, a2, b2), 0.0f) + fmaxf(fmaf(v.w, a1, b1), 0.0f);
      o4[i] = r;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += stride)
      o[i] = fmaxf(fmaf(p2[i], a2, b2), 0.0f) + fmaxf(fmaf(p1[i], a1, b1), 0.0f);
  }
}

// Small tensors (the encoder/decoder layers of VolumeConv and the deep tower stages: <= 256 KB per channel):
// statistics, finalize and normalise in ONE launch.  A 1024-thread block owns a channel and walks its stat
// groups in order (the running-statistics recurrence is sequential anyway): pass 1 reduces the group's
// samples_per_stat x S elements (float per lane, float64 across lanes, fixed order), pass 2 re-reads them
// (L2-resident) and writes y.  With y == nullptr only the affine rows scale/shift (G, ld) are produced
// (the next conv applies them while staging).  Saves a 4-5 us dependent launch per layer.
constexpr int kFusedThreads = 1024;

__global__ __launch_bounds__(kFusedThreads) void channel_bn_fused_kernel(
    const float* __restrict__ x, float* __restrict__ y, int C, int64_t S, int samples_per_stat, int G,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, int relu, float* __restrict__ scale_out,
    float* __restrict__ shift_out, int ld_affine) {
  __shared__ double2 red[kFusedThreads / 64];
  __shared__ double2 total;
  const int c = blockIdx.x, tid = threadIdx.x;
  const double count = (double)samples_per_stat * (double)S;
  const bool vec = ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) && (S & 3) == 0;
  float rm = 0.0f, rv = 0.0f;
  const bool track = running_mean != nullptr;
  if (track) {
    rm = running_mean[c];
    rv = running_var[c];
  }
  for (int g = 0; g < G; ++g) {
    float s0 = 0.0f, q0 = 0.0f;
    for (int j = 0; j < samples_per_stat; ++j) {
      const float* p = x + ((int64_t)(g * samples_per_stat + j) * C + c) * S;
      if (vec) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int64_t i = tid; i < (S >> 2); i += kFusedThreads) {
          const float4 a = p4[i];
          s0 += (a.x + a.y) + (a.z + a.w);
          q0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        }
      } else {
        for (int64_t i = tid; i < S; i += kFusedThreads) {
          const float v = p[i];
          s0 += v;
          q0 += v * v;
        }
      }
    }
    double ds = (double)s0, dq = (double)q0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ds += __shfl_xor(ds, off);
      dq += __shfl_xor(dq, off);
    }
    if ((tid & 63) == 0) red[tid >> 6] = make_double2(ds, dq);
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int w = 0; w < kFusedThreads / 64; ++w) {
        a += red[w].x;
        b += red[w].y;
      }
      total = make_double2(a, b);
    }
    __syncthreads();
    const double mean = total.x / count;
    double var = total.y / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float a = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
    const float b = beta[c] - (float)mean * a;
    if (track) {
      const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
      rm = (1.0f - momentum) * rm + momentum * (float)mean;
      rv = (1.0f - momentum) * rv + momentum * (float)unbiased;
    }
    if (scale_out != nullptr && tid == 0) {
      scale_out[(int64_t)g * ld_affine + c] = a;
      shift_out[(int64_t)g * ld_affine + c] = b;
    }
    if (y != nullptr) {
      for (int j = 0; j < samples_per_stat; ++j) {
        const int64_t o = ((int64_t)(g * samples_per_stat + j) * C + c) * S;
        if (vec) {
          const float4* p4 = reinterpret_cast<const float4*>(x + o);
          float4* o4 = reinterpret_cast<float4*>(y + o);
          for (int64_t i = tid; i < (S >> 2); i += kFusedThreads) {
            float4 v = p4[i];
            v.x = fmaf(v.x, a, b);
            v.y = fmaf(v.y, a, b);
            v.z = fmaf(v.z, a, b);
            v.w = fmaf(v.w, a, b);
            if (relu) {
              v.x = fmaxf(v.x, 0.0f);
              v.y = fmaxf(v.y, 0.0f);
              v.z = fmaxf(v.z, 0.0f);
              v.w = fmaxf(v.w, 0.0f);
            }
            o4[i] = v;
          }
        } else {
          for (int64_t i = tid; i < S; i += kFusedThreads) {
            const float v = fmaf(x[o + i], a, b);
            y[o + i] = relu ? fmaxf(v, 0.0f) : v;
          }
        }
      }
    }
    // (red/total are rewritten only after the next group's first barrier; x may alias y: each element is
    // read in pass 1 before any lane's pass 2 writes thanks to the barriers above)
  }
  if (track && tid == 0) {
    running_mean[c] = rm;
    running_var[c] = rv;
  }
}

}  // namespace

extern "C" {

int pf_channel_bn_fused_f32(const float* x, float* y, int64_t N, int64_t C, int64_t S, int samples_per_stat,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, int relu, float* scale, float* shift, int ld_affine,
                            void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= INT32_MAX && samples_per_stat >= 1);
  PF_REQUIRE(N % samples_per_stat == 0);
  PF_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && (scale == nullptr) == (shift == nullptr));
  PF_REQUIRE(y != nullptr || scale != nullptr);
  PF_REQUIRE(scale == nullptr || ld_affine >= C);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(x && gamma && beta);
  hipLaunchKernelGGL(channel_bn_fused_kernel, dim3((unsigned)C), dim3(kFusedThreads), 0, (hipStream_t)stream, x, y,
                     (int)C, S, samples_per_stat, (int)(N / samples_per_stat), gamma, beta, running_mean, running_var,
                     momentum, eps, relu, scale, shift, ld_affine);
  return pf_launch_status();
}

int pf_norm_blocks(int64_t S) {
  if (S <= 0) return 0;
  const int64_t t = (S + 8191) / 8192;       // >= 8192 elements (32 KB) per block
  return (int)(t > 64 ? 64 : t);
}

int pf_channel_stats_f32(const float* x, int64_t N, int64_t C, int64_t S, double* partials, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(x && partials);
  const int T = pf_norm_blocks(S);
  int64_t chunk = (S + T - 1) / T;
  chunk = (chunk + 3) & ~(int64_t)3;          // keep every block's start 16-byte aligned relative to the plane
  dim3 grid((unsigned)T, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(channel_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (int)C, S, chunk, partials, T);
  return pf_launch_status();
}

int pf_channel_affine_f32(const float* x, float* y, const float* scale, const float* shift, int64_t N, int64_t C,
                          int64_t S, int samples_per_stat, int relu, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1);
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(x && y && scale && shift);
  int64_t blocks = (S / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  dim3 grid((unsigned)blocks, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(channel_affine_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, scale, shift, (int)C, S,
                     samples_per_stat, relu);
  return pf_launch_status();
}

int pf_channel_bn_apply_f32(const float* x, float* y, const double* partials, int T, int64_t N, int64_t C, int64_t S,
                            int samples_per_stat, double count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int relu,
                            const float* addend, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1 && T >= 1);
  PF_REQUIRE(N % samples_per_stat == 0 && count > 0.0);
  PF_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(x && y && partials && gamma && beta);
  int64_t blocks = (S / 4 + 1023) / 1024;     // >= 4096 elements per block amortise the statistics prologue
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  dim3 grid((unsigned)blocks, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(channel_bn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, partials, T, (int)C, S,
                     samples_per_stat, (int)(N / samples_per_stat), count, gamma, beta, running_mean, running_var,
                     momentum, eps, relu, addend);
  return pf_launch_status();
}

int pf_channel_bn_apply2_f32(const float* x1, const double* partials1, int T1, const float* gamma1, const float* beta1,
                             float* running_mean1, float* running_var1, float momentum1, float eps1, const float* x2,
                             const double* partials2, int T2, const float* gamma2, const float* beta2,
                             float* running_mean2, float* running_var2, float momentum2, float eps2, float* y, int64_t N,
                             int64_t C, int64_t S, int samples_per_stat, double count, void* stream) {
  PF_REQUIRE(N >= 0 && C >= 0 && S >= 0 && N <= 65535 && C <= 65535 && samples_per_stat >= 1 && T1 >= 1 && T2 >= 1);
  PF_REQUIRE(N % samples_per_stat == 0 && count > 0.0);
  PF_REQUIRE((running_mean1 == nullptr) == (running_var1 == nullptr));
  PF_REQUIRE((running_mean2 == nullptr) == (running_var2 == nullptr));
  if (N == 0 || C == 0 || S == 0) return PF_OK;
  PF_REQUIRE(x1 && x2 && y && partials1 && partials2 && gamma1 && beta1 && gamma2 && beta2);
  const BnSide B1{partials1, T1, gamma1, beta1, running_mean1, running_var1, momentum1, eps1};
  const BnSide B2{partials2, T2, gamma2, beta2, running_mean2, running_var2, momentum2, eps2};
  int64_t blocks = (S / 4 + 1023) / 1024;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  dim3 grid((unsigned)blocks, (unsigned)C, (unsigned)N);
  hipLaunchKernelGGL(channel_bn_apply2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x1, x2, y, B1, B2, (int)C, S,
                     samples_per_stat, (int)(N / samples_per_stat), count);
  return pf_launch_status();
}

}  // extern "C"
