// BatchNorm finalize folded into the kernel that CONSUMES the normalised tensor (pf_bn_resolve).
//
// Every BatchNorm on the path runs in training mode (reference test.py:58, networks.py:41,77, nn/conv.py:29-35): the
// kernel that produces a layer's raw output also produces per-block float64 (sum, sum of squares) rows, and a global
// reduction of those rows must finish before the next kernel can normalise.  As a separate launch
// (pf_bn_finalize_jobs_f32) that is a 5-7 us latency-bound node per layer.  Folding it into the PRODUCER ("last
// block done": ticket counters, write-through rows, a two-level reduction by whichever block finishes last) was built
// and measured in round 2 -- correct, bit-reproducible and 3 % SLOWER than the separate launch (the chain of dependent
// L2 round trips stays on the critical path, profiles/archive/r02/r02a_fused_bn_ab.log); it was removed in round 3.  On the
// CONSUMER's side the same chain hides behind latency the consumer waits for anyway:
#pragma once

#include "pf_common.h"

// A pending BatchNorm handed to its CONSUMER (`in_bn` of pf_conv2d_wide_f32 / pf_pointwise_gemm_f32 /
// pf_flow_head_f32, resolved per block by pf_bn_resolve below): the job must describe finished statistics rows.
#ifndef PF_RESOLVE_BATCH
#define PF_RESOLVE_BATCH 5
#endif
constexpr int kResolveBatch = PF_RESOLVE_BATCH;
constexpr int kResolveMaxRows = 4096;   // rows behind one statistic; beyond that the re-reduction per block is absurd
static inline int pf_bn_in_check(const pf_bn_job* j, int C, int stat_groups) {
  PF_REQUIRE(j != nullptr && j->partials != nullptr && j->gamma != nullptr && j->beta != nullptr);
  PF_REQUIRE(j->T >= 1 && j->pcols >= 1 && j->col0 >= 0 && j->C == C && C >= 1 && C <= 256 && j->col0 + C <= j->pcols);
  PF_REQUIRE(j->count > 0.0 && j->G >= 1 && j->groups_per_stat >= 1 && (j->G % j->groups_per_stat) == 0);
  PF_REQUIRE(j->G / j->groups_per_stat == stat_groups);
  if ((int64_t)j->groups_per_stat * j->T > kResolveMaxRows) return PF_ERR_UNSUPPORTED;
  return PF_OK;
}

#if defined(__HIPCC__)
// Every
// block of the NEXT kernel reduces the producer's rows of ITS statistic group itself (fixed slices, fixed order:
// bit-reproducible, and every block computes the same bits) while its own first loads -- input patch, weights --
// are in flight.  That pays when a statistic group has few rows (<= ~200: persistent GEMM blocks, the small maps
// of conv2d_wide), i.e. rows x C x 16 bytes of L2 reads per consumer block.  Running statistics and the (scale,
// shift) rows of the job are NOT written here; pf_bn_finalize_jobs_f32 on the same job, off the critical path,
// does that.
//
// All THREADS threads call it; scale/shift of statistic group s for channels [0, J.C) land in sm_scale / sm_shift
// (LDS, J.C floats each); `red`: 2 * THREADS doubles of LDS.  J.C <= THREADS.  Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ void pf_bn_resolve(const pf_bn_job& J, int s, float* sm_scale, float* sm_shift,
                                              double* red) {
  const int tid = threadIdx.x;
  const int C = J.C;
  const int per_stat = J.groups_per_stat * J.T;            // rows behind one statistic (consecutive in memory)
  int slices = THREADS / C;
  if (slices > per_stat) slices = per_stat;
  const int per = (per_stat + slices - 1) / slices;
  const int c = tid % C, sl = tid / C;
  double a = 0.0, b = 0.0;
  if (sl < slices) {
    const double2* base = reinterpret_cast<const double2*>(J.partials) + (int64_t)s * per_stat * J.pcols + J.col0 + c;
    const int r0 = sl * per, r1 = min(per_stat, r0 + per);
    // kResolveBatch rows in flight.  The tower layers hand every slice exactly 20 rows (80 / 160 / 320 / 640 rows for
    // 64 / 32 / 16 / 8 channels); rounds 3-5 kept all 20 in flight (one round trip) -- 80 VGPRs that live only in this
    // prologue but set the whole kernel's register allocation: the 8 -> 8 tower kernel ran 3 waves per SIMD instead of 5,
    // 32 -> 32 and 64 -> 64 2 instead of 4 (build/resource_usage.json).  5 in flight = four round trips of a prologue that
    // other resident blocks cover, and the occupancy of the instantiations without a pending BatchNorm: +1.6 % on the
    // headline, same box, two repetitions (profiles/r06n_resolve_batch.md).  The sums are added in the same order.
    for (int r = r0; r < r1; r += kResolveBatch) {
      double2 v[kResolveBatch];
#pragma unroll
      for (int u = 0; u < kResolveBatch; ++u) v[u] = base[(int64_t)min(r + u, r1 - 1) * J.pcols];   // unconditional loads
#pragma unroll
      for (int u = 0; u < kResolveBatch; ++u) {
        a += (r + u < r1) ? v[u].x : 0.0;
        b += (r + u < r1) ? v[u].y : 0.0;
      }
    }
  }
  red[2 * tid + 0] = a;
  red[2 * tid + 1] = b;
  __syncthreads();
  if (tid < C) {
    double sum = 0.0, sq = 0.0;
    for (int i = 0; i < slices; ++i) {
      sum += red[2 * (i * C + tid) + 0];
      sq += red[2 * (i * C + tid) + 1];
    }
    const double mean = sum / J.count;
    double var = sq / J.count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)J.eps));
    const float a_ = invstd * J.gamma[tid];
    sm_scale[tid] = a_;
    sm_shift[tid] = J.beta[tid] - (float)mean * a_;
  }
  __syncthreads();
}
#endif  // __HIPCC__
