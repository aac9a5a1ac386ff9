// BatchNorm finalize WITHOUT its own launch, two ways: folded into the kernel that produces the statistics ("last block
// done", first half of this file; measured slower than the separate launch, off by default) and folded into the kernel
// that CONSUMES the normalised tensor (pf_bn_resolve, second half; on by default where the statistics rows are few).
//
// BatchNorm finalize folded into the kernel that produces the statistics ("last block done").
//
// Every BatchNorm on the path runs in training mode (reference test.py:58, networks.py:41,77, nn/conv.py:29-35):
// the kernel that produces a layer's raw output also produces per-block float64 (sum, sum of squares) rows,
// and a global reduction of those rows must finish before the next kernel can normalise.  As a separate
// launch (pf_bn_finalize_jobs_f32) that reduction cost 25 dependent graph nodes per depth map -- 5-7 us of
// latency-bound kernel plus a node boundary each, ~8 % of the step.  Here it rides on the producer:
//
//   level 0  every block writes its row write-through (sc1 stores: visible device-wide once vmcnt drains),
//            then takes a ticket on its CLUSTER's counter (F consecutive blocks of one group);
//   level 1  the block that draws the cluster's last ticket sums the cluster's rows in row order into one
//            level-1 row (this overlaps with the other clusters still computing), then takes a ticket on
//            the launch counter;
//   final    the block that draws the last launch ticket sums the <= ~64 level-1 rows per statistic group in
//            (group, cluster) order, computes mean / biased variance / scale / shift, and applies the
//            running-statistics recurrence group by group, exactly like pf_bn_finalize_jobs_f32.
//
// Every sum is over a FIXED set in a FIXED order, so results are bit-reproducible run to run (no float
// atomics); only which block does the summing varies.  Visibility follows the placement-independent recipe
// of the CDNA4 guide (section 6, guideline 16, write-through form): 8-byte sc1 stores -> s_waitcnt vmcnt(0)
// in every storing wave -> __syncthreads -> one relaxed agent-scope fetch_add; the reducer reads with sc1
// loads (never a plain load: a CU's L1 is not refreshed by other CUs' stores).  Counters are zero on entry
// and are reset by the block that draws the last ticket, so a replayed hipGraph needs no memset node.
#pragma once

#include "pf_common.h"

constexpr int kTailMaxJobs = 2;
constexpr int kTailRows = 64;          // target number of rows the final block reads
constexpr int kTailSmemDoubles = 1032; // LDS the tail borrows from its host kernel (8 256 bytes)

struct PfTail {
  pf_bn_job job[kTailMaxJobs];  // job[*].partials / T / pcols / G describe the level-0 rows of THIS launch
  int njobs;                    // 0: no tail (rows are consumed by a later pass)
  int F, clusters;              // fan-in of level 1 (1: level 1 skipped) and clusters per group
  double* l1;                   // (G, clusters, pcols, 2) level-1 rows, unused when F == 1
  unsigned* cnt;                // [0] launch counter, [1 + g*clusters + cl] cluster counters
};

// ---- host side ----------------------------------------------------------------------------------
static inline int pf_tail_fan(int G, int T) {
  const int64_t rows = (int64_t)G * T;
  int64_t F = (rows + kTailRows - 1) / kTailRows;
  if (F < 1) F = 1;
  if (F > T) F = T;
  return (int)F;
}
static inline int pf_tail_clusters(int G, int T) {
  const int F = pf_tail_fan(G, T);
  return (T + F - 1) / F;
}
static inline bool pf_tail_cols_ok(int pcols) {
  return pcols == 8 || pcols == 16 || pcols == 32 || pcols == 64 || pcols == 128;
}

// Fills `t` for a launch whose level-0 rows are partials (G, T, pcols, 2) followed, in the same buffer, by
// pf_bn_tail_rows(G, T) level-1 rows.  Returns PF_OK, or an error code (nothing must be launched then).
static inline int pf_tail_setup(PfTail& t, const pf_bn_job* jobs, int njobs, double* partials, int G, int T, int pcols,
                                unsigned* tickets) {
  t.njobs = 0;
  t.F = 1;
  t.clusters = T;
  t.l1 = nullptr;
  t.cnt = nullptr;
  if (njobs == 0) return PF_OK;
  PF_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= kTailMaxJobs && partials != nullptr && tickets != nullptr);
  if (!pf_tail_cols_ok(pcols)) return PF_ERR_UNSUPPORTED;
  for (int i = 0; i < njobs; ++i) {
    pf_bn_job j = jobs[i];
    PF_REQUIRE(j.col0 >= 0 && j.C >= 1 && j.C <= 128 && j.col0 + j.C <= pcols && j.count > 0.0);
    PF_REQUIRE(j.groups_per_stat >= 1 && (G % j.groups_per_stat) == 0 && j.ld_affine >= j.C);
    PF_REQUIRE(j.gamma && j.beta && j.scale && j.shift);
    PF_REQUIRE((j.running_mean == nullptr) == (j.running_var == nullptr));
    j.partials = partials;
    j.T = T;
    j.pcols = pcols;
    j.G = G;
    t.job[i] = j;
  }
  for (int i = njobs; i < kTailMaxJobs; ++i) t.job[i] = t.job[0];
  t.njobs = njobs;
  t.F = pf_tail_fan(G, T);
  t.clusters = (T + t.F - 1) / t.F;
  t.l1 = partials + (int64_t)G * T * pcols * 2;
  t.cnt = tickets;
  return PF_OK;
}

// A pending BatchNorm handed to its CONSUMER (`in_bn` of pf_conv2d_wide_f32 / pf_pointwise_gemm_f32 /
// pf_flow_head_f32, resolved per block by pf_bn_resolve below): the job must describe finished statistics rows.
#ifndef PF_RESOLVE_BATCH
#define PF_RESOLVE_BATCH 20
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
// ---- device side --------------------------------------------------------------------------------
__device__ __forceinline__ void pf_row_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_store_dwordx2 sc1
}
__device__ __forceinline__ double pf_row_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 sc1
}

// Sum of rows [r0, r1) at stride `stride` doubles, in row order, 16 loads in flight.
__device__ __forceinline__ double pf_tail_sum_rows(const double* base, int64_t stride, int r0, int r1) {
  double acc = 0.0;
  int r = r0;
  for (; r + 16 <= r1; r += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pf_row_load(base + (int64_t)(r + u) * stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (r < r1) {
    // unconditional loads (rows past the end re-read the last row) and a masked add: a per-element
    // "load or zero" select makes the compiler branch around every load and wait for each one in turn
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pf_row_load(base + (int64_t)min(r + u, r1 - 1) * stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (r + u < r1) ? v[u] : 0.0;
  }
  return acc;
}

// ---- the finalize folded into the CONSUMER ---------------------------------------------------------------------
// The producer-side tail above removes a launch but not the chain of dependent L2 round trips behind it (rows ->
// level-1 rows -> statistics): measured, it is no faster than the separate finalize kernel.
// On the consumer's side the same chain costs nothing when the consumer has other latency to wait for anyway: every
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
    // kResolveBatch rows in flight: the tower layers hand every slice exactly 20 rows (80 / 160 / 320 / 640 rows for
    // 64 / 32 / 16 / 8 channels), i.e. ONE round trip instead of two
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

// One finalize job on `nrows` rows per group (the semantics of bn_finalize_kernel, edgeconv.hip).
template <int THREADS>
__device__ __forceinline__ void pf_tail_finalize(const pf_bn_job& J, const double* rows, int nrows, double* smem) {
  const int tid = threadIdx.x;
  const int C = J.C;
  const int S = J.G / J.groups_per_stat;
  const int64_t rstride = (int64_t)J.pcols * 2;
  const int per_stat = J.groups_per_stat * nrows;          // rows behind one statistic (consecutive in memory)
  const int SS = THREADS / C > 0 ? THREADS / C : 1;        // statistic groups per round
  double* red = smem;                                      // [slices][P][2] <= 512 doubles
  double* stat = smem + 512;                               // [P][2]         <= 512 doubles
  const bool track = J.running_mean != nullptr;
  float rm = 0.0f, rv = 0.0f;
  if (track && tid < C) {
    rm = J.running_mean[tid];
    rv = J.running_var[tid];
  }
  for (int s0 = 0; s0 < S; s0 += SS) {
    const int ns = min(SS, S - s0);
    const int P = ns * C;                                  // (statistic group, channel) pairs of this round
    int slices = THREADS / P;                              // row slices per pair
    if (slices > per_stat) slices = per_stat;
    if (slices > 256 / P) slices = 256 / P > 0 ? 256 / P : 1;   // red[] holds 256 (sum, sq) entries
    const int per = (per_stat + slices - 1) / slices;
    double a = 0.0, b = 0.0;
    const bool act = tid < P * slices;
    const int pair = tid % P, sl = tid / P;
    if (act) {
      const int s = s0 + pair / C, c = pair % C;
      const double* base = rows + ((int64_t)s * per_stat * J.pcols + J.col0 + c) * 2;
      const int r0 = sl * per, r1 = min(per_stat, r0 + per);
      a = pf_tail_sum_rows(base, rstride, r0, r1);
      b = pf_tail_sum_rows(base + 1, rstride, r0, r1);
    }
    __syncthreads();                                       // previous round's stat[] has been consumed
    if (act) {
      red[(sl * P + pair) * 2 + 0] = a;
      red[(sl * P + pair) * 2 + 1] = b;
    }
    __syncthreads();
    if (tid < P) {
      double sum = 0.0, sq = 0.0;
      for (int i = 0; i < slices; ++i) {
        sum += red[(i * P + tid) * 2 + 0];
        sq += red[(i * P + tid) * 2 + 1];
      }
      const double mean = sum / J.count;
      double var = sq / J.count - mean * mean;
      stat[tid * 2 + 0] = mean;
      stat[tid * 2 + 1] = var < 0.0 ? 0.0 : var;
    }
    __syncthreads();
    if (tid < C) {
      const float g_ = J.gamma[tid], b_ = J.beta[tid];
      for (int si = 0; si < ns; ++si) {
        const double mean = stat[(si * C + tid) * 2 + 0], var = stat[(si * C + tid) * 2 + 1];
        const float invstd = (float)(1.0 / sqrt(var + (double)J.eps));
        const float a_ = invstd * g_;
        J.scale[(int64_t)(s0 + si) * J.ld_affine + tid] = a_;
        J.shift[(int64_t)(s0 + si) * J.ld_affine + tid] = b_ - (float)mean * a_;
        if (track) {
          const double unbiased = J.unbias_n > 1.0 ? var * (J.unbias_n / (J.unbias_n - 1.0)) : var;
          rm = (1.0f - J.momentum) * rm + J.momentum * (float)mean;
          rv = (1.0f - J.momentum) * rv + J.momentum * (float)unbiased;
        }
      }
    }
  }
  if (track && tid < C) {
    J.running_mean[tid] = rm;
    J.running_var[tid] = rv;
  }
  __syncthreads();                                         // smem is free for the next job
}

// Called by EVERY thread of block (tb, g) after the block's level-0 row has been written with pf_row_store.
// `smem`: kTailSmemDoubles doubles of LDS nobody else uses any more.  THREADS = blockDim.x.
template <int THREADS>
__device__ __forceinline__ void pf_bn_tail(const PfTail& t, int g, int tb, double* smem) {
  const pf_bn_job& J0 = t.job[0];
  const int tid = threadIdx.x;
  const int T = J0.T, G = J0.G;
  const int E = J0.pcols * 2;                              // doubles per row: 16 .. 256, a power of two
  int* flag = reinterpret_cast<int*>(smem + 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's row stores are out
  __syncthreads();
  const double* rows = J0.partials;
  int nrows = T;
  if (t.F > 1) {
    const int cl = tb / t.F;
    const int csize = min(t.F, T - cl * t.F);
    if (tid == 0) {
      unsigned* c = t.cnt + 1 + g * t.clusters + cl;
      const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = old == (unsigned)(csize - 1) ? 1 : 0;
      if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
    // level 1: this cluster's rows, in row order, -> one level-1 row
    const int R = THREADS / E;                             // row slices (>= 1)
    const int e = tid % E, sl = tid / E;
    const int per = (csize + R - 1) / R;
    const int r0 = sl * per, r1 = min(csize, r0 + per);
    double acc = 0.0;
    if (sl < R) acc = pf_tail_sum_rows(rows + ((int64_t)g * T + (int64_t)cl * t.F) * E + e, E, r0, r1);
    __syncthreads();                                       // everybody has read *flag
    if (R > 1) {
      if (sl < R) smem[sl * E + e] = acc;
      __syncthreads();
      if (sl == 0) {
        acc = 0.0;
        for (int i = 0; i < R; ++i) acc += smem[i * E + e];
      }
    }
    if (sl == 0) pf_row_store(t.l1 + ((int64_t)g * t.clusters + cl) * E + e, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    rows = t.l1;
    nrows = t.clusters;
  }
  const unsigned arrivals = (unsigned)G * (unsigned)nrows;
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(t.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == arrivals - 1u ? 1 : 0;
    if (last) __hip_atomic_store(t.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  __syncthreads();
  for (int j = 0; j < t.njobs; ++j) pf_tail_finalize<THREADS>(t.job[j], rows, nrows, smem);
}
#endif  // __HIPCC__
