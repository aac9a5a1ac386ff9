// The inverse of a neighbour index tensor: for every point, WHO gathers it.
//
// The backward of a neighbour gather (reference functions/csrc/gather_knn_kernel.cu:50-89) is a scatter-add:
// grad[idx[n, j]] += g[n, j], done there -- and in rounds 1-2 here -- with float atomics, i.e. in arrival order:
// two runs of the same step give different low bits.  With the pairs grouped by target the scatter becomes a GATHER:
// point m sums the pairs that name it, in ascending pair order, with plain stores -- bit-reproducible and no zero-fill
// of the output.  One inversion serves every backward pass that shares the index tensor (the three EdgeConv layers of
// a PointFlow iteration).
//
//   key(p)   = g * Ng + clamp(idx[g, n, j], 0, Ng - 1)        p = (g * Ng + n) * k + j      (the forward's clamp)
//   start[m] = number of pairs with key < m, m in [0, G * Ng]
//   order    = the pair ids grouped by key, ascending inside a group: pairs of m = order[start[m] .. start[m+1])
//
// A counting sort in five kernel launches (seven beyond 4 M rows) and nothing else -- no memset / memcpy nodes, no library call -- so the whole
// thing can sit inside a captured hipGraph (train_step.GraphedTrainStep; rocPRIM's radix sort clears its look-back
// state with hipMemsetAsync, and memset nodes recorded from the autograd thread are not replayed reliably, see
// pf_common.h):  count (integer atomics: the totals do not depend on arrival order) -> exclusive scan (one
// chained launch) -> fill (a slot per pair from an atomic cursor: arrival order)
// -> every list sorted by pair id (one wave per list, rank counting in LDS), which makes the result independent of the
// arrival order again.
#include "pf_common.h"

namespace {

constexpr int kScanBlock = 1024;   // elements per scan block (256 threads x 4)

__device__ __forceinline__ uint32_t pair_key(const int64_t* __restrict__ idx, int64_t p, int k, int Ng) {
  const int64_t row = p / k;
  const int64_t g = row / Ng;
  int64_t i = idx[p];
  i = i < 0 ? 0 : (i > Ng - 1 ? Ng - 1 : i);
  return (uint32_t)(g * Ng + i);
}

__global__ __launch_bounds__(256) void inverse_zero_kernel(uint32_t* __restrict__ a, int64_t na, uint32_t* __restrict__ b,
                                                           int64_t nb, uint32_t* __restrict__ c, int64_t nc) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = 0u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nc; i += stride) c[i] = 0u;
}

__global__ __launch_bounds__(256) void inverse_count_kernel(const int64_t* __restrict__ idx, int64_t pairs, int k, int Ng,
                                                            uint32_t* __restrict__ count) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < pairs) atomicAdd(count + pair_key(idx, p, k, Ng), 1u);
}

// exclusive scan of count[0 .. n) in place, three launches: block sums, scan of the block sums, block scans
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const uint32_t* __restrict__ v, int64_t n,
                                                              uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[256];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * threadIdx.x;
  uint32_t s = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) s += base + u < n ? v[base + u] : 0u;
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t* __restrict__ sums, int nblocks) {
  // one block: exclusive scan of `nblocks` values, 1024 at a time with a running carry
  __shared__ uint32_t buf[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0u;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t x = i < nblocks ? sums[i] : 0u;
    buf[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {           // Hillis-Steele inclusive scan
      const uint32_t t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) sums[i] = carry + buf[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(uint32_t* __restrict__ v, int64_t n,
                                                          const uint32_t* __restrict__ sums) {
  __shared__ uint32_t part[256];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + 4 * threadIdx.x;
  uint32_t x[4], s = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    x[u] = base + u < n ? v[base + u] : 0u;
    s += x[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + part[threadIdx.x] - s;   // exclusive prefix of this thread's four elements
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (base + u < n) v[base + u] = run;
    run += x[u];
  }
}

// exclusive scan of count[0 .. n) in place, ONE launch (round 6; the three launches above cost ~14 us of launch latency,
// five times per training step).  Every block takes a ticket (so it only ever waits for blocks that already run), scans its
// 1024 elements in LDS, publishes its total at once and then LOOKS BACK: its 256 threads read the totals of all earlier
// tickets in parallel (spinning on the few that are not there yet) and the block adds them up -- no chain of dependent
// waits from block to block (a serial hand-over took 26 us for 101 blocks).  Integer sums: the result does not depend
// on the arrival order.  chain[0] = the ticket counter, chain[1 + t] = (1 << 32) | total of ticket t; all zeroed by
// inverse_zero_kernel.  Quadratic in the number of blocks, so scan_in_place() below uses it up to kChainBlocks.
constexpr int kChainBlocks = 4096;
__global__ __launch_bounds__(256) void scan_chain_kernel(uint32_t* __restrict__ v, int64_t n,
                                                         unsigned long long* __restrict__ chain) {
  __shared__ uint32_t part[256];
  __shared__ uint32_t look[256];
  __shared__ unsigned long long ticket;
  if (threadIdx.x == 0) ticket = atomicAdd(chain, 1ull);
  __syncthreads();
  const int64_t t = (int64_t)ticket;
  const int64_t base = t * kScanBlock + 4 * threadIdx.x;
  uint32_t x[4], s = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    x[u] = base + u < n ? v[base + u] : 0u;
    s += x[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t tt = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += tt;
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(chain + 1 + t, (1ull << 32) | (unsigned long long)part[255]);   // (the slot was zero)
  uint32_t acc = 0;
  for (int64_t u = threadIdx.x; u < t; u += 256) {
    unsigned long long got;
    do {
      got = atomicAdd(chain + 1 + u, 0ull);                     // (an atomic read at the L2: never a stale line)
    } while ((got >> 32) == 0ull);
    acc += (uint32_t)got;
  }
  look[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) look[threadIdx.x] += look[threadIdx.x + w];
    __syncthreads();
  }
  uint32_t run = look[0] + part[threadIdx.x] - s;              // exclusive prefix of this thread's four elements
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (base + u < n) v[base + u] = run;
    run += x[u];
  }
}

__global__ __launch_bounds__(256) void inverse_fill_kernel(const int64_t* __restrict__ idx, int64_t pairs, int k, int Ng,
                                                           const uint32_t* __restrict__ start,
                                                           uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const uint32_t key = pair_key(idx, p, k, Ng);
  order[start[key] + atomicAdd(cursor + key, 1u)] = (uint32_t)p;
}

// The same counting sort for pairs whose keys are given as an array (csrc/warp_bwd.hip: (view, point) pairs keyed by
// the texel cell their bilinear footprint starts in); keys >= nkeys are dropped (no list holds them).
__global__ __launch_bounds__(256) void keys_count_kernel(const uint32_t* __restrict__ keys, int64_t pairs, uint32_t nkeys,
                                                         uint32_t* __restrict__ count) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < pairs && keys[p] < nkeys) atomicAdd(count + keys[p], 1u);
}

__global__ __launch_bounds__(256) void keys_fill_kernel(const uint32_t* __restrict__ keys, int64_t pairs, uint32_t nkeys,
                                                        const uint32_t* __restrict__ start,
                                                        uint32_t* __restrict__ cursor, uint32_t* __restrict__ order) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const uint32_t key = keys[p];
  if (key < nkeys) order[start[key] + atomicAdd(cursor + key, 1u)] = (uint32_t)p;
}

// Lists sorted ascending by pair id, one 64-lane wave per list when lists are long (the coarse cost volume puts ~D
// points on every texel cell): rank of an element = number of smaller ids in its list (ids are distinct), computed
// by all lanes against a copy of the list in LDS; lists longer than kSortCap are sorted in chunks and merged (bounded
// work for hub rows).  Result independent of the arrival order of keys_fill_kernel.
constexpr int kSortCap = 1024;
__global__ __launch_bounds__(256) void sort_lists_wave_kernel(const uint32_t* __restrict__ start, int64_t rows,
                                                              uint32_t* __restrict__ order,
                                                              uint32_t* __restrict__ scratch) {
  __shared__ uint32_t buf[4][kSortCap];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + wave;
  if (m >= rows) return;
  const uint32_t t0 = start[m], t1 = start[m + 1];
  const uint32_t len = t1 - t0;
  if (len < 2) return;
  if (len <= (uint32_t)kSortCap) {
    for (uint32_t i = lane; i < len; i += 64) buf[wave][i] = order[t0 + i];
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < len; i += 64) {
      const uint32_t v = buf[wave][i];
      uint32_t r = 0;
      for (uint32_t j = 0; j < len; ++j) r += buf[wave][j] < v ? 1u : 0u;
      order[t0 + r] = v;
    }
  } else {
    // Degenerate geometry (thousands of pairs on one row: hub points, clamped out-of-range indices).  Bounded work:
    // chunks of kSortCap sorted by rank counting in LDS, then log2(len / kSortCap) merge passes between `scratch` and
    // `order` in which every element finds its place by a binary search in the partner run: O(len log^2 len / 64) for
    // the wave, where a rank count over the whole list is O(len^2 / 64) -- 1.6 M pairs on one row: ~1e7 instead of
    // ~4e10 steps.  (volatile: a pass reads what other lanes of this wave wrote in the pass before.)
    volatile uint32_t* ord = order + t0;
    volatile uint32_t* scr = scratch + t0;
    for (uint32_t c0 = 0; c0 < len; c0 += (uint32_t)kSortCap) {
      const uint32_t cl = min((uint32_t)kSortCap, len - c0);
      for (uint32_t i = lane; i < cl; i += 64) buf[wave][i] = ord[c0 + i];
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < cl; i += 64) {
        const uint32_t v = buf[wave][i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < cl; ++j) r += buf[wave][j] < v ? 1u : 0u;
        scr[c0 + r] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    volatile uint32_t* src = scr;
    volatile uint32_t* dst = ord;
    for (uint32_t width = (uint32_t)kSortCap; width < len; width <<= 1) {
      __threadfence();
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < len; i += 64) {
        const uint32_t run = i / width;
        const uint32_t own0 = run * width, oth0 = (run ^ 1u) * width;
        const uint32_t v = src[i];
        uint32_t pos = i;
        if (oth0 < len) {
          const uint32_t on = min(width, len - oth0);
          uint32_t lo = 0, hi = on;                       // number of partner elements smaller than v (ids are distinct)
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (src[oth0 + mid] < v) {
              lo = mid + 1;
            } else {
              hi = mid;
            }
          }
          pos = min(own0, oth0) + (i - own0) + lo;
        }
        dst[pos] = v;
      }
      volatile uint32_t* t = src;
      src = dst;
      dst = t;
    }
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    if (src != ord)
      for (uint32_t i = lane; i < len; i += 64) ord[i] = src[i];
  }
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// `chain`: (nblocks + 1) zeroed 64-bit words (the one-launch form) that double as the nblocks block sums of the three-launch form
void scan_in_place(uint32_t* v, int64_t n, int nblocks, unsigned long long* chain, hipStream_t s) {
  if (nblocks <= kChainBlocks) {
    hipLaunchKernelGGL(scan_chain_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, v, n, chain);
    return;
  }
  uint32_t* sums = reinterpret_cast<uint32_t*>(chain);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, v, n, sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, sums, nblocks);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, v, n, sums);
}

}  // namespace

extern "C" {

int64_t pf_knn_inverse_workspace(int G, int Ng, int k) {
  if (G <= 0 || Ng <= 0 || k <= 0) return 0;
  const int64_t rows = (int64_t)G * Ng;
  const int64_t nblocks = pf_cdiv(rows + 1, kScanBlock);
  return (int64_t)(align256(sizeof(uint32_t) * (size_t)rows) + align256(sizeof(uint64_t) * (size_t)(nblocks + 1)) +
                   align256(sizeof(uint32_t) * (size_t)rows * (size_t)k));
}

int pf_knn_inverse(const int64_t* idx, int k, int G, int Ng, uint32_t* order, uint32_t* start, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1);
  const int64_t rows = (int64_t)G * Ng, pairs = rows * k;
  PF_REQUIRE(pairs < ((int64_t)1 << 32) - 1);
  if (pairs == 0) return PF_OK;
  PF_REQUIRE(idx && order && start && workspace && workspace_bytes >= pf_knn_inverse_workspace(G, Ng, k));
  hipStream_t s = (hipStream_t)stream;
  char* w = reinterpret_cast<char*>(workspace);
  uint32_t* cursor = reinterpret_cast<uint32_t*>(w);
  unsigned long long* chain = reinterpret_cast<unsigned long long*>(w + align256(sizeof(uint32_t) * (size_t)rows));
  const int64_t n = rows + 1;                                  // start[rows] = pairs closes the last list
  const int nblocks = (int)pf_cdiv(n, kScanBlock);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(w + align256(sizeof(uint32_t) * (size_t)rows) +
                                                  align256(sizeof(uint64_t) * (size_t)(nblocks + 1)));
  const unsigned pb = (unsigned)pf_cdiv(pairs, 256);
  hipLaunchKernelGGL(inverse_zero_kernel, dim3((unsigned)(pf_cdiv(n, 256) > 2048 ? 2048 : pf_cdiv(n, 256))), dim3(256), 0, s,
                     start, n, cursor, rows, reinterpret_cast<uint32_t*>(chain), 2 * ((int64_t)nblocks + 1));
  hipLaunchKernelGGL(inverse_count_kernel, dim3(pb), dim3(256), 0, s, idx, pairs, k, Ng, start);
  scan_in_place(start, n, nblocks, chain, s);
  hipLaunchKernelGGL(inverse_fill_kernel, dim3(pb), dim3(256), 0, s, idx, pairs, k, Ng, start, cursor, order);
  // lists ascending by pair id: one wave per list, rank counting in LDS (round 4; the one-thread-per-list insertion sort
  // took 300-430 us at 1.6 M pairs and was quadratic in the list length)
  hipLaunchKernelGGL(sort_lists_wave_kernel, dim3((unsigned)pf_cdiv(rows, 4)), dim3(256), 0, s, start, rows, order, scratch);
  return pf_launch_status();
}


int64_t pf_sort_pairs_workspace(int64_t pairs, int64_t nkeys) {
  if (pairs <= 0 || nkeys <= 0) return 0;
  const int64_t nblocks = pf_cdiv(nkeys + 1, kScanBlock);
  return (int64_t)(align256(sizeof(uint32_t) * (size_t)nkeys) + align256(sizeof(uint64_t) * (size_t)(nblocks + 1)) +
                   align256(sizeof(uint32_t) * (size_t)pairs));
}

int pf_sort_pairs_by_key(const uint32_t* keys, int64_t pairs, int64_t nkeys, uint32_t* order, uint32_t* start,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  PF_REQUIRE(pairs >= 0 && nkeys >= 1 && pairs < ((int64_t)1 << 32) - 1 && nkeys < ((int64_t)1 << 32) - 1);
  PF_REQUIRE(start && workspace && workspace_bytes >= pf_sort_pairs_workspace(pairs > 0 ? pairs : 1, nkeys));
  hipStream_t s = (hipStream_t)stream;
  char* w = reinterpret_cast<char*>(workspace);
  uint32_t* cursor = reinterpret_cast<uint32_t*>(w);
  unsigned long long* chain = reinterpret_cast<unsigned long long*>(w + align256(sizeof(uint32_t) * (size_t)nkeys));
  const int64_t n = nkeys + 1;
  const int nblocks = (int)pf_cdiv(n, kScanBlock);
  uint32_t* scratch = reinterpret_cast<uint32_t*>(w + align256(sizeof(uint32_t) * (size_t)nkeys) +
                                                  align256(sizeof(uint64_t) * (size_t)(nblocks + 1)));
  hipLaunchKernelGGL(inverse_zero_kernel, dim3((unsigned)(pf_cdiv(n, 256) > 2048 ? 2048 : pf_cdiv(n, 256))), dim3(256), 0, s,
                     start, n, cursor, nkeys, reinterpret_cast<uint32_t*>(chain), 2 * ((int64_t)nblocks + 1));
  if (pairs > 0) {
    PF_REQUIRE(keys && order);
    hipLaunchKernelGGL(keys_count_kernel, dim3((unsigned)pf_cdiv(pairs, 256)), dim3(256), 0, s, keys, pairs,
                       (uint32_t)nkeys, start);
  }
  scan_in_place(start, n, nblocks, chain, s);
  if (pairs > 0) {
    hipLaunchKernelGGL(keys_fill_kernel, dim3((unsigned)pf_cdiv(pairs, 256)), dim3(256), 0, s, keys, pairs,
                       (uint32_t)nkeys, start, cursor, order);
    hipLaunchKernelGGL(sort_lists_wave_kernel, dim3((unsigned)pf_cdiv(nkeys, 4)), dim3(256), 0, s, start, nkeys, order,
                       scratch);
  }
  return pf_launch_status();
}

}  // extern "C"
