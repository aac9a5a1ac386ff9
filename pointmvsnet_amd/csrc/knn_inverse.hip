// The inverse of a neighbour index tensor: for every point, WHO gathers it.
//
// The backward of a neighbour gather (reference functions/csrc/gather_knn_kernel.cu:50-89) is a scatter-add:
// grad[idx[n, j]] += g[n, j], done there -- and in rounds 1-2 here -- with float atomics, i.e. in arrival order:
// two runs of the same step give different low bits.  With the pairs sorted by target the scatter becomes a GATHER:
// point m sums the pairs that name it, in ascending pair order (a stable sort keeps them so), with plain stores --
// bit-reproducible and no zero-fill of the output.  One inversion serves every backward pass that shares the index
// tensor (the three EdgeConv layers of a PointFlow iteration).
//
//   keys[p]  = g * Ng + clamp(idx[g, n, j], 0, Ng - 1)        p = (g * Ng + n) * k + j      (the forward's clamp)
//   order    = the pair ids p sorted by key (rocPRIM radix sort: stable)
//   start[m] = first position in `order` whose key is >= m, m in [0, G * Ng]  ->  pairs of m: order[start[m] .. start[m+1])
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "pf_common.h"

namespace {

__global__ __launch_bounds__(256) void inverse_keys_kernel(const int64_t* __restrict__ idx, int64_t pairs, int k, int Ng,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  const int64_t row = p / k;
  const int64_t g = row / Ng;
  int64_t i = idx[p];
  i = i < 0 ? 0 : (i > Ng - 1 ? Ng - 1 : i);
  keys[p] = (uint32_t)(g * Ng + i);
  vals[p] = (uint32_t)p;
}

// start[m] for every m in (key[t-1], key[t]] is t; one thread per sorted position (runs of equal keys: one writer)
__global__ __launch_bounds__(256) void inverse_starts_kernel(const uint32_t* __restrict__ sorted, int64_t pairs, int64_t rows,
                                                             uint32_t* __restrict__ start) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t > pairs) return;
  const int64_t lo = t == 0 ? -1 : (int64_t)sorted[t - 1];
  const int64_t hi = t == pairs ? rows : (int64_t)sorted[t];          // start[rows] = pairs closes the last list
  for (int64_t m = lo + 1; m <= hi; ++m) start[m] = (uint32_t)t;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

int sort_bits(int64_t rows) {
  int bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < rows) ++bits;
  return bits;
}

}  // namespace

extern "C" {

int64_t pf_knn_inverse_workspace(int G, int Ng, int k) {
  if (G <= 0 || Ng <= 0 || k <= 0) return 0;
  const int64_t pairs = (int64_t)G * Ng * k;
  size_t tmp = 0;
  uint32_t* nul = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)pairs, 0, sort_bits((int64_t)G * Ng),
                                (hipStream_t) nullptr) != hipSuccess)
    return -1;
  return (int64_t)(3 * align256(sizeof(uint32_t) * (size_t)pairs) + align256(tmp));
}

int pf_knn_inverse(const int64_t* idx, int k, int G, int Ng, uint32_t* order, uint32_t* start, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  PF_REQUIRE(G >= 0 && Ng >= 0 && k >= 1);
  const int64_t rows = (int64_t)G * Ng, pairs = rows * k;
  PF_REQUIRE(pairs < ((int64_t)1 << 32) - 1);
  if (pairs == 0) return PF_OK;
  PF_REQUIRE(idx && order && start && workspace && workspace_bytes >= pf_knn_inverse_workspace(G, Ng, k));
  hipStream_t s = (hipStream_t)stream;
  const size_t arr = align256(sizeof(uint32_t) * (size_t)pairs);
  char* w = reinterpret_cast<char*>(workspace);
  uint32_t* keys = reinterpret_cast<uint32_t*>(w);
  uint32_t* sorted = reinterpret_cast<uint32_t*>(w + arr);
  uint32_t* vals = reinterpret_cast<uint32_t*>(w + 2 * arr);
  void* tmp = w + 3 * arr;
  size_t tmp_bytes = (size_t)workspace_bytes - 3 * arr;
  hipLaunchKernelGGL(inverse_keys_kernel, dim3((unsigned)pf_cdiv(pairs, 256)), dim3(256), 0, s, idx, pairs, k, Ng, keys,
                     vals);
  PF_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, sorted, vals, order, (size_t)pairs, 0, sort_bits(rows), s));
  hipLaunchKernelGGL(inverse_starts_kernel, dim3((unsigned)pf_cdiv(pairs + 1, 256)), dim3(256), 0, s, sorted, pairs, rows,
                     start);
  return pf_launch_status();
}

}  // extern "C"
