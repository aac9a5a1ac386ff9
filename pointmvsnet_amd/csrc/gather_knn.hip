// Row G: gather_knn forward / backward (replaces reference functions/csrc/gather_knn_kernel.cu).
//
// Layout is the reference's: feature (B,C,N), index (B,N,K) int64, out (B,C,N,K).  One thread owns one
// (n,j) slot of the flattened N*K axis and walks the C channels, so the K-expanded tensor is written
// with fully coalesced stores (the bound: C*N*K*4 bytes of stores) while the neighbour reads hit L2.
// The backward comes in two forms: the same walk with hardware float atomics (global_atomic_add_f32/f64) into a
// zeroed (B,C,N) gradient, i.e. the reference's scatter (gather_knn_kernel.cu:50-89); or, given the inverted index
// lists of pf_knn_inverse (knn_inverse.hip), a GATHER: one thread per (b, c, m) sums the slots that name point m in
// ascending slot order and stores once -- no atomics, no zero-fill, bit-reproducible.
#include "pf_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(const T* __restrict__ x,
                                                         const int64_t* __restrict__ idx,
                                                         T* __restrict__ out, int C, int64_t N, int64_t NK,
                                                         unsigned* status) {
  const int64_t b = blockIdx.y;
  const int64_t nk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (nk >= NK) return;
  int64_t i = idx[b * NK + nk];
  const bool ok = (i >= 0) && (i < N);
  if (!ok) {
    atomicOr(status, PF_STATUS_BAD_INDEX);
    i = 0;
  }
  const T* xb = x + b * C * N + i;
  T* ob = out + b * C * NK + nk;
  int c = 0;
  for (; c + 4 <= C; c += 4) {
    const T v0 = xb[(int64_t)(c + 0) * N];
    const T v1 = xb[(int64_t)(c + 1) * N];
    const T v2 = xb[(int64_t)(c + 2) * N];
    const T v3 = xb[(int64_t)(c + 3) * N];
    ob[(int64_t)(c + 0) * NK] = ok ? v0 : T(0);
    ob[(int64_t)(c + 1) * NK] = ok ? v1 : T(0);
    ob[(int64_t)(c + 2) * NK] = ok ? v2 : T(0);
    ob[(int64_t)(c + 3) * NK] = ok ? v3 : T(0);
  }
  for (; c < C; ++c) ob[(int64_t)c * NK] = ok ? xb[(int64_t)c * N] : T(0);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_kernel(const T* __restrict__ gout,
                                                         const int64_t* __restrict__ idx,
                                                         T* __restrict__ gin, int C, int64_t N, int64_t NK,
                                                         unsigned* status) {
  const int64_t b = blockIdx.y;
  const int64_t nk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (nk >= NK) return;
  const int64_t i = idx[b * NK + nk];
  if (i < 0 || i >= N) {
    atomicOr(status, PF_STATUS_BAD_INDEX);
    return;
  }
  const T* gb = gout + b * C * NK + nk;
  T* ib = gin + b * C * N + i;
  for (int c = 0; c < C; ++c) unsafeAtomicAdd(ib + (int64_t)c * N, gb[(int64_t)c * NK]);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_inverse_kernel(const T* __restrict__ gout,
                                                                 const int64_t* __restrict__ idx,
                                                                 const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ start,
                                                                 T* __restrict__ gin, int C, int64_t N, int64_t NK,
                                                                 unsigned* status) {
  const int64_t b = blockIdx.z, c = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= N) return;
  const uint32_t t0 = start[b * N + m], t1 = start[b * N + m + 1];
  const T* gb = gout + (b * C + c) * NK - b * NK;      // order holds pair ids over all batches: b * NK + slot
  T acc = T(0);
  for (uint32_t t = t0; t < t1; ++t) {
    const uint32_t p = order[t];
    if (idx[p] == m) acc += gb[p];                     // (an out-of-range index was filed under the clamped row:
    else if (c == 0) atomicOr(status, PF_STATUS_BAD_INDEX);   //  the forward wrote zeros for it, it carries no gradient)
  }
  gin[(b * C + c) * N + m] = acc;
}

template <typename T>
int gather_forward(const T* feature, const int64_t* index, T* out, int64_t B, int64_t C, int64_t N, int64_t K,
                   void* stream) {
  PF_REQUIRE(B >= 0 && C >= 0 && N >= 0 && K >= 0);
  PF_REQUIRE(B <= 65535 && C <= INT32_MAX);
  if (B == 0 || C == 0 || N == 0 || K == 0) return PF_OK;
  PF_REQUIRE(feature != nullptr && index != nullptr && out != nullptr);
  unsigned* status = pf_status_ptr();
  PF_REQUIRE(status != nullptr);
  const int64_t NK = N * K;
  dim3 grid((unsigned)pf_cdiv(NK, 256), (unsigned)B);
  hipLaunchKernelGGL(gather_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, feature, index, out, (int)C, N,
                     NK, status);
  return pf_launch_status();
}

template <typename T>
int gather_backward(const T* grad_out, const int64_t* index, T* grad_in, int64_t B, int64_t C, int64_t N,
                    int64_t K, const uint32_t* inv_order, const uint32_t* inv_start, void* stream) {
  PF_REQUIRE(B >= 0 && C >= 0 && N >= 0 && K >= 0);
  PF_REQUIRE(B <= 65535 && C <= INT32_MAX && (inv_order == nullptr) == (inv_start == nullptr));
  if (B == 0 || C == 0 || N == 0) return PF_OK;
  PF_REQUIRE(grad_in != nullptr);
  if (inv_order != nullptr && K > 0) {
    PF_REQUIRE(grad_out != nullptr && index != nullptr && C <= 65535);
    unsigned* status = pf_status_ptr();
    PF_REQUIRE(status != nullptr);
    dim3 grid((unsigned)pf_cdiv(N, 256), (unsigned)C, (unsigned)B);
    hipLaunchKernelGGL(gather_bwd_inverse_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, grad_out, index, inv_order,
                       inv_start, grad_in, (int)C, N, N * K, status);
    return pf_launch_status();
  }
  {
    const int zrc = pf_zero_async(grad_in, sizeof(T) * (size_t)(B * C * N), (hipStream_t)stream);
    if (zrc != PF_OK) return zrc;
  }
  if (K == 0) return PF_OK;
  PF_REQUIRE(grad_out != nullptr && index != nullptr);
  unsigned* status = pf_status_ptr();
  PF_REQUIRE(status != nullptr);
  const int64_t NK = N * K;
  dim3 grid((unsigned)pf_cdiv(NK, 256), (unsigned)B);
  hipLaunchKernelGGL(gather_bwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, grad_out, index, grad_in,
                     (int)C, N, NK, status);
  return pf_launch_status();
}

}  // namespace

extern "C" {

int pf_gather_knn_forward_f32(const float* feature, const int64_t* index, float* out, int64_t B, int64_t C,
                              int64_t N, int64_t K, void* stream) {
  return gather_forward<float>(feature, index, out, B, C, N, K, stream);
}
int pf_gather_knn_forward_f64(const double* feature, const int64_t* index, double* out, int64_t B, int64_t C,
                              int64_t N, int64_t K, void* stream) {
  return gather_forward<double>(feature, index, out, B, C, N, K, stream);
}
int pf_gather_knn_backward_f32(const float* grad_out, const int64_t* index, float* grad_in, int64_t B, int64_t C,
                               int64_t N, int64_t K, const uint32_t* inv_order, const uint32_t* inv_start,
                               void* stream) {
  return gather_backward<float>(grad_out, index, grad_in, B, C, N, K, inv_order, inv_start, stream);
}
int pf_gather_knn_backward_f64(const double* grad_out, const int64_t* index, double* grad_in, int64_t B, int64_t C,
                               int64_t N, int64_t K, const uint32_t* inv_order, const uint32_t* inv_start,
                               void* stream) {
  return gather_backward<double>(grad_out, index, grad_in, B, C, N, K, inv_order, inv_start, stream);
}

}  // extern "C"
