// A HIP-on-CPU shim for TESTS (tests/hipemu): enough of the HIP programming model to run this repository's kernels --
// unchanged source, recompiled for the host with clang++ -- on a machine without a GPU, one OS thread per GPU thread.
//
// Why: round 5 lost its GPU access while a new kernel (conv2d_wide_split_kernel) was being written.  hipcc -S shows that
// a kernel compiles, not that its indexing is right; this shim executes it.  What it models:
//   * a launch = the blocks of the grid dealt to a few OS threads; inside a block every GPU thread is a FIBER of its
//     block's OS thread, switched at __syncthreads() (a barrier over the block's fibers) and at the wave collectives;
//     threadIdx / blockIdx / gridDim / blockDim are thread-local and set by the scheduler when it resumes a fiber;
//   * `extern __shared__` / `__shared__` storage is one `static thread_local` buffer per declaration and OS thread;
//   * wave-collective operations (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x16_bf16, __shfl_xor)
//     rendezvous the 64 threads of a wave: every lane publishes its operands, the lanes meet, every lane computes its own
//     registers from the published operands with the lane <-> element maps of the CDNA4 ISA (A: row = lane % 32,
//     k = 8 * (lane / 32) + j for 32x32x16; C/D: col = lane % 32, row = (r % 4) + 8 * (r / 4) + 4 * (lane / 32)), f32
//     products accumulated as an fmaf chain in k order -- the documented behaviour of the f32 forms; for the bf16 form the
//     order inside a 16-deep block is the emulator's choice (the hardware's is unspecified).
// What it does NOT model: timing, bank conflicts, occupancy, memory faults (an out-of-bounds access is a host
// out-of-bounds access: run the tests under -fsanitize=address to catch those), divergent collectives.
#pragma once
#include <atomic>
#include <cstdio>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __constant__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct hipemu_uint3 { unsigned x, y, z; };
extern thread_local hipemu_uint3 threadIdx, blockIdx;
extern thread_local dim3 gridDim, blockDim;

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
constexpr hipError_t hipErrorNotInitialized = 3;
constexpr int hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToDevice = 3;
struct hipDeviceProp_t {
  int multiProcessorCount = 256;
  size_t sharedMemPerBlock = 65536;
  char gcnArchName[32] = "gfx950 (tests/hipemu)";
};
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu error"; }
template <class T>
static inline hipError_t hipMalloc(T** p, size_t n) { *p = reinterpret_cast<T*>(::operator new(n)); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline long long wall_clock64() { return 0; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct longlong2 { long long x, y; };
struct ulonglong2 { unsigned long long x, y; };
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline longlong2 make_longlong2(long long a, long long b) { return longlong2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

// ---- the block / wave machinery (tests/hipemu/hipemu.cpp) -------------------------------------------------------------
namespace hipemu {
struct Wave {                       // the rendezvous area of one wave: what every lane publishes for a collective
  alignas(16) unsigned char a[64][64];
  alignas(16) unsigned char b[64][64];
};
extern thread_local Wave* wave;     // this thread's wave
extern thread_local int lane;       // 0..63
void block_barrier();
void wave_barrier();                // the 64 lanes of this thread's wave meet
void launch(dim3 grid, dim3 block, size_t dynamic_lds, const std::function<void()>& body);
}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  ::hipemu::launch((grid), (block), (size_t)(lds), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void hipemu_wave_barrier() { hipemu::wave_barrier(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
template <class To, class From>
static inline To hipemu_bits(From v) { static_assert(sizeof(To) == sizeof(From)); To o; std::memcpy(&o, &v, sizeof(To)); return o; }
static inline unsigned __float_as_uint(float v) { return hipemu_bits<unsigned>(v); }
static inline int __float_as_int(float v) { return hipemu_bits<int>(v); }
static inline float __uint_as_float(unsigned v) { return hipemu_bits<float>(v); }
static inline float __int_as_float(int v) { return hipemu_bits<float>(v); }
static inline double __longlong_as_double(long long v) { return hipemu_bits<double>(v); }
static inline long long __double_as_longlong(double v) { return hipemu_bits<long long>(v); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_s_barrier() hipemu::block_barrier()
template <class F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 2; return hipSuccess; }
// atomics on global / shared memory: the threads of a block really run concurrently here
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
  float old = *p, want;
  do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
#define unsafeAtomicAdd atomicAdd
static inline double atomicAdd(double* p, double v) {
  double old = *p, want;
  do { want = old + v; } while (!__atomic_compare_exchange(p, &old, &want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

template <class T>
static inline T __shfl_xor(T v, int mask, int = 64) {
  static_assert(sizeof(T) <= 64, "shuffle payload");
  std::memcpy(hipemu::wave->a[hipemu::lane], &v, sizeof(T));
  hipemu::wave_barrier();
  T out;
  std::memcpy(&out, hipemu::wave->a[hipemu::lane ^ mask], sizeof(T));
  hipemu::wave_barrier();
  return out;
}

template <class T>
static inline T hipemu_shfl_from(T v, int src) {
  std::memcpy(hipemu::wave->a[hipemu::lane], &v, sizeof(T));
  hipemu::wave_barrier();
  T out;
  std::memcpy(&out, hipemu::wave->a[src & 63], sizeof(T));
  hipemu::wave_barrier();
  return out;
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  const int base = hipemu::lane & ~(width - 1);
  return hipemu_shfl_from(v, base + (src & (width - 1)));
}
// wave votes (round 6: the inverted-list gather of the EdgeConv backward): every lane of the wave takes part
static inline unsigned long long __ballot(int pred) {
  const int mine = pred ? 1 : 0;
  std::memcpy(hipemu::wave->a[hipemu::lane], &mine, sizeof(int));
  hipemu::wave_barrier();
  unsigned long long out = 0ull;
  for (int i = 0; i < 64; ++i) {
    int v;
    std::memcpy(&v, hipemu::wave->a[i], sizeof(int));
    if (v) out |= 1ull << i;
  }
  hipemu::wave_barrier();
  return out;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) { return __ballot(!pred) == 0ull; }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }

template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  const int self = hipemu::lane, pos = self & (width - 1);
  return hipemu_shfl_from(v, pos + (int)delta < width ? self + (int)delta : self);
}
template <class T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  const int self = hipemu::lane, pos = self & (width - 1);
  return hipemu_shfl_from(v, pos >= (int)delta ? self - (int)delta : self);
}

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));

// v_mfma_f32_32x32x2_f32: A[i = lane % 32][k = lane / 32], B[k = lane / 32][j = lane % 32]; D += A B as an fmaf chain in k
static inline hipemu_f32x16 hipemu_mfma_32x32x2_f32(float a, float b, hipemu_f32x16 c) {
  using namespace hipemu;
  std::memcpy(wave->a[lane], &a, 4);
  std::memcpy(wave->b[lane], &b, 4);
  wave_barrier();
  const int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      std::memcpy(&av, wave->a[row + 32 * k], 4);
      std::memcpy(&bv, wave->b[col + 32 * k], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_16x16x4_f32: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16]; C/D: col = lane % 16,
// row = 4 * (lane / 16) + r
static inline hipemu_f32x4 hipemu_mfma_16x16x4_f32(float a, float b, hipemu_f32x4 c) {
  using namespace hipemu;
  std::memcpy(wave->a[lane], &a, 4);
  std::memcpy(wave->b[lane], &b, 4);
  wave_barrier();
  const int col = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      std::memcpy(&av, wave->a[row + 16 * k], 4);
      std::memcpy(&bv, wave->b[col + 16 * k], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_32x32x16_bf16: A[i = lane % 32][k = 8 (lane / 32) + j], B[k = 8 (lane / 32) + j][col = lane % 32], j = 0..7
static inline float hipemu_bf16_bits_to_float(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c) {
  using namespace hipemu;
  std::memcpy(wave->a[lane], &a, 16);
  std::memcpy(wave->b[lane], &b, 16);
  wave_barrier();
  const int col = lane & 31, hi = lane >> 5;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int kh = 0; kh < 2; ++kh) {
      uint16_t av[8], bv[8];
      std::memcpy(av, wave->a[row + 32 * kh], 16);
      std::memcpy(bv, wave->b[col + 32 * kh], 16);
      for (int j = 0; j < 8; ++j) acc = fmaf(hipemu_bf16_bits_to_float(av[j]), hipemu_bf16_bits_to_float(bv[j]), acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_32x32x2_f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_16x16x4_f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_32x32x16_bf16((a), (b), (c))
