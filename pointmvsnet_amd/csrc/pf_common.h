// Shared host/device helpers for libpointflow_hip.so (gfx950 only; no CUDA paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pointflow_hip.h"

#define PF_WAVE 64

// Per-device sticky status word (bad neighbour index etc.); allocated on first use.
unsigned* pf_status_ptr();

#define PF_REQUIRE(cond)                 \
  do {                                   \
    if (!(cond)) return PF_ERR_INVALID_ARG; \
  } while (0)

#define PF_HIP(expr)                        \
  do {                                      \
    hipError_t _e = (expr);                 \
    if (_e != hipSuccess) return (int)_e;   \
  } while (0)

static inline int pf_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PF_OK : (int)e;
}

static inline int64_t pf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: `flags` is one bit per device ordinal, owned by the
// caller (one static word per kernel instantiation), so a process that drives several GPUs opts in on each of them.
#include <atomic>
static inline int pf_allow_big_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& flags) {
  int dev = 0;
  PF_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (flags.load(std::memory_order_acquire) & bit) return PF_OK;
  PF_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  flags.fetch_or(bit, std::memory_order_release);
  return PF_OK;
}

// Zero `bytes` bytes (a multiple of 4, 4-byte aligned) with a KERNEL.  The scatter-add backward passes need their output
// cleared first; hipMemsetAsync did that until the training step was captured in a hipGraph: memset nodes recorded
// from the autograd thread were not replayed reliably (the cleared buffers kept the previous replay's sums and the
// gradients grew from replay to replay, tools/dbg_train.py), a kernel node is.
#if defined(__HIPCC__)
static __global__ __launch_bounds__(256) void pf_zero_kernel(unsigned* __restrict__ p, size_t words) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
static inline int pf_zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return PF_OK;
  const size_t words = bytes / 4;
  size_t blocks = (words + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pf_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<unsigned*>(p), words);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PF_OK : (int)e;
}
#endif

// ---- XCD-aware block order ------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin in dispatch order (linear block id % 8) and every XCD has its own 4 MB
// L2.  With tile = block, the eight neighbours of a tile -- which read the same halo rows / gather the same point rows --
// sit on eight different L2s.  pf_xcd_chunk relabels the blocks so that XCD x owns the x-th contiguous eighth of the
// launch's tiles: neighbours share an L2.  A pure relabelling (a bijection of the grid): results are bit-identical as long
// as everything a kernel derives from blockIdx -- the BatchNorm partial row included -- comes from the relabelled ids.
// PF_XCD is a bit mask of kernel families (tools/experiments/build_xcd_variants.sh: A/B builds); 0 = tile = block
// everywhere.  Measured same-box on the headline (profiles/r06t_xcd_block_order.md): towers +1.0 %, conv3d -0.6 %, flow
// feature assembly +-0, the coarse warp by pixel-row bands (bit 32) +0.4 % inside the spread -- so only the towers take it by default.  (The EdgeConv gather passes use their own band order,
// csrc/edgeconv.hip: xcd_tile, +2.3 %.)
#ifndef PF_XCD
#define PF_XCD 1
#endif
#define PF_XCD_TOWER 1
#define PF_XCD_CONV3D 2
#define PF_XCD_FETCH 4
#define PF_XCD_KNN 8
#define PF_XCD_EDGE 16
#define PF_XCD_FRUSTUM 32
#ifdef __HIPCC__
__device__ __forceinline__ unsigned pf_xcd_chunk(unsigned lin, unsigned total) {
  const unsigned per = total >> 3, rem = total & 7u;
  const unsigned x = lin & 7u, j = lin >> 3;
  return x * per + (x < rem ? x : rem) + j;
}
// Band order for point tiles of a D x (H x W) volume (tile = `unit` consecutive points, x fastest, then y, then d): XCD x owns
// the x-th eighth of EVERY plane's tiles -- a band of pixel rows at all depths -- instead of every eighth tile.  Needs whole
// tiles per plane and a multiple of 8 of them (else: identity).
__device__ __forceinline__ unsigned pf_xcd_band(unsigned b, unsigned tiles_per_plane) {
  if ((tiles_per_plane & 7u) != 0u) return b;
  const unsigned band = tiles_per_plane >> 3;
  const unsigned x = b & 7u, j = b >> 3;
  const unsigned d = j / band, o = j - d * band;
  return d * tiles_per_plane + x * band + o;
}
// the relabelled (blockIdx.x, blockIdx.y) of a 2-D grid (x fastest in dispatch order)
template <int FAMILY>
__device__ __forceinline__ void pf_xcd_xy(unsigned& bx, unsigned& by) {
  if ((PF_XCD & FAMILY) == 0) {
    bx = blockIdx.x;
    by = blockIdx.y;
    return;
  }
  const unsigned gx = gridDim.x;
  const unsigned L = pf_xcd_chunk(blockIdx.y * gx + blockIdx.x, gx * gridDim.y);
  by = L / gx;
  bx = L - by * gx;
}
#endif

// ---- projection + bilinear taps shared by the fetch kernels -------------------------------------
// Follows reference utils/feature_fetcher.py:36-55 on the arithmetic of ATen's CPU grid_sample with
// align_corners=True (the oracle): un-normalise with (g + 1) * ((size - 1) / 2), weights
// nw = (1-wy)(1-wx) ..., out = ((nw*a + ne*b) + sw*c) + se*d.  Compiled with -ffp-contract=off so the
// products/sums below stay separate unless written as fmaf.
struct PfTaps {
  int off[4];     // y*W + x of the nw, ne, sw, se taps; 0 (a valid address) when the tap is outside the map
  float wgt[4];   // bilinear weights; exactly 0 for a tap outside the map (zero padding)
  bool ok[4];     // tap inside the map (only the backward scatter needs it)
  int xi, yi;     // column / row of the nw tap (tap k sits at (xi + (k & 1), yi + (k >> 1)) when ok[k])
  float fx, fy;   // fractional parts: wgt = {(1-fy)(1-fx), (1-fy) fx, fy (1-fx), fy fx} where ok
};

__device__ __forceinline__ void pf_project_taps(float X, float Y, float Z, const float* __restrict__ Kv,
                                                const float* __restrict__ Ev, int H, int W, PfTaps& t) {
  float px = X, py = Y, pz = Z;
  if (Ev != nullptr) {
    px = fmaf(Ev[2], Z, fmaf(Ev[1], Y, Ev[0] * X)) + Ev[3];
    py = fmaf(Ev[6], Z, fmaf(Ev[5], Y, Ev[4] * X)) + Ev[7];
    pz = fmaf(Ev[10], Z, fmaf(Ev[9], Y, Ev[8] * X)) + Ev[11];
  }
  const float nx = px / pz;
  const float ny = py / pz;
  const float u = fmaf(Kv[2], 1.0f, fmaf(Kv[1], ny, Kv[0] * nx));
  const float v = fmaf(Kv[5], 1.0f, fmaf(Kv[4], ny, Kv[3] * nx));
  const float gx = ((u - 0.5f) / (float)(W - 1)) * 2.0f - 1.0f;
  const float gy = ((v - 0.5f) / (float)(H - 1)) * 2.0f - 1.0f;
  const float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
  const float iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
  const float x0 = floorf(ix), y0 = floorf(iy);
  const float wx1 = ix - x0, wy1 = iy - y0;
  const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const float xmax = (float)(W - 1), ymax = (float)(H - 1);
  const bool vx0 = (x0 >= 0.0f) && (x0 <= xmax);
  const bool vx1 = (x0 + 1.0f >= 0.0f) && (x0 + 1.0f <= xmax);
  const bool vy0 = (y0 >= 0.0f) && (y0 <= ymax);
  const bool vy1 = (y0 + 1.0f >= 0.0f) && (y0 + 1.0f <= ymax);
  const int xi = (vx0 || vx1) ? (int)x0 : 0;
  const int yi = (vy0 || vy1) ? (int)y0 : 0;
  t.xi = xi;
  t.yi = yi;
  t.fx = wx1;
  t.fy = wy1;
  t.ok[0] = vy0 && vx0;
  t.ok[1] = vy0 && vx1;
  t.ok[2] = vy1 && vx0;
  t.ok[3] = vy1 && vx1;
  t.off[0] = t.ok[0] ? yi * W + xi : 0;
  t.off[1] = t.ok[1] ? yi * W + xi + 1 : 0;
  t.off[2] = t.ok[2] ? (yi + 1) * W + xi : 0;
  t.off[3] = t.ok[3] ? (yi + 1) * W + xi + 1 : 0;
  // zero weight == the reference's masked gather (value 0) times the weight: the term is +-0 either way.
  // (Feature maps are finite activations; 0 * inf is not a case the reference can produce a number for.)
  t.wgt[0] = t.ok[0] ? wy0 * wx0 : 0.0f;
  t.wgt[1] = t.ok[1] ? wy0 * wx1 : 0.0f;
  t.wgt[2] = t.ok[2] ? wy1 * wx0 : 0.0f;
  t.wgt[3] = t.ok[3] ? wy1 * wx1 : 0.0f;
}

// Unconditional loads (outside taps read element 0 with weight 0): the compiler can issue the taps of
// several views / channels back to back instead of one exec-masked branch per tap.
__device__ __forceinline__ float pf_sample(const float* __restrict__ plane, const PfTaps& t) {
  const float a = plane[t.off[0]];
  const float b = plane[t.off[1]];
  const float c = plane[t.off[2]];
  const float d = plane[t.off[3]];
  return ((a * t.wgt[0] + b * t.wgt[1]) + c * t.wgt[2]) + d * t.wgt[3];
}
