// Does a CDNA4 SIMD overlap matrix (MFMA) and vector (VALU) instructions of DIFFERENT waves?
// One 512-thread block per CU: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (one per SIMD) issue VALU FMAs.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 1: MFMA waves only work, 2: VALU waves only, 3: both; 4: both kinds of work in EVERY wave, interleaved
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float r = 0.0f;
  if (MODE == 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    const float x = seed + threadIdx.x, yv = seed * 0.5f;
    for (int i = 0; i < iters / 2; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a0, 0, 0, 0);
      v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a1, 0, 0, 0);
      v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a2, 0, 0, 0);
      v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a3, 0, 0, 0);
      v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  } else if (mf) {
    if (MODE & 1) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      const float x = seed + threadIdx.x, yv = seed * 0.5f;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, yv, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (MODE & 2) {
      float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
      for (int i = 0; i < iters; ++i) {     // 32 independent-ish VALU FMAs per iteration: 128 cycles at 4 cycles each
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
          v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
        }
      }
      r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.0f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;   // 80 000 MFMAs of 32 cycles = 2.56 M cycles ~ 1.07 ms at 2.4 GHz; VALU: 640 000 FMAs x 4 cycles = the same
  printf("per SIMD: one MFMA wave (4 x %d v_mfma_f32_16x16x4_f32) and one VALU wave (32 x %d v_fma_f32)\n", iters, iters);
  printf("MFMA waves only : %8.1f us\n", run<1>(out, iters));
  printf("VALU waves only : %8.1f us\n", run<2>(out, iters));
  printf("both            : %8.1f us\n", run<3>(out, iters));
  printf("interleaved in every wave (half the iterations, 8 waves): %8.1f us\n", run<4>(out, iters));
  return 0;
}
