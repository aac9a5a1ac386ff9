// How fast is v_pk_fma_f32 on gfx950, and does a scalar-register (SGPR pair) source or op_sel slow it down?
// (round 6: the direct tower kernel reached a quarter of the packed-f32 peak; this separates the instruction's own rate
// from the kernel's operand traffic.)   hipcc --offload-arch=gfx950 -O3 pk_fma_rate.hip -o pk_fma_rate && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ACC = 16, ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* w, float seed) {
  f2 acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = (f2){seed * i, seed + i};
  f2 a = {seed + threadIdx.x, seed - threadIdx.x};
  const f2 ws = *reinterpret_cast<const f2*>(w + (blockIdx.x & 1) * 2);       // wave-uniform -> SGPR pair
  f2 wv = {ws[0] + threadIdx.x * 1e-9f, ws[1]};                                  // VGPR pair
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(wv));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(ws));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(a), "s"(ws));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(wv));
      if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a[0]), "v"(wv[0]));
      if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a[0]), "s"(ws[0]));
      if (MODE == 6) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) float*>(&acc[i & ~1])) : "v"(a[0]), "v"(wv[0]));
    }
  }
  float s = 0;
  for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what, double flop_per_instr_lane, int waves_per_simd) {
  float *out, *w;
  const int blocks = 256 * waves_per_simd;                 // 4 waves per block = 1 per SIMD per block
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&w, 64);
  hipMemset(w, 0, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, w, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(out, w, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double instr = (double)blocks * 4 * ITER * ACC;    // wave instructions
  const double cyc_per_instr = ms * 1e-3 * 2.4e9 * 1024 / instr;   // SIMD-cycles per wave instruction at 2.4 GHz, 1024 SIMDs
  printf("%-58s %d waves/SIMD: %7.3f ms, %5.2f SIMD-cycles per instruction, %6.1f TFLOP/s\n", what, waves_per_simd, ms,
         cyc_per_instr, instr * 64 * flop_per_instr_lane / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(w);
}

int main() {
  for (int wps : {1, 2, 4}) {
    run<0>("v_pk_fma_f32 v, v, v", 4, wps);
    run<1>("v_pk_fma_f32 v, v, s[pair]", 4, wps);
    run<2>("v_pk_fma_f32 v, v, s[pair] op_sel:[1,0,0] op_sel_hi:[1,1,1]", 4, wps);
    run<3>("v_pk_fma_f32 v, v, v op_sel:[0,0,0] op_sel_hi:[0,1,1]", 4, wps);
    run<4>("v_fma_f32 v, v, v", 2, wps);
    run<5>("v_fma_f32 v, v, s", 2, wps);
    run<6>("v_mfma_f32_16x16x4_f32", 2.0 * 16 * 16 * 4 / 64, wps);
  }
  return 0;
}
