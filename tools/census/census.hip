// Where do the workgroups of an under-filled grid land?  Every block records (XCC id, HW_ID) and start/end
// timestamps; the host prints blocks per (XCD, SE, CU) for a grid of `blocks` workgroups with `lds` bytes of LDS,
// each spinning for ~`spin_us` microseconds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void census(unsigned* out, long long spin_cycles) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
  }
  lds[threadIdx.x] = (float)threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_cycles) {
  }
  if (lds[(threadIdx.x + 1) & 255] < 0.0f) out[0] = 0;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 400;
  const int lds = argc > 2 ? atoi(argv[2]) : 35000;
  const long long spin = (argc > 3 ? atoll(argv[3]) : 20) * 100;   // clock64 ticks at 100 MHz
  unsigned* d;
  hipMalloc(&d, sizeof(unsigned) * 2 * blocks);
  if (lds > 65536) hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(census, dim3(blocks), dim3(256), lds, 0, d, spin);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(2 * blocks);
  hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * blocks, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  std::map<unsigned, int> per_xcc;
  for (int b = 0; b < blocks; ++b) {
    const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    per_xcc[xcc]++;
  }
  int hist[16] = {0};
  for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
  printf("blocks %d lds %d: kernel %.1f us; distinct CUs used %zu; XCDs %zu; blocks-per-CU histogram:", blocks, lds,
         ms * 1000.0f, per_cu.size(), per_xcc.size());
  for (int i = 1; i < 16; ++i)
    if (hist[i]) printf(" %dx%d", hist[i], i);
  printf("\n");
  return 0;
}
