// The block / wave machinery of tests/hipemu/hip/hip_runtime.h: persistent OS threads, one per GPU thread of a block,
// walking the grid's blocks one after the other.
#include <hip/hip_runtime.h>

#include <memory>

thread_local hipemu_uint3 threadIdx, blockIdx;
thread_local dim3 gridDim, blockDim;

namespace hipemu {
thread_local Wave* wave = nullptr;
thread_local int lane = 0;
static std::barrier<>* g_block_bar = nullptr;
void block_barrier() { g_block_bar->arrive_and_wait(); }

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  const int nwaves = (nthreads + 63) / 64;
  std::barrier<> block_bar(nthreads);
  g_block_bar = &block_bar;
  std::vector<std::unique_ptr<Wave>> waves;
  std::vector<std::unique_ptr<std::barrier<>>> wbars;
  for (int w = 0; w < nwaves; ++w) {
    const int lanes = (w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64;
    wbars.emplace_back(new std::barrier<>(lanes));
    waves.emplace_back(new Wave());
    waves.back()->bar = wbars.back().get();
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; ++t) {
    pool.emplace_back([&, t]() {
      threadIdx.x = (unsigned)(t % (int)block.x);
      threadIdx.y = (unsigned)((t / (int)block.x) % (int)block.y);
      threadIdx.z = (unsigned)(t / (int)(block.x * block.y));
      blockDim = block;
      gridDim = grid;
      wave = waves[t / 64].get();
      lane = t % 64;
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            blockIdx.x = bx;
            blockIdx.y = by;
            blockIdx.z = bz;
            body();
            block_bar.arrive_and_wait();      // the next block reuses the shared buffers
          }
    });
  }
  for (auto& th : pool) th.join();
  g_block_bar = nullptr;
}
}  // namespace hipemu

unsigned* pf_status_ptr() {
  static unsigned word = 0;
  return &word;
}
