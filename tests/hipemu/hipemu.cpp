// The block / wave machinery of tests/hipemu/hip/hip_runtime.h.
//
// A GPU thread is a FIBER (ucontext): the 256 threads of a block are 256 fibers of one OS thread, switched only where the
// programming model synchronises -- __syncthreads() (a block barrier) and the wave-collective instructions (a rendezvous
// of the 64 lanes of a wave).  A fiber that waits yields to the scheduler, which resumes the next fiber that can run; a
// collective therefore costs ~64 context switches (microseconds), not 64 OS-thread wake-ups (milliseconds).  The blocks
// of a grid are independent: they are dealt to HIPEMU_THREADS OS threads (default: the machine's cores, at most 16), each
// with its own fibers, its own copy of every __shared__ array (`static thread_local`) and its own dynamic-LDS buffer.
#include <hip/hip_runtime.h>

#include <ucontext.h>

#include <cstdlib>
#include <memory>
#include <mutex>

thread_local hipemu_uint3 threadIdx, blockIdx;
thread_local dim3 gridDim, blockDim;

namespace hipemu {
thread_local Wave* wave = nullptr;
thread_local int lane = 0;

namespace {
constexpr size_t kStack = 256 * 1024;

struct Fiber {
  ucontext_t ctx;
  bool done = false;
  const volatile unsigned* wait_ptr = nullptr;   // blocked while *wait_ptr == wait_val
  unsigned wait_val = 0;
};

struct Worker {                       // everything one OS thread needs to run blocks
  int nthreads = 0;
  std::vector<Fiber> fibers;
  std::vector<char> stacks;
  std::vector<Wave> waves;
  std::vector<unsigned> wave_gen, wave_cnt, wave_size;
  unsigned block_gen = 0, block_cnt = 0;
  ucontext_t sched;
  int cur = 0;
  const std::function<void()>* body = nullptr;
  dim3 block;
  std::vector<int> visit;            // HIPEMU_ORDER's permutation of the block's threads
};
thread_local Worker* W = nullptr;
thread_local float* g_dynamic_lds = nullptr;   // this OS thread's dynamic-LDS block: EXACTLY the bytes the launch asked for

void trampoline() {
  Worker* w = W;
  (*w->body)();
  w->fibers[w->cur].done = true;
  swapcontext(&w->fibers[w->cur].ctx, &w->sched);
}

void wait_until_changed(const volatile unsigned* p, unsigned v) {
  Worker* w = W;
  Fiber& f = w->fibers[w->cur];
  f.wait_ptr = p;
  f.wait_val = v;
  swapcontext(&f.ctx, &w->sched);      // the scheduler resumes this fiber only after *p != v
  f.wait_ptr = nullptr;
}

void run_block(Worker* w) {
  const int n = w->nthreads;
  w->block_gen = w->block_cnt = 0;
  std::fill(w->wave_gen.begin(), w->wave_gen.end(), 0u);
  std::fill(w->wave_cnt.begin(), w->wave_cnt.end(), 0u);
  for (int t = 0; t < n; ++t) {
    Fiber& f = w->fibers[t];
    f.done = false;
    f.wait_ptr = nullptr;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = w->stacks.data() + (size_t)t * kStack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, trampoline, 0);
  }
  int live = n;
  while (live > 0) {
    bool progressed = false;
    for (int v = 0; v < n; ++v) {
      // HIPEMU_ORDER: the order in which runnable fibers are visited -- 0 ascending (default), 1 waves descending (lanes
      // ascending), 2 everything descending.  The hardware promises no order between waves: a kernel without a race
      // (and without floating-point atomics) gives bit-identical results under all three
      const int t = w->visit[v];
      Fiber& f = w->fibers[t];
      if (f.done) continue;
      if (f.wait_ptr != nullptr && *f.wait_ptr == f.wait_val) continue;
      w->cur = t;
      threadIdx.x = (unsigned)(t % (int)w->block.x);
      threadIdx.y = (unsigned)((t / (int)w->block.x) % (int)w->block.y);
      threadIdx.z = (unsigned)(t / (int)(w->block.x * w->block.y));
      wave = &w->waves[t / 64];
      lane = t % 64;
      swapcontext(&w->sched, &f.ctx);
      progressed = true;
      if (f.done) --live;
    }
    if (!progressed) {                  // every live fiber waits on a barrier that cannot complete: a divergent barrier
      std::fprintf(stderr, "hipemu: deadlock -- a barrier / collective was not reached by all of its threads\n");
      std::abort();
    }
  }
}
}  // namespace

void block_barrier() {
  Worker* w = W;
  const unsigned g = w->block_gen;
  if (++w->block_cnt == (unsigned)w->nthreads) {
    w->block_cnt = 0;
    w->block_gen = g + 1;
  } else {
    wait_until_changed(&w->block_gen, g);
  }
}

void wave_barrier() {
  Worker* w = W;
  const int wi = w->cur / 64;
  const unsigned g = w->wave_gen[wi];
  if (++w->wave_cnt[wi] == w->wave_size[wi]) {
    w->wave_cnt[wi] = 0;
    w->wave_gen[wi] = g + 1;
  } else {
    wait_until_changed(&w->wave_gen[wi], g);
  }
}

void launch(dim3 grid, dim3 block, size_t dynamic_lds, const std::function<void()>& body) {
  if (dynamic_lds > 160 * 1024) {
    std::fprintf(stderr, "hipemu: a launch asks for %zu bytes of dynamic LDS (gfx950 has 160 KB per workgroup)\n", dynamic_lds);
    std::abort();
  }
  const int nthreads = (int)(block.x * block.y * block.z);
  const long long nblocks = (long long)grid.x * grid.y * grid.z;
  if (nthreads <= 0 || nblocks <= 0) return;
  int nworkers = 0;
  if (const char* e = std::getenv("HIPEMU_THREADS")) nworkers = std::atoi(e);
  if (nworkers <= 0) nworkers = (int)std::thread::hardware_concurrency();
  if (nworkers > 16) nworkers = 16;
  if (nworkers < 1) nworkers = 1;
  if (nworkers > nblocks) nworkers = (int)nblocks;
  std::atomic<long long> next{0};
  auto work = [&]() {
    Worker w;
    w.nthreads = nthreads;
    w.block = block;
    w.body = &body;

    w.fibers.resize(nthreads);
    w.stacks.resize((size_t)nthreads * kStack);
    const int nwaves = (nthreads + 63) / 64;
    w.waves.resize(nwaves);
    w.wave_gen.assign(nwaves, 0u);
    w.wave_cnt.assign(nwaves, 0u);
    w.wave_size.resize(nwaves);
    for (int i = 0; i < nwaves; ++i) w.wave_size[i] = (unsigned)((i + 1) * 64 <= nthreads ? 64 : nthreads - i * 64);
    int order = 0;
    if (const char* e = std::getenv("HIPEMU_ORDER")) order = std::atoi(e);
    for (int wi = 0; wi < nwaves; ++wi) {
      const int src = order == 0 ? wi : nwaves - 1 - wi;
      for (unsigned l = 0; l < w.wave_size[src]; ++l) w.visit.push_back(src * 64 + (int)(order == 2 ? w.wave_size[src] - 1 - l : l));
    }
    W = &w;
    // exactly the requested bytes from the heap: under HIPEMU_ASAN=1 (build_emu.py) an access past the launch's own
    // figure lands in a redzone instead of in the slack of a fixed buffer
    void* lds = nullptr;
    if (const char* e = std::getenv("HIPEMU_LDS_SHRINK")) {      // positive control of the sanitizer build: hand out LESS
      const size_t cut = (size_t)std::atoi(e);                   // than asked, a kernel that uses its whole figure must trip
      dynamic_lds = dynamic_lds > cut ? dynamic_lds - cut : 0;
    }
    if (dynamic_lds > 0 && posix_memalign(&lds, 64, dynamic_lds) != 0) std::abort();
    g_dynamic_lds = static_cast<float*>(lds);
    blockDim = block;
    gridDim = grid;
    for (long long b = next.fetch_add(1); b < nblocks; b = next.fetch_add(1)) {
      blockIdx.x = (unsigned)(b % grid.x);
      blockIdx.y = (unsigned)((b / grid.x) % grid.y);
      blockIdx.z = (unsigned)(b / ((long long)grid.x * grid.y));
      run_block(&w);
    }
    W = nullptr;
    g_dynamic_lds = nullptr;
    std::free(lds);
  };
  if (nworkers == 1) {
    work();
    return;
  }
  std::vector<std::thread> pool;
  for (int i = 0; i < nworkers; ++i) pool.emplace_back(work);
  for (auto& th : pool) th.join();
}
}  // namespace hipemu

float* hipemu_shared_memory() { return hipemu::g_dynamic_lds; }
