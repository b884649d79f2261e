// hipsim runtime: fibres, barriers, cross-lane exchange and the workgroup scheduler (TEST INFRASTRUCTURE ONLY, see hip/hip_runtime.h).
//
// One workgroup runs on one host thread; its threads are fibres that switch at barriers and cross-lane operations (round robin, a
// fibre runs until it has to wait).  Workgroups of a launch are handed to a small pool of host threads.  A launch returns when all
// its workgroups have finished, so every stream is synchronous.  If all fibres of a workgroup wait and none can be released (a
// barrier or a cross-lane operation in divergent control flow) the run aborts with the kernel's name.
//
// Lock step.  On the device the lanes of a wavefront execute one instruction stream together: when lane 0 stores to a location the
// other lanes loaded from EARLIER in the program ("a = sN[p]; ...; if (lane == 0) sN[p] = a - t;"), they have all loaded.  A fibre
// that ran ahead would store first.  The kernel sources are therefore compiled with store callbacks (the address-sanitizer
// instrumentation of the compiler, stores only, calls only; the callbacks are defined here, no sanitizer runtime is linked): a
// store to anything but the fibre's own stack is HELD BACK until no other fibre of the workgroup can run, i.e. until every other
// lane has reached its next barrier, cross-lane operation, fence or held-back store.  Loads that precede such a point in the program
// have then been done.  (The opposite pattern - store, then a load by another lane with no meeting point in between - needs a
// fence in the source, which is a meeting point here.)
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" void hipsim_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipsim_switch,.-hipsim_switch
)");

namespace hipsim {

thread_local Fiber* cur = nullptr;
thread_local Block* blk = nullptr;

namespace {

constexpr int MAX_THREADS = 1024;
constexpr size_t STACK = 256 * 1024;
constexpr size_t LDS_BYTES = 160 * 1024;

struct Worker {
  Fiber fibers[MAX_THREADS];
  Wave waves[MAX_THREADS / 64];
  Block block;
  int nf = 0;
  void* main_sp = nullptr;
  char* stacks = nullptr;
  char* lds = nullptr;
  void (*tramp)(void*) = nullptr;
  void* arg = nullptr;
  const char* name = "";
};
thread_local Worker* W = nullptr;

Worker* worker() {
  if (!W) {
    W = new Worker;
    W->stacks = static_cast<char*>(mmap(nullptr, STACK * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (W->stacks == MAP_FAILED) { perror("[hipsim] mmap"); abort(); }
    W->lds = static_cast<char*>(aligned_alloc(128, LDS_BYTES));
  }
  return W;
}

inline bool runnable(const Worker* w, const Fiber& f) {
  if (f.done) return false;
  if (f.waiting == 0) return true;
  if (f.waiting == 1) return w->block.gen != f.wait_gen;
  if (f.waiting == 3) return w->block.store_gen != f.wait_gen;
  return f.wave->gen != f.wait_gen;
}

// nobody can run: let the held-back stores go (returns false if there are none)
inline bool release_stores(Worker* w) {
  for (int i = 0; i < w->nf; ++i)
    if (!w->fibers[i].done && w->fibers[i].waiting == 3 && w->fibers[i].wait_gen == w->block.store_gen) {
      w->block.store_gen++;
      return true;
    }
  return false;
}

[[noreturn]] void deadlock(Worker* w) {
  int nb = 0, nw = 0;
  for (int i = 0; i < w->nf; ++i) {
    if (w->fibers[i].done) continue;
    (w->fibers[i].waiting == 1 ? nb : nw)++;
  }
  fprintf(stderr,
          "[hipsim] deadlock in %s, workgroup (%u,%u,%u): %d threads wait at a workgroup barrier, %d at a wavefront operation "
          "(barrier or cross-lane operation in divergent control flow?)\n",
          w->name, w->block.bid.x, w->block.bid.y, w->block.bid.z, nb, nw);
  abort();
}

// switch to the next fibre that can run; returns when the calling fibre can run again
void yield_from(Worker* w, Fiber* f) {
  const int n = w->nf, me = f->lin;
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 1; k <= n; ++k) {
      int c = me + k;
      if (c >= n) c -= n;
      Fiber& g = w->fibers[c];
      if (!runnable(w, g)) continue;
      if (c == me) return;
      cur = &g;
      hipsim_switch(&f->sp, g.sp);
      return;  // somebody switched back to us: our wait is over (they checked runnable())
    }
    if (!release_stores(w)) break;
  }
  deadlock(w);
}

void release_if_complete(Block& b) {
  if (b.live > 0 && b.arrived == b.live) { b.gen++; b.arrived = 0; }
}
void release_if_complete(Wave& v) {
  if (v.live > 0 && v.arrived == v.live) { v.gen++; v.arrived = 0; }
}

void fiber_main() {
  Worker* w = W;
  w->tramp(w->arg);
  // the kernel returned for this thread: leave the barriers' head counts
  Fiber* f = cur;
  f->done = true;
  w->block.live--;
  f->wave->live--;
  f->wave->live_mask &= ~(1ull << f->lane);
  release_if_complete(w->block);
  release_if_complete(*f->wave);
  const int n = w->nf;
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 1; k < n; ++k) {
      int c = f->lin + k;
      if (c >= n) c -= n;
      Fiber& g = w->fibers[c];
      if (!runnable(w, g)) continue;
      cur = &g;
      hipsim_switch(&f->sp, g.sp);
      abort();  // a finished fibre is never resumed
    }
    if (!release_stores(w)) break;
  }
  if (w->block.live != 0) deadlock(w);
  void* dummy;
  hipsim_switch(&dummy, w->main_sp);
  abort();
}

void run_block(Worker* w, dim3 grid, dim3 block, dim3 bid, size_t shmem) {
  const int n = (int)(block.x * block.y * block.z);
  w->nf = n;
  w->block.bid = bid;
  w->block.bdim = block;
  w->block.gdim = grid;
  w->block.gen = 0;
  w->block.store_gen = 0;
  w->block.arrived = 0;
  w->block.live = n;
  w->block.dyn = w->lds;
  if (shmem) std::memset(w->lds, 0xFF, shmem);  // NaNs: LDS is not initialised on the device either
  const int nw = (n + 63) / 64;
  for (int v = 0; v < nw; ++v) {
    Wave& wv = w->waves[v];
    wv.gen = 0;
    wv.arrived = 0;
    wv.live = std::min(64, n - 64 * v);
    wv.live_mask = wv.live == 64 ? ~0ull : ((1ull << wv.live) - 1);
    std::memset(wv.xbuf, 0, sizeof(wv.xbuf));
  }
  for (int i = 0; i < n; ++i) {
    Fiber& f = w->fibers[i];
    f.lin = i;
    f.lane = i & 63;
    f.wave = &w->waves[i >> 6];
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    f.waiting = 0;
    f.xop = 0;
    f.done = false;
    f.stack_lo = reinterpret_cast<uintptr_t>(w->stacks + STACK * (size_t)i);
    // initial frame: six callee-saved registers, then the entry address where `ret` finds it (16-byte aligned slot)
    void** top = reinterpret_cast<void**>(w->stacks + STACK * (size_t)(i + 1));
    void** sp = top - 2;
    *sp = reinterpret_cast<void*>(&fiber_main);
    sp -= 6;
    for (int r = 0; r < 6; ++r) sp[r] = nullptr;
    f.sp = sp;
  }
  blk = &w->block;
  cur = &w->fibers[0];
  hipsim_switch(&w->main_sp, w->fibers[0].sp);
  cur = nullptr;
  blk = nullptr;
}

// ---- pool --------------------------------------------------------------------------------------------------------------------
struct Job {
  dim3 grid, block;
  size_t shmem = 0;
  void (*tramp)(void*) = nullptr;
  void* arg = nullptr;
  const char* name = "";
  std::atomic<long> next{0};
  long total = 0;
};

struct Pool {
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  Job* job = nullptr;
  unsigned long epoch = 0;
  int busy = 0;
  bool stop = false;

  static void work(Job* j) {
    Worker* w = worker();
    w->tramp = j->tramp;
    w->arg = j->arg;
    w->name = j->name;
    // HIPSIM_BLOCK_ORDER=reverse: workgroups are handed out last to first.  HIP promises no dispatch order, so results must not
    // change; with HIPSIM_THREADS=1 this is a deterministic second schedule that exposes workgroups reading what a "later" one writes.
    static const bool reverse = getenv("HIPSIM_BLOCK_ORDER") && std::string(getenv("HIPSIM_BLOCK_ORDER")) == "reverse";
    for (;;) {
      long b = j->next.fetch_add(1);
      if (b >= j->total) break;
      if (reverse) b = j->total - 1 - b;
      const dim3 bid((uint32_t)(b % j->grid.x), (uint32_t)((b / j->grid.x) % j->grid.y), (uint32_t)(b / ((long)j->grid.x * j->grid.y)));
      run_block(w, j->grid, j->block, bid, j->shmem);
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      Job* j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return stop || epoch != seen; });
        if (stop) return;
        seen = epoch;
        j = job;
        if (!j) continue;  // woke after the launch had finished
        busy++;
      }
      work(j);
      {
        std::lock_guard<std::mutex> lk(m);
        busy--;
      }
      cv_done.notify_all();
    }
  }
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) threads.emplace_back([this] { loop(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto& t : threads) t.join();
  }
  void run(Job& j) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = &j;
      epoch++;
    }
    cv_work.notify_all();
    work(&j);
    std::unique_lock<std::mutex> lk(m);
    // helpers that have not woken yet will find no workgroup left; wait for those that have
    cv_done.wait(lk, [&] { return busy == 0; });
    job = nullptr;
  }
};

Pool* pool() {
  static Pool* p = [] {
    const char* e = getenv("HIPSIM_THREADS");
    int n = e ? atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    return n > 1 ? new Pool(n - 1) : nullptr;
  }();
  return p;
}
std::mutex launch_mutex;  // one launch at a time (several engines on several host threads share the pool)

}  // namespace

void sync_block() {
  Worker* w = W;
  Fiber* f = cur;
  Block& b = w->block;
  const unsigned g = b.gen;
  if (++b.arrived == b.live) { b.gen++; b.arrived = 0; return; }
  f->waiting = 1;
  f->wait_gen = g;
  yield_from(w, f);
  f->waiting = 0;
}

void sync_wave() {
  Worker* w = W;
  Fiber* f = cur;
  Wave& v = *f->wave;
  const unsigned g = v.gen;
  if (++v.arrived == v.live) { v.gen++; v.arrived = 0; return; }
  f->waiting = 2;
  f->wait_gen = g;
  yield_from(w, f);
  f->waiting = 0;
}

const unsigned long long (*exchange(unsigned long long a, unsigned long long b))[2] {
  Fiber* f = cur;
  Wave& v = *f->wave;
  const int p = (int)(f->xop++ & 1u);
  v.xbuf[p][f->lane][0] = a;
  v.xbuf[p][f->lane][1] = b;
  if (v.live < 64) {
    // lanes that never existed or have left the kernel read as zero
    for (int l = 0; l < 64; ++l)
      if (!((v.live_mask >> l) & 1ull)) v.xbuf[p][l][0] = v.xbuf[p][l][1] = 0;
  }
  sync_wave();
  return v.xbuf[p];
}

// a store of device code to memory other lanes can see (see "Lock step" above)
void store_hook(uintptr_t addr) {
  Fiber* f = cur;
  if (!f) return;                          // host code
  if (addr - f->stack_lo < STACK) return;  // the fibre's own stack
  Worker* w = W;
  f->waiting = 3;
  f->wait_gen = w->block.store_gen;
  yield_from(w, f);
  f->waiting = 0;
}

struct hipsim_bound { const char* name; int threads; };
extern "C" const hipsim_bound hipsim_bounds[];  // generated by build.py from the kernels' __launch_bounds__

namespace {
// "(jacobi_diag_kernel<8>)" -> "jacobi_diag_kernel"
int declared_thread_limit(const char* launch_name) {
  std::string n(launch_name);
  size_t b = 0;
  while (b < n.size() && (n[b] == '(' || n[b] == ' ')) ++b;
  size_t e = b;
  while (e < n.size() && (isalnum((unsigned char)n[e]) || n[e] == '_')) ++e;
  const std::string base = n.substr(b, e - b);
  for (const hipsim_bound* p = hipsim_bounds; p->name; ++p)
    if (base == p->name) return p->threads;
  return 0;
}
std::mutex attr_mutex;
std::vector<std::pair<const void*, int>> lds_attr;  // kernels that were granted more than the default dynamic LDS
}  // namespace

void set_max_dynamic_lds(const void* kernel, int bytes) {
  std::lock_guard<std::mutex> lk(attr_mutex);
  for (auto& e : lds_attr)
    if (e.first == kernel) { e.second = bytes; return; }
  lds_attr.emplace_back(kernel, bytes);
}

void launch_impl(const char* name, dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* arg, const void* kernel) {
  if (shmem > 64 * 1024) {  // the device refuses such a launch unless hipFuncSetAttribute raised the kernel's limit first
    int granted = 0;
    {
      std::lock_guard<std::mutex> lk(attr_mutex);
      for (auto& e : lds_attr)
        if (e.first == kernel) granted = e.second;
    }
    if ((size_t)granted < shmem) {
      fprintf(stderr, "[hipsim] launch of %s with %zu bytes of dynamic LDS, but hipFuncAttributeMaxDynamicSharedMemorySize was %s (%d)\n", name, shmem,
              granted ? "set lower" : "never set", granted);
      abort();
    }
  }
  const long total = (long)grid.x * grid.y * grid.z;
  const int nt = (int)(block.x * block.y * block.z);
  if (total <= 0 || nt <= 0) return;
  if (const int limit = declared_thread_limit(name)) {
    if ((int)(block.x * block.y * block.z) > limit) {
      fprintf(stderr, "[hipsim] launch of %s with %u threads, its __launch_bounds__ allow %d\n", name, block.x * block.y * block.z, limit);
      abort();
    }
  }
  if (grid.y > 65535u || grid.z > 65535u || block.z > 64u) {
    fprintf(stderr, "[hipsim] launch of %s: grid (%u,%u,%u) block (%u,%u,%u) exceeds the device limits (grid y / z <= 65535)\n", name, grid.x, grid.y, grid.z,
            block.x, block.y, block.z);
    abort();
  }
  if (nt > MAX_THREADS || shmem > LDS_BYTES) {
    fprintf(stderr, "[hipsim] launch of %s: %d threads, %zu bytes of LDS exceed the device limits\n", name, nt, shmem);
    abort();
  }
  static const bool trace = getenv("HIPSIM_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[hipsim] %s grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
  if (cur) { fprintf(stderr, "[hipsim] launch from device code\n"); abort(); }
  Job j;
  j.grid = grid;
  j.block = block;
  j.shmem = shmem;
  j.tramp = tramp;
  j.arg = arg;
  j.name = name;
  j.total = total;
  Pool* p = pool();
  if (!p || total == 1) {
    Pool::work(&j);
    return;
  }
  std::lock_guard<std::mutex> lk(launch_mutex);
  p->run(j);
}

}  // namespace hipsim

// ---- store callbacks emitted by the compiler for the kernel sources (-fsanitize=address, stores only, as calls) ----------------------
extern "C" {
void __asan_init() {}
void __asan_version_mismatch_check_v8() {}
void __asan_store1(uintptr_t a) { hipsim::store_hook(a); }
void __asan_store2(uintptr_t a) { hipsim::store_hook(a); }
void __asan_store4(uintptr_t a) { hipsim::store_hook(a); }
void __asan_store8(uintptr_t a) { hipsim::store_hook(a); }
void __asan_store16(uintptr_t a) { hipsim::store_hook(a); }
void __asan_storeN(uintptr_t a, size_t) { hipsim::store_hook(a); }
void __asan_load1(uintptr_t) {}
void __asan_load2(uintptr_t) {}
void __asan_load4(uintptr_t) {}
void __asan_load8(uintptr_t) {}
void __asan_load16(uintptr_t) {}
void __asan_loadN(uintptr_t, size_t) {}
void __asan_handle_no_return() {}
void* __asan_memcpy(void* d, const void* s, size_t n) { return std::memcpy(d, s, n); }
void* __asan_memmove(void* d, const void* s, size_t n) { return std::memmove(d, s, n); }
void* __asan_memset(void* d, int c, size_t n) { return std::memset(d, c, n); }
}
