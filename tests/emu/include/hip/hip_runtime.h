// =============================================================================
// tests/emu/include/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny functional emulator of the subset of the HIP programming model that csrc/*.hip uses, so that the
// *unmodified* kernel sources can be compiled with g++ (-I tests/emu/include shadows the real
// <hip/hip_runtime.h>) and their LOGIC exercised on the CPU-only build container (`-m "not gpu"` tests).
// It is NOT a product path and NOT a CPU fallback: the product loader (circuitscape.jl_amd/lib.py) only ever
// loads libcsgpu.so built by hipcc for gfx950 and fails loudly if that is missing; the emulated library
// (tests/emu/libcsgpu_emu.so) is opened exclusively by tests that name it explicitly.
//
// Model: every thread of a workgroup is a ucontext fiber on ONE OS thread (so `__shared__` can be a
// `static thread_local`), `__syncthreads()` and the 64-lane wave collectives are cooperative yields;
// workgroups of one launch are distributed over a few OS threads, global atomics are real atomics.
// Wavefront width is 64, as on gfx950.
// =============================================================================
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#ifndef __restrict__
#define __restrict__ __restrict
#endif
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)::hipemu::cur()->dyn_smem;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint3_emu {
  unsigned x, y, z;
};

typedef int hipError_t;
typedef void* hipStream_t;
struct hipEmuEvent {
  std::chrono::steady_clock::time_point t;
};
typedef hipEmuEvent* hipEvent_t;
// stream capture / graphs: a captured graph is the recorded list of launches, replayed in order
struct hipEmuGraph { std::vector<std::function<void()>> nodes; };
typedef hipEmuGraph* hipGraph_t;
typedef hipEmuGraph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
namespace hipemu {
inline hipEmuGraph*& capturing() { static thread_local hipEmuGraph* g = nullptr; return g; }
}
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventDefault = 0 };
struct hipDeviceProp_t {
  char name[256];
  size_t totalGlobalMem;
  int multiProcessorCount;
  int warpSize;
  char gcnArchName[256];
};

namespace hipemu {

struct Fiber {
  ucontext_t ctx;
  uint3_emu tid;
  int flat;
  int lane, wave;
  bool done;
  char* stack;
};

struct Wave {
  uint64_t slot[64];
  int arrived;
  int live;
  unsigned gen;
  int garrived[64];   // sub-wave group barriers, indexed by the group's base lane
  unsigned ggen[64];
};

struct BlockCtx {
  uint3_emu bid, bdim, gdim;
  int nthreads;
  int live;
  int arrived;
  unsigned gen;
  Fiber* fibers;
  Wave* waves;
  Fiber* current;
  ucontext_t sched;
  char* dyn_smem;
  const std::function<void()>* body;
};

inline BlockCtx*& cur() {
  static thread_local BlockCtx* c = nullptr;
  return c;
}

inline void yield_to_scheduler() {
  BlockCtx* b = cur();
  swapcontext(&b->current->ctx, &b->sched);
}

inline void fiber_entry() {
  BlockCtx* b = cur();
  Fiber* f = b->current;
  (*b->body)();
  f->done = true;
  b->live--;
  b->waves[f->wave].live--;
  swapcontext(&f->ctx, &b->sched);
}

struct ThreadState {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  std::vector<char> dyn;
  BlockCtx ctx;
};

inline ThreadState& tstate() {
  static thread_local ThreadState ts;
  return ts;
}

static const size_t kStack = 128 * 1024;

inline void run_block(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem, unsigned bx, unsigned by,
                      unsigned bz) {
  ThreadState& ts = tstate();
  const int nt = (int)(block.x * block.y * block.z);
  if ((int)ts.fibers.size() < nt) {
    size_t old = ts.fibers.size();
    ts.fibers.resize(nt);
    for (size_t i = old; i < (size_t)nt; ++i) ts.fibers[i].stack = (char*)malloc(kStack);
  }
  const int nw = (nt + 63) / 64;
  if ((int)ts.waves.size() < nw) ts.waves.resize(nw);
  if (ts.dyn.size() < shmem + 64) ts.dyn.resize(shmem + 64);
  BlockCtx& b = ts.ctx;
  b.bid = {bx, by, bz};
  b.bdim = {block.x, block.y, block.z};
  b.gdim = {grid.x, grid.y, grid.z};
  b.nthreads = nt;
  b.live = nt;
  b.arrived = 0;
  b.gen = 0;
  b.fibers = ts.fibers.data();
  b.waves = ts.waves.data();
  b.dyn_smem = (char*)(((uintptr_t)ts.dyn.data() + 63) & ~(uintptr_t)63);
  b.body = &body;
  cur() = &b;
  for (int w = 0; w < nw; ++w) {
    b.waves[w].arrived = 0;
    b.waves[w].gen = 0;
    b.waves[w].live = std::min(64, nt - 64 * w);
    for (int l = 0; l < 64; ++l) {
      b.waves[w].garrived[l] = 0;
      b.waves[w].ggen[l] = 0;
    }
  }
  for (int t = 0; t < nt; ++t) {
    Fiber& f = b.fibers[t];
    f.flat = t;
    f.tid.x = t % block.x;
    f.tid.y = (t / block.x) % block.y;
    f.tid.z = t / (block.x * block.y);
    f.lane = t & 63;
    f.wave = t >> 6;
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  long spins = 0;
  while (b.live > 0) {
    for (int t = 0; t < nt; ++t) {
      Fiber& f = b.fibers[t];
      if (f.done) continue;
      b.current = &f;
      swapcontext(&b.sched, &f.ctx);
    }
    if (++spins > 50000000L) {
      fprintf(stderr, "hipemu: workgroup (%u,%u,%u) appears deadlocked (divergent barrier?)\n", bx, by, bz);
      abort();
    }
  }
  cur() = nullptr;
}

inline int& num_workers() {
  static int n = [] {
    const char* e = getenv("HIPEMU_THREADS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(v, 16));
  }();
  return n;
}

// Persistent worker pool: fiber stacks are thread_local, so workers must outlive launches.
struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<void(size_t, size_t)> job;
  std::atomic<size_t> next{0};
  size_t total = 0, chunk = 1;
  unsigned long epoch = 0;
  int active = 0;
  bool stop = false;
  void worker() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return stop || epoch != seen; });
        if (stop) return;
        seen = epoch;
      }
      run_chunks();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--active == 0) cv_done.notify_all();
      }
    }
  }
  void run_chunks() {
    for (;;) {
      size_t s = next.fetch_add(chunk);
      if (s >= total) break;
      job(s, std::min(total, s + chunk));
    }
  }
  void run(size_t n, const std::function<void(size_t, size_t)>& f) {
    const int nw = num_workers();
    if ((int)th.size() < nw - 1)
      for (int t = (int)th.size(); t < nw - 1; ++t) th.emplace_back([this] { worker(); });
    {
      std::lock_guard<std::mutex> lk(m);
      job = f;
      total = n;
      chunk = std::max<size_t>(1, n / (size_t)(nw * 8));
      next = 0;
      active = (int)th.size();
      ++epoch;
    }
    cv_job.notify_all();
    run_chunks();
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return active == 0; });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_job.notify_all();
    for (auto& t : th) t.join();
  }
};
inline Pool& pool() {
  static Pool* p = new Pool();  // leaked on purpose: workers may still be parked at process exit
  return *p;
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const size_t total = (size_t)grid.x * grid.y * grid.z;
  if (total == 0) return;
  auto work = [&](size_t b0, size_t b1) {
    for (size_t i = b0; i < b1; ++i) {
      unsigned bx = (unsigned)(i % grid.x), by = (unsigned)((i / grid.x) % grid.y), bz = (unsigned)(i / ((size_t)grid.x * grid.y));
      run_block(body, grid, block, shmem, bx, by, bz);
    }
  };
  if (num_workers() <= 1 || total < 4) {
    work(0, total);
    return;
  }
  // one kernel at a time through the shared worker pool (host threads driving different emulated devices serialise
  // here, like launches on one stream)
  static std::mutex launch_mu;
  std::lock_guard<std::mutex> lk(launch_mu);
  pool().run(total, work);
}

inline void block_barrier() {
  BlockCtx* b = cur();
  unsigned g = b->gen;
  if (++b->arrived >= b->live) {
    b->arrived = 0;
    b->gen++;
    return;
  }
  while (b->gen == g) {
    yield_to_scheduler();
    if (b->gen == g && b->arrived >= b->live) {  // others exited meanwhile
      b->arrived = 0;
      b->gen++;
    }
  }
}

inline void wave_barrier() {
  BlockCtx* b = cur();
  Wave& w = b->waves[b->current->wave];
  unsigned g = w.gen;
  if (++w.arrived >= w.live) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == g) {
    yield_to_scheduler();
    if (w.gen == g && w.arrived >= w.live) {
      w.arrived = 0;
      w.gen++;
    }
  }
}

// Barrier among the live lanes of the width-`width` lane group containing the calling lane.
inline void group_barrier(int width) {
  if (width >= 64) {
    wave_barrier();
    return;
  }
  BlockCtx* b = cur();
  Fiber* f = b->current;
  Wave& w = b->waves[f->wave];
  const int base = f->lane & ~(width - 1);
  auto live_in_group = [&]() {
    int n = 0;
    for (int l = base; l < base + width; ++l) {
      const int t = 64 * f->wave + l;
      if (t < b->nthreads && !b->fibers[t].done) ++n;
    }
    return n;
  };
  unsigned g = w.ggen[base];
  if (++w.garrived[base] >= live_in_group()) {
    w.garrived[base] = 0;
    w.ggen[base]++;
    return;
  }
  while (w.ggen[base] == g) {
    yield_to_scheduler();
    if (w.ggen[base] == g && w.garrived[base] >= live_in_group()) {
      w.garrived[base] = 0;
      w.ggen[base]++;
    }
  }
}

template <class T>
inline T wave_exchange(T v, int src_lane, int width = 64) {
  static_assert(sizeof(T) <= 8, "shuffle payload too large");
  BlockCtx* b = cur();
  Fiber* f = b->current;
  Wave& w = b->waves[f->wave];
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.slot[f->lane] = raw;
  group_barrier(width);
  const int wave_lanes = std::min(64, b->nthreads - 64 * f->wave);
  T out = v;
  if (src_lane >= 0 && src_lane < wave_lanes) {
    uint64_t r = w.slot[src_lane];
    memcpy(&out, &r, sizeof(T));
  }
  group_barrier(width);
  return out;
}

}  // namespace hipemu

// ---- built-in variables
#define threadIdx (::hipemu::cur()->current->tid)
#define blockIdx (::hipemu::cur()->bid)
#define blockDim (::hipemu::cur()->bdim)
#define gridDim (::hipemu::cur()->gdim)
static const int warpSize = 64;

inline void __syncthreads() { ::hipemu::block_barrier(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() {}

template <class T>
inline T __shfl(T v, int src, int width = 64) {
  int lane = ::hipemu::cur()->current->lane;
  int base = lane & ~(width - 1);
  return ::hipemu::wave_exchange(v, base + (src & (width - 1)), width);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = ::hipemu::cur()->current->lane;
  int base = lane & ~(width - 1);
  int src = lane + (int)d;
  return ::hipemu::wave_exchange(v, src < base + width ? src : lane, width);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = ::hipemu::cur()->current->lane;
  int base = lane & ~(width - 1);
  int src = lane - (int)d;
  return ::hipemu::wave_exchange(v, src >= base ? src : lane, width);
}
template <class T>
inline T __shfl_xor(T v, int m, int width = 64) {
  int lane = ::hipemu::cur()->current->lane;
  int base = lane & ~(width - 1);
  int src = lane ^ m;
  return ::hipemu::wave_exchange(v, (src >= base && src < base + width) ? src : lane, width);
}
inline unsigned long long __ballot(int pred) {
  int lane = ::hipemu::cur()->current->lane;
  unsigned long long m = 0;
  // deposit, then every lane reads all slots
  auto* b = ::hipemu::cur();
  auto& w = b->waves[b->current->wave];
  w.slot[lane] = pred ? 1 : 0;
  ::hipemu::wave_barrier();
  const int wave_lanes = std::min(64, b->nthreads - 64 * b->current->wave);
  for (int l = 0; l < wave_lanes; ++l)
    if (!b->fibers[64 * b->current->wave + l].done && w.slot[l]) m |= (1ull << l);
  ::hipemu::wave_barrier();
  return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline double __longlong_as_double(long long x) { double d; memcpy(&d, &x, 8); return d; }
inline long long __double_as_longlong(double x) { long long d; memcpy(&d, &x, 8); return d; }

// ---- atomics (global or shared)
template <class T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  float old = *p, nv;
  do { nv = old + v; } while (!__atomic_compare_exchange(p, &old, &nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
inline double atomicAdd(double* p, double v) {
  double old = *p, nv;
  do { nv = old + v; } while (!__atomic_compare_exchange(p, &old, &nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
template <class T>
inline T atomicCAS(T* p, T cmp, T val) {
  __atomic_compare_exchange(p, &cmp, &val, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}
template <class T>
inline T atomicMax(T* p, T v) {
  T old = *p;
  while (old < v && !__atomic_compare_exchange(p, &old, &v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T>
inline T atomicMin(T* p, T v) {
  T old = *p;
  while (old > v && !__atomic_compare_exchange(p, &old, &v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <class T>
inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }

// clang builtins used by the kernels
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// ---- kernel launch
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                         \
  do {                                                                                                     \
    auto _hipemu_args = std::make_tuple(__VA_ARGS__);                                                      \
    std::function<void()> _hipemu_fn([=]() { std::apply(kernel, _hipemu_args); });                        \
    const dim3 _hipemu_g = dim3(grid), _hipemu_b = dim3(block);                                            \
    const size_t _hipemu_s = (size_t)(shmem);                                                              \
    if (::hipemu::capturing())                                                                             \
      ::hipemu::capturing()->nodes.push_back(                                                              \
          [=]() { ::hipemu::launch(_hipemu_g, _hipemu_b, _hipemu_s, _hipemu_fn); });                       \
    else                                                                                                   \
      ::hipemu::launch(_hipemu_g, _hipemu_b, _hipemu_s, _hipemu_fn);                                       \
  } while (0)

// HIP exposes min/max in the global namespace for device code
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
inline double min(double a, double b) { return a < b ? a : b; }
inline double max(double a, double b) { return a > b ? a : b; }

// ---- runtime API subset
// HIPEMU_DEVICES emulated devices share the host's memory (multi-device code paths run, nothing is isolated)
inline int hipemu_device_count() {
  const char* e = getenv("HIPEMU_DEVICES");
  const int v = e ? atoi(e) : 1;
  return v < 1 ? 1 : v;
}
inline int& hipemu_current_device() {
  static thread_local int d = 0;
  return d;
}
inline hipError_t hipGetDeviceCount(int* n) { *n = hipemu_device_count(); return hipSuccess; }
inline hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= hipemu_device_count()) return hipErrorInvalidDevice;
  hipemu_current_device() = d;
  return hipSuccess;
}
inline hipError_t hipGetDevice(int* d) { *d = hipemu_current_device(); return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipemu (CPU fiber emulator, tests only)");
  strcpy(p->gcnArchName, "emu");
  p->totalGlobalMem = (size_t)8 << 30;
  p->multiProcessorCount = 8;
  p->warpSize = 64;
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  if (n == 0) return hipSuccess;
  if (posix_memalign(p, 256, n) != 0) return hipErrorOutOfMemory;
  memset(*p, 0xCD, std::min<size_t>(n, 4096));  // poison the head: catch reads of uninitialised buffers
  return hipSuccess;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T>
inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) {
  if (hipemu::capturing()) { hipemu::capturing()->nodes.push_back([=]() { hipMemcpy(d, s, n, k); }); return hipSuccess; }
  return hipMemcpy(d, s, n, k);
}
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) {
  if (hipemu::capturing()) { hipemu::capturing()->nodes.push_back([=]() { hipMemset(d, v, n); }); return hipSuccess; }
  return hipMemset(d, v, n);
}
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
  if (hipemu::capturing()) return hipErrorInvalidValue;
  hipemu::capturing() = new hipEmuGraph();
  return hipSuccess;
}
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  *g = hipemu::capturing();
  hipemu::capturing() = nullptr;
  return *g ? hipSuccess : hipErrorInvalidValue;
}
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new hipEmuGraph(*g); return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto& f : e->nodes) f(); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)0x1; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEmuEvent(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
