// common.h -- error handling, device buffers, CSR container shared by all csrc/ headers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/csgpu.h"

namespace csgpu {

struct Error : public std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define CS_HIP(expr)                                                                                    \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) {                                                                             \
      char _buf[512];                                                                                   \
      snprintf(_buf, sizeof(_buf), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
               __LINE__);                                                                               \
      throw ::csgpu::Error((int)_e == (int)hipErrorOutOfMemory ? CSGPU_OOM : CSGPU_HIP_ERROR, _buf);    \
    }                                                                                                   \
  } while (0)

#define CS_REQUIRE(cond, code, msg)                  \
  do {                                               \
    if (!(cond)) throw ::csgpu::Error((code), (msg)); \
  } while (0)

// Checked after every kernel launch sequence in setup code paths (cheap: no sync).
inline void check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(CSGPU_HIP_ERROR, std::string(what) + ": " + hipGetErrorString(e));
}

// ---- knobs ----------------------------------------------------------------------------------------------------------
// Every decision that used to be an environment switch inside the library lives here: the RESOLVED
// choices of ONE handle -- library defaults, overridden by the caller's csgpu_opts fields of the same names (0 = default
// there), overridden by the CSGPU_* environment variables as a DEBUG aid, read ONCE when the handle is set up
// (knobs_from_opts, csgpu.hip) and never on a call path. The handle that is being set up or solved on puts its knobs in
// scope for the calling thread (KnobScope at the top of every Solver method); code without a handle in reach -- launch
// helpers, the set-up routines of the levels -- reads knobs(). Two handles of one process may differ.
struct Knobs {
  int last_level_sweeps = 0;      // damped-Jacobi sweeps standing in for the coarsest solve of a level too large for a dense
                                  // inverse (vcycle, pcg.h): 0 = the library's rule (8 below a coarsened hierarchy,
                                  // kSingleLevelSweeps for a hierarchy of one level), -1 = none (plain Jacobi scaling)
  bool enrich = true;             // second coarse function on badly shaped aggregates (enrich.h)
  int enrich_steps = 6;
  double enrich_tau = 0.06;
  int64_t dia25_min_rows = 16384;  // smallest level that takes the 25-point lattice form (dia25.h); < 0: none
  bool dia25_prefetch = true;
  int dia25_waves = 0;            // 0: the rule of dia25_waves()
  bool dia25_fused_j0 = true;     // first two sweeps of a 25-point level as one marching pass
  int stream = 0;                 // streaming pair solves: 0 = decided by the first batch's spread, 1 = from the first pair, -1 = never
  int64_t stream_min = (int64_t)1 << 25;
  double hetero_fp64_frac = 0.03;  // fp32 hierarchy replaced by fp64 above this fraction of cells leaving their tile (>= 1: never)
  int tail_rows = 4096;           // levels with at most this many rows run in the single-launch coarse tail; <= 0: no tail
  int poly_lattice = 0;           // polygon rasters on the lattice path: 0 = shape rule, 1 = any shape, -1 = never
  double poly_strength = 0.0, poly_coef = 1.0, poly_smin = 8.0, poly_smax = 1000.0;
  bool cellspace = true, cellspace_from_csr = true;
  double cellspace_min_frac = 0.5;
  bool lattice_l1 = true;
  int lattice_l1_min_rows = 16384;
  bool stencil = true, two_product = true;  // (csgpu_opts.stencil / .two_product = -1 or the NO_STENCIL / NO_TWO_PRODUCT variables)
  bool direct_lattice = true, lattice_s = true, lattice_q = true, direct_tiles = true, tile_pieces = true, direct_at = true;
  double tile_theta = 0.03, tile_split_min = 0.005;
  bool dirichlet_coarse = true, deflation = true, tail_projection = true;
  bool expander_probe = true;     // large graphs without coordinates: predict the expander bail-out before the MIS(2) aggregation
  int coarse_smoother = 0;        // 0 = Chebyshev weights unless the hierarchy is fp32 above 3e7 rows, 1 = Chebyshev, 2 = damped Jacobi
  int nu_l1 = 0, nu_deep = 0;     // sweeps on level 1 / below; 0 = nu_coarse / nu_coarse + 1
  int64_t host_stream_block = 0;  // entries per block of a streamed host matrix; 0 = stream only matrices with >= 2^31 entries
  bool wide_csr = false;          // fp64 CSR-path handles at K = 32 too
  bool fixed_k = false;           // every batch of a call at the call's width (round-5 behaviour)
  bool recompute_ap = true;       // residual update recomputes A p from the lattice form instead of storing it
  int fused_restrict = 0;         // residual update and the V-cycle's restriction in one marching pass (lattice.h): 1 on, -1 off,
                                  // 0 = on in double precision only. Measured at 10000^2, K = 32:
                                  // fp64 45.2 - 46.8 -> 49.1 - 49.4 pair-solves/s; single precision 117.5 - 119.4 with two
                                  // passes against 111.7 - 113.7 fused (512 / 256 threads): half the bytes per entry, the same
                                  // LDS traffic and barriers
  int fused_seg = 64;             // coarse columns per tile of that pass (restrict_seg, when given, sets both)
  int fused_level1 = 0;           // lattice V(2,2) levels: x = S b and b_c = Q2' b in one pass over b (1 on, -1 off, 0 = fp64 only)
  bool sparse_init = true;        // fused path, pair solves: r0 = e_dst - e_src is never stored (PcgParams::pair_src, pcg.h)
  int64_t collapse_min = -1;      // < 0: the default rule of pcg.h
  bool longrow = true, narrow_tile = false;
  int spmv_grid_cap = 65536, dia_seg = 0, restrict_seg = 32;
  int verbose = 0;
  // experiments / debugging only (no csgpu_opts field)
  double pinv_cut = 0.0;
  bool kernel_gain_ref = false, tail_debug = false;
  bool galerkin_staged = false;   // level-1 Galerkin product of the lattice set-up from LDS-staged fine columns (lattice_setup.h):
                                  // same bits, 36.7 against 19.4 ms at 10000^2 fp64 (profiles/r6_setup_staged_galerkin_ab.jsonl) -- off
  bool raster_transpose = true;   // csgpu_raster_setup: the raster kernels of the index-free pipeline read a column-major copy
  bool enrich_fused = true;       // enriched levels take the fused residual update + restriction (enrich_coarse_fix, enrich.h)
  int apq_nt = 256;               // cells per workgroup of lattice_ap_q_kernel (256 / 128 / 64: 22.2 / 23.8 / 24 ms, same file --
                                  // measured BEFORE its loads were issued up front, which is what that kernel was waiting for)
  int timed_launches = 512;
};

inline const Knobs*& knobs_slot() {
  static thread_local const Knobs* p = nullptr;
  return p;
}
inline const Knobs& knobs() {
  static const Knobs dflt;
  const Knobs* p = knobs_slot();
  return p ? *p : dflt;
}
struct KnobScope {
  const Knobs* prev;
  explicit KnobScope(const Knobs* k) : prev(knobs_slot()) { knobs_slot() = k; }
  ~KnobScope() { knobs_slot() = prev; }
  KnobScope(const KnobScope&) = delete;
  KnobScope& operator=(const KnobScope&) = delete;
};

static std::atomic<int64_t> g_live_bytes{0};  // bookkeeping only (handles of a multi-device set are built by concurrent threads)

// ---- device memory pool ---------------------------------------------------------------------------------------------
// Measured on MI355X (tools/alloc_probe.py): releasing the ~40-100 GB of a 10000^2 handle with hipFree and allocating
// the next handle's buffers costs 3.5-4.5 s of driver time (page-table teardown / set-up), an order of magnitude more
// than the whole AMG setup -- and the reference's workloads do exactly that (one factorisation per component, per
// focal region, per one-to-all source: src/core.jl:146-167, src/raster/onetoall.jl:106-151). Released blocks are
// therefore kept in a per-device pool and handed out again for requests they fit with at most 1/8 of slack (the sizes
// of "the same problem again" repeat exactly; temporaries whose size follows a list length vary a little).
//   * A released block first sits in a PENDING list: kernels of its previous owner may still be running. It becomes
//     reusable at the next point the pool drains the device -- once per group of releases (when an allocation finds
//     nothing ready), not once per block.
//   * The pool is capped (CSGPU_POOL_MAX_GB, default 60 % of the device's memory): beyond the cap the blocks released
//     longest ago go back to the driver, so buffer sizes that never repeat cannot strand memory without bound.
//   * csgpu_trim_memory() / a failed hipMalloc return everything to the driver. CSGPU_NO_POOL=1 disables pooling.
// Handles are NOT trimmed when the last one of a device is freed: "free the factorisation, build the next one" is the
// very pattern the pool exists for.
struct DevicePool {
  static const int kMaxDev = 64;
  struct Blk {
    void* p;
    size_t cap;
    uint64_t seq;  // release order (eviction: oldest first)
  };
  std::mutex mu[kMaxDev];
  std::multimap<size_t, Blk> ready[kMaxDev];  // capacity -> block, device drained since the release
  std::vector<Blk> pending[kMaxDev];          // released, device not drained yet
  size_t pooled[kMaxDev] = {0};               // bytes in ready + pending
  size_t limit[kMaxDev] = {0};                // cap in bytes (0: not initialised yet)
  std::atomic<uint64_t> seq{0};  // one release counter for all devices, bumped under different per-device mutexes
  bool enabled = getenv("CSGPU_NO_POOL") == nullptr;
  // (caller holds mu[dev] and has `dev` current) drain the device and make the pending blocks reusable
  void promote(int dev) {
    if (pending[dev].empty()) return;
    (void)hipDeviceSynchronize();
    for (const Blk& b : pending[dev]) ready[dev].emplace(b.cap, b);
    pending[dev].clear();
  }
  // a block of at least b bytes with at most b/8 of slack; *cap = its capacity. The current device must be `dev`.
  void* take(int dev, size_t b, size_t* cap) {
    std::lock_guard<std::mutex> lk(mu[dev]);
    for (int pass = 0; pass < 2; ++pass) {
      auto it = ready[dev].lower_bound(b);
      if (it != ready[dev].end() && it->first <= b + b / 8) {
        void* p = it->second.p;
        *cap = it->second.cap;
        ready[dev].erase(it);
        pooled[dev] -= *cap;
        return p;
      }
      if (pass == 0) {
        bool fits = false;
        for (const Blk& q : pending[dev]) fits = fits || (q.cap >= b && q.cap <= b + b / 8);
        if (!fits) break;
        promote(dev);
      }
    }
    return nullptr;
  }
  void give(int dev, size_t cap, void* p) {
    std::vector<void*> victims;
    {
      std::lock_guard<std::mutex> lk(mu[dev]);
      if (limit[dev] == 0) {
        size_t fr = 0, tot = 0;
        const char* e = getenv("CSGPU_POOL_MAX_GB");
        if (e) limit[dev] = (size_t)(atof(e) * 1073741824.0) + 1;
        else {
          // the cap is a fraction of THIS device's memory: hipMemGetInfo answers for the current device
          int cur = dev;
          (void)hipGetDevice(&cur);
          if (cur != dev) (void)hipSetDevice(dev);
          const bool ok = hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0;
          if (cur != dev) (void)hipSetDevice(cur);
          limit[dev] = ok ? tot / 10 * 6 : (size_t)64 << 30;
        }
      }
      pending[dev].push_back(Blk{p, cap, ++seq});
      pooled[dev] += cap;
      if (pooled[dev] > limit[dev]) {
        int cur = dev;
        (void)hipGetDevice(&cur);
        if (cur != dev) (void)hipSetDevice(dev);
        promote(dev);
        if (cur != dev) (void)hipSetDevice(cur);
        while (pooled[dev] > limit[dev] && !ready[dev].empty()) {
          auto oldest = ready[dev].begin();
          for (auto it = ready[dev].begin(); it != ready[dev].end(); ++it)
            if (it->second.seq < oldest->second.seq) oldest = it;
          victims.push_back(oldest->second.p);
          pooled[dev] -= oldest->second.cap;
          ready[dev].erase(oldest);
        }
      }
    }
    for (void* v : victims) hipFree(v);
  }
  // return the pooled blocks of one device (dev < 0: of every device) to the driver; bytes released
  size_t trim(int dev) {
    size_t freed = 0;
    for (int d = 0; d < kMaxDev; ++d) {
      if (dev >= 0 && d != dev) continue;
      std::vector<void*> victims;
      {
        std::lock_guard<std::mutex> lk(mu[d]);
        for (auto& kv : ready[d]) victims.push_back(kv.second.p);
        for (const Blk& b : pending[d]) victims.push_back(b.p);
        freed += pooled[d];
        ready[d].clear();
        pending[d].clear();
        pooled[d] = 0;
      }
      for (void* p : victims) hipFree(p);  // (hipFree waits for the device: pending blocks are safe to free)
    }
    return freed;
  }
};
inline DevicePool& device_pool() {
  static DevicePool* p = new DevicePool();  // leaked on purpose: device memory is reclaimed at process exit
  return *p;
}

// Owning device allocation (pooled, see above).
struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;  // bytes asked for
  size_t cap = 0;    // capacity of the underlying block (>= bytes: a pooled block may be slightly larger)
  int dev = -1;      // device the block lives on
  DBuf() {}
  explicit DBuf(size_t b) { alloc(b); }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : p(o.p), bytes(o.bytes), cap(o.cap), dev(o.dev) {
    o.p = nullptr;
    o.bytes = 0;
    o.cap = 0;
  }
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      cap = o.cap;
      dev = o.dev;
      o.p = nullptr;
      o.bytes = 0;
      o.cap = 0;
    }
    return *this;
  }
  ~DBuf() { release(); }
  void alloc(size_t b) {
    release();
    if (b == 0) return;
    CS_HIP(hipGetDevice(&dev));
    DevicePool& pool = device_pool();
    const bool poolable = pool.enabled && dev >= 0 && dev < DevicePool::kMaxDev;
    cap = b;
    if (poolable) p = pool.take(dev, b, &cap);
    if (!p) {
      cap = b;
      hipError_t e = hipMalloc(&p, b);
      if (e != hipSuccess && poolable && pool.trim(dev) > 0) {  // the pool may be what fills the device
        (void)hipGetLastError();
        e = hipMalloc(&p, b);
      }
      if (e != hipSuccess) {
        p = nullptr;
        cap = 0;
        CS_HIP(e);
      }
    }
    bytes = b;
    g_live_bytes += (int64_t)b;
  }
  void release() {
    if (p) {
      DevicePool& pool = device_pool();
      if (pool.enabled && dev >= 0 && dev < DevicePool::kMaxDev)
        pool.give(dev, cap, p);
      else
        hipFree(p);
      g_live_bytes -= (int64_t)bytes;
    }
    p = nullptr;
    bytes = 0;
    cap = 0;
  }
  template <class U>
  U* as() const {
    return (U*)p;
  }
};

template <class U>
inline U* dptr(const DBuf& b) {
  return (U*)b.p;
}

template <class U>
inline DBuf dalloc(size_t count) {
  return DBuf(count * sizeof(U));
}

// Device CSR matrix, int32 / 0-based, columns sorted within a row.
template <class T>
struct Csr {
  int nrows = 0, ncols = 0;
  int64_t nnz = 0;
  DBuf rowptr, col, val;
  const int* rp() const { return rowptr.as<int>(); }
  const int* ci() const { return col.as<int>(); }
  const T* va() const { return val.as<T>(); }
  int* rp() { return rowptr.as<int>(); }
  int* ci() { return col.as<int>(); }
  T* va() { return val.as<T>(); }
  size_t device_bytes() const { return rowptr.bytes + col.bytes + val.bytes; }
};

// Lattice ("symmetric diagonal") form of a symmetric matrix whose node i is coupled to i+-1, i+-(R-1), i+-R, i+-(R+1)
// only (an all-valid raster in column-major numbering, see stencil.h):
//   rows[i] = { M[i,i], M[i,i+1], M[i,i+R-1], M[i,i+R], M[i,i+R+1] }   (0 where the entry is absent)
template <class T>
struct Dia {
  int64_t n = 0;
  int R = 0;   // lattice period (raster height)
  DBuf rows;   // [n][5] of T
  const T* data() const { return rows.as<T>(); }
  size_t device_bytes() const { return rows.bytes; }
};

// Index-free form of the level-0 transfer operator Q (n x n_c) of a raster whose aggregates are the regular 3x3
// tiles: fine cell (i, j) belongs to tile (I, J) = (min(i/3, Rc-1), min(j/3, Cc-1)), coarse node id J*Rc + I, and row
// (i, j) of Q only touches the 3 x 3 block of tiles around (I, J):
//   q[node][(dJ+1)*3 + (dI+1)] = Q[node, (J+dJ)*Rc + (I+dI)]      (0 where the entry is absent / outside the raster)
template <class T>
struct LatticeQ {
  int64_t n = 0;
  int R = 0, C = 0, Rc = 0, Cc = 0;
  DBuf q;  // [n][9] of T
  const T* data() const { return q.as<T>(); }
  size_t device_bytes() const { return q.bytes; }
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid size for grid-stride elementwise / reduction kernels: fixed cap so partial-sum layouts (and hence
// floating-point summation order) do not depend on the device.
static const int kBlock = 256;
static const int kMaxGrid = 2048;
inline int grid_for(int64_t work_items, int per_block = kBlock) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

}  // namespace csgpu
