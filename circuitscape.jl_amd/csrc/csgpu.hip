// csgpu.hip -- C ABI of libcsgpu.so (see include/csgpu.h for the reference interfaces each entry point replaces).
// Single translation unit: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC csgpu.hip -o libcsgpu.so
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

#include "currents.h"
#include "pairs.h"
#include "pcg.h"
#include "raster.h"
#include "lattice_setup.h"

namespace csgpu {

static thread_local std::string g_last_error;

// One call of the Dirichlet-masked solves (csgpu_solve_grounded / csgpu_solve_sources): right-hand sides dense (host column-major
// n x nrhs) or as sparse columns, one ground set per column, what to hand back.
struct GroundedJob {
  const void* rhs = nullptr;                                 // dense right-hand sides, or ...
  const int64_t *sptr = nullptr, *sidx = nullptr;            // ... sparse columns: entries [sptr[c], sptr[c+1]) of (sidx, sval)
  const void* sval = nullptr;                                //     (NULL: every entry is 1)
  const int64_t *gptr = nullptr, *gidx = nullptr;            // ground sets
  const int64_t* check = nullptr;                            // per column: node whose voltage goes to check_out[c] (< 0: none)
  void *check_out = nullptr, *x_out = nullptr, *curr_out = nullptr, *cum_inout = nullptr, *max_inout = nullptr;
};

// The knobs of one handle: library defaults <- the caller's csgpu_opts (0 = default in every field) <- the CSGPU_* environment
// variables as a debug aid. This is the ONLY place the library reads its environment for them, once per set-up.
inline Knobs knobs_from_opts(const csgpu_opts& o) {
  Knobs k;
  auto flag = [](int32_t v, bool dflt) { return v == 0 ? dflt : v > 0; };
  k.last_level_sweeps = o.last_level_sweeps;
  k.enrich = flag(o.enrich, true);
  if (o.enrich_steps > 0) k.enrich_steps = o.enrich_steps;
  if (o.enrich_tau > 0.0) k.enrich_tau = o.enrich_tau;
  if (o.dia25_min_rows != 0) k.dia25_min_rows = o.dia25_min_rows;
  k.dia25_prefetch = flag(o.dia25_prefetch, true);
  k.dia25_waves = std::max(o.dia25_waves, 0);
  k.dia25_fused_j0 = flag(o.dia25_fused_j0, true);
  k.stream = o.stream;
  if (o.stream_min > 0) k.stream_min = o.stream_min;
  if (o.hetero_fp64_frac > 0.0) k.hetero_fp64_frac = o.hetero_fp64_frac;
  if (o.tail_rows != 0) k.tail_rows = o.tail_rows;
  k.poly_lattice = o.poly_lattice;
  if (o.poly_strength > 0.0) k.poly_strength = o.poly_strength;
  if (o.poly_coef > 0.0) k.poly_coef = o.poly_coef;
  if (o.poly_smin > 0.0) k.poly_smin = o.poly_smin;
  if (o.poly_smax > 0.0) k.poly_smax = o.poly_smax;
  k.cellspace = flag(o.cellspace, true);
  k.cellspace_from_csr = flag(o.cellspace_from_csr, true);
  if (o.cellspace_min_frac > 0.0) k.cellspace_min_frac = o.cellspace_min_frac;
  k.lattice_l1 = flag(o.lattice_level1, true);
  if (o.lattice_level1_min_rows > 0) k.lattice_l1_min_rows = o.lattice_level1_min_rows;
  k.stencil = o.stencil >= 0;
  k.two_product = o.two_product >= 0;
  k.direct_lattice = flag(o.lattice_setup, true);
  k.lattice_s = flag(o.lattice_s, true);
  k.lattice_q = flag(o.lattice_q, true);
  k.direct_tiles = flag(o.direct_tiles, true);
  k.tile_pieces = flag(o.tile_pieces, true);
  k.direct_at = flag(o.direct_at, true);
  if (o.tile_theta != 0.0) k.tile_theta = std::max(o.tile_theta, 0.0);
  if (o.tile_split_min > 0.0) k.tile_split_min = o.tile_split_min;
  k.dirichlet_coarse = flag(o.dirichlet_coarse, true);
  k.deflation = flag(o.deflation, true);
  k.tail_projection = flag(o.tail_projection, true);
  k.coarse_smoother = o.coarse_smoother;
  k.expander_probe = flag(o.expander_probe, true);
  k.nu_l1 = std::max(o.nu_l1, 0);
  k.nu_deep = std::max(o.nu_deep, 0);
  k.host_stream_block = std::max<int64_t>(o.host_stream_block, 0);
  k.wide_csr = o.wide_csr > 0;
  k.fixed_k = o.fixed_k > 0;
  k.recompute_ap = flag(o.recompute_ap, true);
  k.fused_restrict = o.fused_restrict > 0 ? 1 : (o.fused_restrict < 0 ? -1 : 0);
  if (o.sparse_init != 0) k.sparse_init = o.sparse_init > 0;
  k.fused_level1 = o.fused_level1 > 0 ? 1 : (o.fused_level1 < 0 ? -1 : 0);
  k.longrow = flag(o.longrow, true);
  k.narrow_tile = o.narrow_tile > 0;
  if (o.spmv_grid_cap != 0) k.spmv_grid_cap = std::max(o.spmv_grid_cap, 0);
  k.dia_seg = std::max(o.dia_seg, 0);
  if (o.restrict_seg > 0) k.restrict_seg = k.fused_seg = o.restrict_seg;
  if (o.collapse_min > 0) k.collapse_min = o.collapse_min;
  k.verbose = o.verbose > 0;
  // ---- debug overrides (the variables that used to BE the switches) ----
  auto env = [](const char* name) -> const char* { return getenv((std::string("CSGPU_") + name).c_str()); };
  auto on = [&](const char* name) { return env(name) != nullptr; };
  auto num = [&](const char* name, auto set) {
    if (const char* e = env(name)) set(atof(e));
  };
  num("LAST_SWEEPS", [&](double v) { k.last_level_sweeps = v <= 0 ? -1 : (int)v; });
  num("ENRICH", [&](double v) { if (v <= 0.0) k.enrich = false; });
  num("ENRICH_TAU", [&](double v) { k.enrich_tau = v; });
  num("ENRICH_STEPS", [&](double v) { k.enrich_steps = (int)v; });
  num("DIA25", [&](double v) { if (v != 1.0) k.dia25_min_rows = v <= 0 ? -1 : (int64_t)v; });
  num("DIA25_PF", [&](double v) { k.dia25_prefetch = v != 0.0; });
  num("DIA25_WAVES", [&](double v) { k.dia25_waves = (int)v; });
  if (on("DIA25_NO_J0")) k.dia25_fused_j0 = false;
  num("STREAM", [&](double v) { if (v > 0) k.stream = 1; });
  if (on("NO_STREAM")) k.stream = -1;
  num("STREAM_MIN", [&](double v) { k.stream_min = (int64_t)v; });
  num("HETERO_FP64_FRAC", [&](double v) { k.hetero_fp64_frac = v; });
  num("TAIL_ROWS", [&](double v) { k.tail_rows = v <= 0 ? -1 : (int)v; });
  if (on("NO_POLY_LATTICE")) k.poly_lattice = -1;
  if (on("POLY_LATTICE_ANY_SHAPE")) k.poly_lattice = 1;
  num("POLY_STRENGTH", [&](double v) { k.poly_strength = v; });
  num("POLY_COEF", [&](double v) { k.poly_coef = v; });
  num("POLY_SMIN", [&](double v) { k.poly_smin = v; });
  num("POLY_SMAX", [&](double v) { k.poly_smax = v; });
  if (on("NO_CELLSPACE")) k.cellspace = false;
  if (on("NO_CELLSPACE_FROM_CSR")) k.cellspace_from_csr = false;
  num("CELLSPACE_MIN_FRAC", [&](double v) { k.cellspace_min_frac = v; });
  if (on("NO_LATTICE_L1")) k.lattice_l1 = false;
  num("LATTICE_L1_MIN_ROWS", [&](double v) { k.lattice_l1_min_rows = (int)v; });
  if (on("NO_STENCIL")) k.stencil = false;
  if (on("NO_TWO_PRODUCT")) k.two_product = false;
  if (on("NO_DIRECT_LATTICE")) k.direct_lattice = false;
  if (on("NO_LATTICE_S")) k.lattice_s = false;
  if (on("NO_LATTICE_Q")) k.lattice_q = false;
  if (on("NO_DIRECT_TILES")) k.direct_tiles = false;
  if (on("NO_TILE_PIECES")) k.tile_pieces = false;
  if (on("NO_DIRECT_AT")) k.direct_at = false;
  num("TILE_THETA", [&](double v) { k.tile_theta = v; });
  num("TILE_SPLIT_MIN", [&](double v) { k.tile_split_min = v; });
  if (on("NO_DIRICHLET_COARSE")) k.dirichlet_coarse = false;
  if (on("NO_DEFLATION")) k.deflation = false;
  if (on("NO_TAIL_PROJECTION")) k.tail_projection = false;
  if (on("NO_EXPANDER_PROBE")) k.expander_probe = false;
  if (on("COARSE_JACOBI")) k.coarse_smoother = 2;
  if (on("COARSE_CHEBYSHEV")) k.coarse_smoother = 1;
  num("NU_L1", [&](double v) { k.nu_l1 = (int)v; });
  num("NU_DEEP", [&](double v) { k.nu_deep = (int)v; });
  num("STREAM_HOST_CSR", [&](double v) { k.host_stream_block = (int64_t)v; });
  if (on("WIDE_CSR")) k.wide_csr = true;
  if (on("FIXED_K")) k.fixed_k = true;
  if (on("NO_RECOMPUTE")) k.recompute_ap = false;
  num("FUSED_RESTRICT", [&](double v) { k.fused_restrict = v > 0 ? 1 : -1; });
  num("SPARSE_INIT", [&](double v) { k.sparse_init = v > 0; });
  num("FUSED_LEVEL1", [&](double v) { k.fused_level1 = v > 0 ? 1 : -1; });
  num("COLLAPSE_MIN", [&](double v) { k.collapse_min = (int64_t)v; });
  if (on("NO_LONGROW")) k.longrow = false;
  if (on("NARROW_TILE")) k.narrow_tile = true;
  num("SPMV_GRID_CAP", [&](double v) { k.spmv_grid_cap = (int)v; });
  num("DIA_SEG", [&](double v) { k.dia_seg = (int)v; });
  num("RESTRICT_SEG", [&](double v) { k.restrict_seg = k.fused_seg = (int)v; });
  if (on("VERBOSE")) k.verbose = 1;
  num("PINV_CUT", [&](double v) { k.pinv_cut = v; });
  if (on("KERNEL_GAIN_REF")) k.kernel_gain_ref = true;
  if (on("TAIL_DEBUG")) k.tail_debug = true;
  if (on("GALERKIN_STAGED")) k.galerkin_staged = true;
  if (on("NO_RASTER_TRANSPOSE")) k.raster_transpose = false;
  if (on("NO_ENRICH_FUSED")) k.enrich_fused = false;
  num("APQ_NT", [&](double v) { k.apq_nt = (int)v; });
  num("TIMED_LAUNCHES", [&](double v) { k.timed_launches = (int)v; });
  return k;
}

struct ISolver {
  virtual ~ISolver() {}
  Knobs kn;  // resolved at construction (knobs_from_opts); in scope of the calling thread inside every method
  bool rebuilt_fp64 = false;  // the C API replaced the fp32 hierarchy the caller asked for by an fp64 one (csgpu_info)
  virtual void solve_pairs(const int64_t* src, const int64_t* dst, int64_t npairs, void* volt_out,
                           const int64_t* gather, int64_t ngather, void* gathered_out, void* resist_out,
                           csgpu_stats* stats, const int32_t* weights = nullptr, void* curr_out = nullptr,
                           void* cum_inout = nullptr, void* max_inout = nullptr, void* branch_out = nullptr) = 0;
  virtual void solve_rhs(const void* rhs, int64_t nrhs, void* x_out, csgpu_stats* stats) = 0;
  virtual void solve_grounded(const GroundedJob& job, int64_t nrhs, csgpu_stats* stats) = 0;
  virtual void solve_region_pairs(const int64_t* set_ptr, const int64_t* set_nodes, int64_t nsets, const int64_t* src_set,
                                  const int64_t* dst_set, int64_t npairs, double* resistances, csgpu_stats* stats) = 0;
  virtual void get_info(csgpu_info* info) const = 0;
  virtual double hetero_frac() const = 0;  // Hierarchy::hetero_frac of the handle's hierarchy
  virtual double spmv_bench(int k, int reps) = 0;
  virtual void spmv_host(const void* x, void* y, int k) = 0;
  virtual void raster_nodemap(int32_t* out, int64_t* rows, int64_t* cols) = 0;
  virtual int64_t components(int32_t* out) = 0;
  virtual void solve_raster(const void* source, void* curr_out, void* volt_out, csgpu_stats* stats) = 0;
  virtual void level_spmv_host(int lvl, int which, const void* x, void* y, int k, double* dots) = 0;
  virtual void get_level_matrix(int lvl, int which, int64_t* nrows, int64_t* ncols, int64_t* nnz, int32_t* rowptr,
                                int32_t* colidx, void* vals) const = 0;
  virtual void dia_product_host(const void* z, const void* pin, const double* beta, void* pout, void* y, int k,
                                double* dots) = 0;
};

// index conversion kernels (Julia hands Int64 / 1-based arrays by default: src/run.jl:34, src/config.jl:28)
template <class I>
__global__ __launch_bounds__(256) void convert_index_kernel(int64_t n, const I* __restrict__ in, int base,
                                                            int* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int)(in[i] - (I)base);
}

// a slice of 64-bit (or 32-bit) row pointers rebased to the first entry of a block of rows (streamed host matrices)
template <class I>
__global__ __launch_bounds__(256) void rebase_index_kernel(int64_t n, const I* __restrict__ in, int64_t base,
                                                           int* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int)((int64_t)in[i] - base);
}

// raster coordinates of the nodes of a lattice matrix (column-major numbering: node i = cell (i % R, i / R)); lets a
// matrix handed over by a Julia host (no coordinates) get the same tile-seeded aggregation as a raster built here
__global__ __launch_bounds__(256) void lattice_coords_kernel(int64_t n, int R, int* __restrict__ row,
                                                             int* __restrict__ col) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    row[i] = (int)(i % R);
    col[i] = (int)(i / R);
  }
}

// weight of every row of a cell-space matrix from the cell -> node map (1 = a real node, 0 = NODATA row)
__global__ __launch_bounds__(256) void weights_from_map_kernel(int64_t n, const int* __restrict__ cell2node,
                                                               long long* __restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] = cell2node[i] != 0 ? 1 : 0;
}

// element-wise precision conversion of a CSR matrix (pattern copied)
template <class T, class TP>
inline void convert_csr(const Csr<T>& A, Csr<TP>& B, hipStream_t st) {
  B.nrows = A.nrows;
  B.ncols = A.ncols;
  B.nnz = A.nnz;
  B.rowptr.alloc(A.rowptr.bytes);
  B.col.alloc(A.col.bytes);
  B.val.alloc((size_t)std::max<int64_t>(A.nnz, 1) * sizeof(TP));
  CS_HIP(hipMemcpyAsync(B.rowptr.p, A.rowptr.p, A.rowptr.bytes, hipMemcpyDeviceToDevice, st));
  CS_HIP(hipMemcpyAsync(B.col.p, A.col.p, A.col.bytes, hipMemcpyDeviceToDevice, st));
  if (A.nnz > 0)
    hipLaunchKernelGGL((convert_kernel<T, TP>), dim3(grid_for(A.nnz)), dim3(256), 0, st, A.nnz, A.va(), B.va());
  CS_HIP(hipStreamSynchronize(st));
}

// ---- csgpu_solve_region_pairs helpers --------------------------------------------------------------------------------
// v[node, c] = 1 for the nodes of column c's source set (lists in the layout of mask_grounds_kernel: ptr[K+1], idx[])
template <class T, int K>
__global__ __launch_bounds__(256) void scatter_ones_kernel(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                           T* __restrict__ v) {
  const int total = ptr[K];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int c = 0;
    while (c + 1 < K && e >= ptr[c + 1]) ++c;
    v[(size_t)idx[e] * K + c] = T(1);
  }
}
// v -= y
template <class T>
__global__ __launch_bounds__(256) void subtract_kernel(int64_t total, T* __restrict__ v, const T* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) v[e] -= y[e];
}

// largest and smallest non-zero off-diagonal magnitude of a CSR matrix: part[2 * block] / part[2 * block + 1]
template <class T>
__global__ __launch_bounds__(256) void offdiag_range_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                            const T* __restrict__ va, double* __restrict__ part) {
  __shared__ double s_hi[256], s_lo[256];
  double hi = 0.0, lo = 1e300;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      if (ci[k] == i) continue;
      const double v = fabs((double)va[k]);
      if (v > 0.0) {
        hi = fmax(hi, v);
        lo = fmin(lo, v);
      }
    }
  s_hi[threadIdx.x] = hi;
  s_lo[threadIdx.x] = lo;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s_hi[threadIdx.x] = fmax(s_hi[threadIdx.x], s_hi[threadIdx.x + o]);
      s_lo[threadIdx.x] = fmin(s_lo[threadIdx.x], s_lo[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s_hi[0];
    part[2 * blockIdx.x + 1] = s_lo[0];
  }
}

// T: precision of the CG iteration and of the C-ABI's vectors; TP: precision of the AMG preconditioner.
template <class T, class TP>
struct Solver : ISolver {
  static constexpr bool MIXED = !std::is_same<T, TP>::value;
  int device = 0;
  hipStream_t st = nullptr;
  csgpu_opts opts;
  Csr<T> Aouter;      // the matrix CG sees when MIXED (otherwise level 0 of the hierarchy is used)
  Hierarchy<TP> H;
  double hetero_frac() const override { return H.hetero_frac; }
  Dia<T> dia;         // lattice form of the CG matrix (all-valid rasters; empty otherwise)
  PcgWork<T, TP> W;
  double upload_ms = 0;
  int host_blocks = 0;               // setup_from_host_streamed: blocks of rows the host matrix came in
  int64_t n = 0, nnz = 0;             // dimension / stored entries of the matrix the device solves with
  // Cell space. A raster with NODATA cells (construct_node_map drops every cell with conductance <= 0,
  // src/raster/pairwise.jl:271-301 -- nearly every real landscape has them) is solved on the FULL R x C lattice: every
  // cell keeps a row, a NODATA cell's row is an isolated unknown (diagonal 1, no couplings, right-hand side always 0, so
  // x, r, z, p stay exactly 0 there and every dot product is that of the real graph). The matrix is then a lattice
  // again and the index-free marching kernels (stencil.h, lattice.h) apply; NODATA cells weigh 0 in the aggregates, so
  // the transfer operators and every Galerkin operator are those of the real graph. The C ABI keeps speaking the
  // reference's node numbering: ids and n-vectors are translated at the boundary (node2cell / cell2node).
  bool cellspace = false;
  // Rasters set up through the lattice pipeline (lattice_setup.h) hold NO CSR form of the fine-level matrix: resistance-
  // only pair solves never touch one. Entry points that do (current maps, explicit residual checks of full solutions,
  // region pairs, components, the product hooks) build it from the lattice form on first use (ensure_csr).
  bool csr_ready = true;
  // Test hooks only: level 0 rebuilt by the CSR pipeline of amg_setup.h (A, P, R, Q, Q^T, [S Q] in CSR) for a handle
  // whose level 0 came from the lattice pipeline -- csgpu_get_level_matrix / csgpu_level_spmv_host then compare the two
  // pipelines with each other.
  std::unique_ptr<Hierarchy<TP>> Href;
  int64_t n_api = 0, nnz_api = 0;     // what the caller sees (reference numbering); == n, nnz unless cellspace
  DBuf node2cell, cell2node;          // cellspace: [n_api] column-major cell id of a node; [n] 1-based node id of a cell, 0 = none
  DBuf cellmap;                       // cellspace: row-major [rows][cols], 1-based ROW id of the cell, 0 = no node
  DBuf comp_label_api;                // cellspace: component label per NODE in the reference's terms (csgpu_components)
  int64_t ncomp_api = -1;
  DBuf nodemap;                       // csgpu_raster_setup: row-major [rows][cols], 1-based node id, 0 = no node
  DBuf ground_node;                   // csgpu_raster_setup_grounded: finite ground conductance per node (T)
  DBuf comp_label;                    // connected-component label per node (computed on first use)
  int64_t ncomp = -1;
  int64_t raster_rows = 0, raster_cols = 0;
  // Rasters with short-circuit polygons on the lattice path (poly.h): the handle solves in cell space (cellspace mapping:
  // a polygon's node <-> its representative cell) with PCG projected onto the polygon-wise constants. Resistance-only
  // pair solves run here; every other entry point of such a handle is served by a second solver built on first use from
  // the MERGED graph (the CSR path csgpu_raster_setup_poly always had), from host copies of the two rasters.
  bool poly_proj = false;
  PolyProj proj;                       // device member lists (owned by the buffers below)
  DBuf proj_ptr, proj_cells, proj_chunk_first, proj_chunk_poly, proj_poly_chunk0, proj_chunk_sum, cell_poly;
  std::vector<int> poly_size;          // cells per polygon (dense index)
  std::vector<T> poly_cond_host;       // the rasters the handle was built from (fallback solver)
  std::vector<int32_t> poly_map_host;
  int poly_four = 0, poly_avg = 0, poly_reg = 0;
  std::unique_ptr<Solver<T, TP>> poly_fb;
  std::mutex mu_fb;
  std::mutex mu;

  explicit Solver(const csgpu_opts& o) : opts(o) {
    kn = knobs_from_opts(o);
    if (opts.device >= 0) {
      CS_HIP(hipSetDevice(opts.device));
      device = opts.device;
    } else {
      CS_HIP(hipGetDevice(&device));
    }
    CS_HIP(hipStreamCreate(&st));
  }
  ~Solver() override {
    if (st) hipStreamDestroy(st);
  }

  SetupParams setup_params() const {
    SetupParams sp;
    sp.max_levels = opts.max_levels;
    sp.max_coarse = opts.max_coarse;
    sp.aggregation = opts.aggregation;
    sp.theta = opts.theta;
    sp.omega_p = opts.omega_p;
    sp.omega_s = opts.omega_s;
    sp.two_product = opts.nu_pre == 1 && opts.nu_post == 1 && kn.two_product;
    // coarse levels: Chebyshev weights for the sweep counts the solve phase will run (CSGPU_COARSE_JACOBI=1: the damped
    // Jacobi of round 1, A/B knob)
    const int nuc = opts.nu_coarse > 0 ? opts.nu_coarse : 1;
    sp.nu_l1 = kn.nu_l1 > 0 ? kn.nu_l1 : nuc;
    sp.nu_deep = kn.nu_deep > 0 ? kn.nu_deep : nuc + 1;
    sp.coarse_chebyshev = kn.coarse_smoother != 2;
    return sp;
  }
  PcgParams pcg_params(int K = 1) const {
    PcgParams pp;
    pp.rtol = opts.rtol;
    pp.atol = opts.atol;
    pp.criterion = opts.criterion;
    pp.itmax = opts.itmax;
    // auto: poll every iteration once an iteration takes milliseconds (a 20 us host round trip is then free and no
    // surplus iteration is ever enqueued), every 4th in the launch-latency regime
    pp.check_every = opts.check_every > 0 ? opts.check_every : ((int64_t)n * K >= ((int64_t)1 << 25) ? 1 : 4);
    pp.nu_pre = opts.nu_pre;
    pp.nu_post = opts.nu_post;
    pp.nu_coarse = opts.nu_coarse > 0 ? opts.nu_coarse : 1;
    pp.use_graph = opts.use_graph;
    pp.proj = poly_proj ? &proj : nullptr;
    return pp;
  }
  const Dia<T>* dia_ptr() const { return dia.n > 0 ? &dia : nullptr; }

  // ---- boundary translation (no-ops unless cellspace) ----------------------------------------------------------------
  // node ids of the caller -> row ids of the device matrix
  struct Ids {
    const int64_t* p = nullptr;
    std::vector<int64_t> own;
    const int64_t& operator[](int64_t k) const { return p[k]; }
  };
  Ids rows_of(const int64_t* ids, int64_t cnt) {
    Ids r;
    r.p = ids;
    if (!cellspace || cnt <= 0) return r;
    DBuf in = dalloc<int64_t>((size_t)cnt), out = dalloc<int64_t>((size_t)cnt);
    CS_HIP(hipMemcpyAsync(in.p, ids, (size_t)cnt * sizeof(int64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(map_ids_kernel, dim3(grid_for(cnt)), dim3(256), 0, st, cnt, (const int64_t*)dptr<int64_t>(in),
                       (const int*)dptr<int>(node2cell), dptr<int64_t>(out));
    r.own.resize((size_t)cnt);
    CS_HIP(hipMemcpyAsync(r.own.data(), out.p, (size_t)cnt * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
    r.p = r.own.data();
    return r;
  }
  // host column-major n_api x ncols  ->  device column-major n x ncols (zeros at the NODATA rows)
  void upload_cols(const T* host, int64_t ncols, T* dev) {
    if (!cellspace) {
      CS_HIP(hipMemcpyAsync(dev, host, (size_t)n * ncols * sizeof(T), hipMemcpyHostToDevice, st));
      return;
    }
    DBuf tmp((size_t)n_api * ncols * sizeof(T));
    CS_HIP(hipMemcpyAsync(tmp.p, host, tmp.bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((cells_from_nodes_kernel<T>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n, n_api,
                       (const int*)dptr<int>(cell2node), (const T*)dptr<T>(tmp), (int)ncols, dev);
    CS_HIP(hipStreamSynchronize(st));  // tmp is released on return
  }
  // device column-major n x ncols  ->  host column-major n_api x ncols; blocks until the copy has landed
  void download_cols(const T* dev, int64_t ncols, T* host) {
    if (!cellspace) {
      CS_HIP(hipMemcpyAsync(host, dev, (size_t)n * ncols * sizeof(T), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      return;
    }
    DBuf tmp((size_t)n_api * ncols * sizeof(T));
    hipLaunchKernelGGL((nodes_from_cells_kernel<T>), dim3(grid_for(n_api * ncols)), dim3(256), 0, st, n, n_api,
                       (const int*)dptr<int>(node2cell), dev, (int)ncols, dptr<T>(tmp));
    CS_HIP(hipMemcpyAsync(host, tmp.p, tmp.bytes, hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
  }
  const int* raster_rowmap() const { return cellspace ? (const int*)dptr<int>(cellmap) : (const int*)dptr<int>(nodemap); }

  // Lattice form of the CG matrix: period known (raster built here, every cell valid) or detected from the band
  // structure of a host-built matrix (candidates around the dominant band offset found by spmv_block_order).
  void detect_lattice(const Csr<T>& A, int known_period) {
    dia = Dia<T>();
    if (!kn.stencil || known_period < 0) return;
    if (known_period > 0) {
      dia_from_csr(A, known_period, dia, st, /*trusted=*/true);
      return;
    }
    // node 0 of a lattice is coupled to 1, R and (8 neighbours) R+1: its last column gives the period
    if (A.nrows < 16 || A.nnz < 1) return;
    const int len0 = read_int(A.rp() + 1, st);
    if (len0 < 2 || len0 > 4) return;
    const long long last = read_int(A.ci() + (len0 - 1), st);
    for (long long cand : {last - 1, last})
      if (cand >= 4 && cand < (1ll << 30) && dia_from_csr(A, (int)cand, dia, st)) return;
  }

  // Host matrices with 2^31 stored entries and more (the reference's use_64bit_indexing, src/run.jl:34, src/config.jl:28: a
  // raster pairwise problem above 238 M cells; VERDICT r4 missing 8). The device never holds such a matrix in CSR form --
  // its int32 entry offsets end at 2^31 -- and does not need to: a raster graph handed over with the raster cell of every
  // node (csgpu_opts.node_row / node_col, which the Julia binding sends for every raster problem) is scattered into the
  // lattice form (5 values per cell) block of rows by block of rows, each block uploaded, converted and dropped again, and
  // the index-free pipeline (lattice_setup.h) builds the hierarchy from there -- the handle csgpu_raster_setup would have
  // built from the raster, with the caller's node numbering at the boundary. Covers all-valid rasters (numbered column-
  // major, as construct_node_map does) and rasters with NODATA cells; anything else of that size (polygons, networks,
  // options that switch the index-free pipeline off) is refused with the reason. CSGPU_STREAM_HOST_CSR=<entries per
  // block> sends matrices of any size down this path (tests; 0 / unset: only those that need it); a matrix below 2^31
  // entries that the path declines takes the ordinary one.
  bool setup_from_host_streamed(const void* rowptr, const void* colidx, const void* vals, int64_t n_, int64_t nnz_,
                                int idx_bytes, int index_base, int64_t block_entries) {
    auto t0 = std::chrono::steady_clock::now();
    const bool must = nnz_ >= ((int64_t)1 << 31);
    auto decline = [&](const char* why) {
      CS_REQUIRE(!must, CSGPU_BAD_ARGS, std::string("a matrix with 2^31 stored entries or more must be a raster graph the "
                 "index-free pipeline takes (node_row / node_col given, one node per cell, couplings between neighbouring "
                 "cells only, default smoother / aggregation options): ") + why);
      return false;
    };
    if (!(opts.node_row && opts.node_col)) return decline("csgpu_opts.node_row / node_col missing");
    auto rp_at = [&](int64_t i) -> int64_t {
      return (idx_bytes == 8 ? ((const int64_t*)rowptr)[i] : (int64_t)((const int32_t*)rowptr)[i]) - index_base;
    };
    if (rp_at(0) != 0 || rp_at(n_) != nnz_) return decline("row pointers do not span nnz");
    // the cut search below bisects on the host's row pointers and the blocks' rebased int32 pointers trust them: one pass
    // over them first (monotone, at most 9 entries per row -- a raster node has no more)
    for (int64_t i = 0, prev = 0; i < n_; ++i) {
      const int64_t cur = rp_at(i + 1);
      if (cur < prev || cur - prev > 9) return decline("row pointers not monotone, or a row with more entries than a raster node can have");
      prev = cur;
    }
    n = n_api = n_;
    nnz = nnz_api = nnz_;
    DBuf drow((size_t)n_ * sizeof(int)), dcol((size_t)n_ * sizeof(int));
    CS_HIP(hipMemcpyAsync(drow.p, opts.node_row, (size_t)n_ * sizeof(int), hipMemcpyHostToDevice, st));
    CS_HIP(hipMemcpyAsync(dcol.p, opts.node_col, (size_t)n_ * sizeof(int), hipMemcpyHostToDevice, st));
    const int* prow = dptr<int>(drow);
    const int* pcol = dptr<int>(dcol);
    DBuf mm = dalloc<int>(4);
    const int init[4] = {-0x7fffffff, -0x7fffffff, 0x7fffffff, 0x7fffffff};
    CS_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(coord_range_kernel, dim3(grid_for(n_)), dim3(256), 0, st, (int)n_, prow, pcol, dptr<int>(mm));
    int hbox[4];
    CS_HIP(hipMemcpyAsync(hbox, mm.p, sizeof(hbox), hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
    const int r0 = hbox[2], c0 = hbox[3];
    const int64_t R = (int64_t)hbox[0] - r0 + 1, C = (int64_t)hbox[1] - c0 + 1, ncells = R * C;
    if (R < 6 || C < 6 || ncells >= ((int64_t)1 << 31) - 1 || ncells < n_) return decline("bounding box of the coordinates");
    const bool all_valid = ncells == n_;
    if (!want_lattice_pipeline(R, C) || (!all_valid && !want_cellspace(n_, ncells, R, C)))
      return decline("options / shape outside the index-free pipeline");
    DBuf n2c((size_t)n_ * sizeof(int)), c2n = dalloc<int>((size_t)ncells), bad = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(c2n.p, 0, (size_t)ncells * sizeof(int), st));
    CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(csr_cells_kernel, dim3(grid_for(n_)), dim3(256), 0, st, (int)n_, (int)R, r0, c0, prow, pcol,
                       dptr<int>(n2c), dptr<int>(c2n), dptr<int>(bad));
    if (all_valid)
      hipLaunchKernelGGL(identity_numbering_kernel, dim3(grid_for(n_)), dim3(256), 0, st, (int)n_, (const int*)dptr<int>(n2c),
                         dptr<int>(bad));
    check_launch("streamed host matrix: cells");
    if (read_int(dptr<int>(bad), st) != 0) return decline("two nodes on one cell, or an all-valid raster not numbered column-major");
    DBuf rows((size_t)ncells * 5 * sizeof(T));
    CS_HIP(hipMemsetAsync(rows.p, 0, rows.bytes, st));
    // blocks of whole rows holding at most block_entries entries (a single row never exceeds 9)
    const int64_t cap = std::max<int64_t>(block_entries, 64);
    int64_t max_rows = 0;
    std::vector<int64_t> cuts(1, 0);
    while (cuts.back() < n_) {
      const int64_t a = cuts.back(), lim = rp_at(a) + cap;
      int64_t lo = a + 1, hi = n_;            // largest b with rp[b] <= rp[a] + cap
      while (lo < hi) {
        const int64_t mid = (lo + hi + 1) / 2;
        if (rp_at(mid) <= lim) lo = mid; else hi = mid - 1;
      }
      if (rp_at(lo) > lim) return decline("a row with more entries than a raster node can have");
      cuts.push_back(lo);
      max_rows = std::max(max_rows, lo - a);
    }
    {
      DBuf raw((size_t)std::max<int64_t>(cap, max_rows + 1) * idx_bytes), brp = dalloc<int>((size_t)max_rows + 1),
           bci = dalloc<int>((size_t)cap), bva((size_t)cap * sizeof(T));
      for (size_t b = 0; b + 1 < cuts.size(); ++b) {
        const int64_t a = cuts[b], e = cuts[b + 1], nloc = e - a, k0 = rp_at(a), cnt = rp_at(e) - k0;
        CS_HIP(hipMemcpyAsync(raw.p, (const char*)rowptr + (size_t)a * idx_bytes, (size_t)(nloc + 1) * idx_bytes,
                              hipMemcpyHostToDevice, st));
        if (idx_bytes == 8)
          hipLaunchKernelGGL((rebase_index_kernel<int64_t>), dim3(grid_for(nloc + 1)), dim3(256), 0, st, nloc + 1,
                             dptr<int64_t>(raw), k0 + index_base, dptr<int>(brp));
        else
          hipLaunchKernelGGL((rebase_index_kernel<int32_t>), dim3(grid_for(nloc + 1)), dim3(256), 0, st, nloc + 1,
                             dptr<int32_t>(raw), k0 + index_base, dptr<int>(brp));
        if (cnt > 0) {
          CS_HIP(hipMemcpyAsync(raw.p, (const char*)colidx + (size_t)k0 * idx_bytes, (size_t)cnt * idx_bytes,
                                hipMemcpyHostToDevice, st));
          if (idx_bytes == 8)
            hipLaunchKernelGGL((rebase_index_kernel<int64_t>), dim3(grid_for(cnt)), dim3(256), 0, st, cnt, dptr<int64_t>(raw),
                               (int64_t)index_base, dptr<int>(bci));
          else
            hipLaunchKernelGGL((rebase_index_kernel<int32_t>), dim3(grid_for(cnt)), dim3(256), 0, st, cnt, dptr<int32_t>(raw),
                               (int64_t)index_base, dptr<int>(bci));
          CS_HIP(hipMemcpyAsync(bva.p, (const char*)vals + (size_t)k0 * sizeof(T), (size_t)cnt * sizeof(T),
                                hipMemcpyHostToDevice, st));
        }
        hipLaunchKernelGGL((csr_block_to_cell_dia_kernel<T>), dim3(grid_for(nloc)), dim3(256), 0, st, (int)a, (int)nloc, (int)R,
                           (const int*)dptr<int>(brp), (const int*)dptr<int>(bci), (const T*)dptr<T>(bva), prow, pcol,
                           (const int*)dptr<int>(n2c), dptr<T>(rows), dptr<int>(bad));
        check_launch("streamed host matrix: block of rows");
        CS_HIP(hipStreamSynchronize(st));  // (the staging buffers are reused by the next block)
      }
    }
    DBuf size0;
    if (!all_valid) {
      size0.alloc((size_t)ncells * sizeof(long long));
      hipLaunchKernelGGL((cell_identity_kernel<T>), dim3(grid_for(ncells)), dim3(256), 0, st, ncells, (const int*)dptr<int>(c2n),
                         dptr<T>(rows), dptr<long long>(size0));
      check_launch("streamed host matrix: identity rows");
    }
    if (read_int(dptr<int>(bad), st) != 0) {
      n = n_api;
      nnz = nnz_api;
      return decline("a coupling between cells that are not neighbours (polygons, networks), or a row without diagonal");
    }
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    opts.node_row = opts.node_col = nullptr;  // host pointers are never retained
    cellspace = !all_valid;
    n = ncells;
    nnz = nnz_api + (ncells - n_api);
    if (cellspace) {
      node2cell = std::move(n2c);
      cell2node = std::move(c2n);
    }
    dia.n = ncells;
    dia.R = (int)R;
    dia.rows = std::move(rows);
    if (lattice_pipeline_hierarchy(size0, R, C)) {
      host_blocks = (int)cuts.size() - 1;
      if (kn.verbose)
        fprintf(stderr, "csgpu: host matrix streamed in %zu block(s) of rows (%lld stored entries, %lld x %lld cells)\n",
                cuts.size() - 1, (long long)nnz_api, (long long)R, (long long)C);
      return true;
    }
    cellspace = false;
    n = n_api;
    nnz = nnz_api;
    node2cell.release();
    cell2node.release();
    dia = Dia<T>();
    return decline("the index-free pipeline declined the matrix");
  }

  void setup_from_host(const void* rowptr, const void* colidx, const void* vals, int64_t n_, int64_t nnz_,
                       int idx_bytes, int index_base) {
    KnobScope ks(&kn);
    auto t0 = std::chrono::steady_clock::now();
    {
      const int64_t forced = kn.host_stream_block;  // (csgpu_opts.host_stream_block: a test / tuning knob)
      if (nnz_ >= ((int64_t)1 << 31) || forced > 0) {
        const csgpu_opts keep = opts;
        try {
          if (setup_from_host_streamed(rowptr, colidx, vals, n_, nnz_, idx_bytes, index_base,
                                       forced > 0 ? forced : ((int64_t)1 << 28)))
            return;
        } catch (const Error& e) {
          // a matrix below 2^31 entries sent down the streamed path by the knob: out of memory there is not fatal, the
          // ordinary path below is tried; everything else, and every failure of a matrix that MUST stream, is
          if (nnz_ >= ((int64_t)1 << 31) || e.code != CSGPU_OOM) throw;
          (void)hipGetLastError();
        }
        opts = keep;
      }
    }
    n = n_api = n_;
    nnz = nnz_api = nnz_;
    Csr<T> A;
    A.nrows = A.ncols = (int)n;
    A.nnz = nnz;
    A.rowptr.alloc((size_t)(n + 1) * sizeof(int));
    A.col.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(int));
    A.val.alloc((size_t)std::max<int64_t>(nnz, 1) * sizeof(T));
    if (idx_bytes == 4 && index_base == 0) {
      CS_HIP(hipMemcpyAsync(A.rp(), rowptr, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(A.ci(), colidx, (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    } else {
      DBuf raw((size_t)std::max<int64_t>(std::max<int64_t>(nnz, n + 1), 1) * idx_bytes);
      CS_HIP(hipMemcpyAsync(raw.p, rowptr, (size_t)(n + 1) * idx_bytes, hipMemcpyHostToDevice, st));
      if (idx_bytes == 8)
        hipLaunchKernelGGL((convert_index_kernel<int64_t>), dim3(grid_for(n + 1)), dim3(256), 0, st, n + 1,
                           dptr<int64_t>(raw), index_base, A.rp());
      else
        hipLaunchKernelGGL((convert_index_kernel<int32_t>), dim3(grid_for(n + 1)), dim3(256), 0, st, n + 1,
                           dptr<int32_t>(raw), index_base, A.rp());
      CS_HIP(hipStreamSynchronize(st));
      if (nnz > 0) {
        CS_HIP(hipMemcpyAsync(raw.p, colidx, (size_t)nnz * idx_bytes, hipMemcpyHostToDevice, st));
        if (idx_bytes == 8)
          hipLaunchKernelGGL((convert_index_kernel<int64_t>), dim3(grid_for(nnz)), dim3(256), 0, st, nnz,
                             dptr<int64_t>(raw), index_base, A.ci());
        else
          hipLaunchKernelGGL((convert_index_kernel<int32_t>), dim3(grid_for(nnz)), dim3(256), 0, st, nnz,
                             dptr<int32_t>(raw), index_base, A.ci());
      }
      check_launch("index conversion");
      CS_HIP(hipStreamSynchronize(st));
    }
    if (nnz > 0) CS_HIP(hipMemcpyAsync(A.va(), vals, (size_t)nnz * sizeof(T), hipMemcpyHostToDevice, st));
    DBuf drow, dcol;
    const int* prow = nullptr;
    const int* pcol = nullptr;
    if (opts.node_row && opts.node_col) {
      drow.alloc((size_t)n * sizeof(int));
      dcol.alloc((size_t)n * sizeof(int));
      CS_HIP(hipMemcpyAsync(drow.p, opts.node_row, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(dcol.p, opts.node_col, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
      prow = dptr<int>(drow);
      pcol = dptr<int>(dcol);
    }
    CS_HIP(hipStreamSynchronize(st));
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    opts.node_row = opts.node_col = nullptr;  // host pointers are never retained
    // Conductances over more than five decades leave an fp32 hierarchy nothing to work with (7 digits): found by
    // tools/fuzz_networks.py ON THE DEVICE at the end of round 4 -- a 696-node star with conductances over six decades,
    // fp32 hierarchy: "relative residual 338", the same case converging in 23 iterations on the emulator build (where the
    // compiler does not contract a * b + c). Matrices handed over in CSR form have no strength test to measure their
    // heterogeneity (Hierarchy::hetero_frac), so the contrast of the off-diagonal entries stands in for it: above 1e5 the
    // handle reports itself heterogeneous and the C API rebuilds it with an fp64 hierarchy (hetero_wants_fp64).
    double contrast = 1.0;
    if (MIXED && nnz > 0) {
      const int g = std::min(grid_for(n), 1024);
      DBuf part = dalloc<double>((size_t)2 * g);
      hipLaunchKernelGGL((offdiag_range_kernel<T>), dim3(g), dim3(256), 0, st, (int)n, (const int*)A.rp(), (const int*)A.ci(),
                         (const T*)A.va(), dptr<double>(part));
      std::vector<double> hp((size_t)2 * g);
      CS_HIP(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      double hi = 0.0, lo = 1e300;
      for (int b = 0; b < g; ++b) {
        hi = std::max(hi, hp[(size_t)2 * b]);
        lo = std::min(lo, hp[(size_t)2 * b + 1]);
      }
      if (hi > 0.0 && lo < 1e300) contrast = hi / lo;
    }
    if (prow && pcol && setup_cellspace_from_csr(A, prow, pcol)) return;
    finish_setup(std::move(A), prow, pcol, 0);
    // (rasters -- lattice detected, strength test run -- are judged by that test: hetero_frac >= 0)
    if (contrast > 1e5 && H.hetero_frac < 0.0) H.hetero_frac = 1.0;
  }

  // The Julia host path for rasters WITH NODATA cells (lattice_setup.h, csr_to_cell_dia_kernel): a compact CSR Laplacian
  // with the raster cell of every node becomes the cell-space lattice matrix csgpu_raster_setup would have built, and takes
  // the index-free pipeline and the marching kernels from there (an all-valid raster needs none of this: its lattice
  // period is detected from the matrix, detect_lattice). False -- nothing changed -- when the matrix is no such raster
  // (polygons: couplings between cells that are not neighbours; two nodes on one cell; too few valid cells).
  bool setup_cellspace_from_csr(Csr<T>& A, const int* prow, const int* pcol) {
    if (!kn.cellspace_from_csr || n_api < 36 || A.nnz < 1) return false;
    DBuf mm = dalloc<int>(4);
    const int init[4] = {-0x7fffffff, -0x7fffffff, 0x7fffffff, 0x7fffffff};
    CS_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(coord_range_kernel, dim3(grid_for(n_api)), dim3(256), 0, st, (int)n_api, prow, pcol, dptr<int>(mm));
    int h[4];
    CS_HIP(hipMemcpyAsync(h, mm.p, sizeof(h), hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
    // the lattice is the bounding box of the nodes' cells (a connected component of a larger raster brings its own box)
    const int r0 = h[2], c0 = h[3];
    const int64_t R = (int64_t)h[0] - r0 + 1, C = (int64_t)h[1] - c0 + 1, ncells = R * C;
    if (R < 6 || C < 6 || ncells >= ((int64_t)1 << 31) - 1 || ncells <= n_api) return false;
    if (!(want_cellspace(n_api, ncells, R, C) && want_lattice_pipeline(R, C))) return false;
    DBuf n2c((size_t)n_api * sizeof(int)), c2n = dalloc<int>((size_t)ncells), bad = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(c2n.p, 0, (size_t)ncells * sizeof(int), st));
    CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(csr_cells_kernel, dim3(grid_for(n_api)), dim3(256), 0, st, (int)n_api, (int)R, r0, c0, prow, pcol,
                       dptr<int>(n2c), dptr<int>(c2n), dptr<int>(bad));
    DBuf rows((size_t)ncells * 5 * sizeof(T)), size0((size_t)ncells * sizeof(long long));
    CS_HIP(hipMemsetAsync(rows.p, 0, rows.bytes, st));
    hipLaunchKernelGGL((csr_to_cell_dia_kernel<T>), dim3(grid_for(n_api)), dim3(256), 0, st, (int)n_api, (int)R, A.rp(), A.ci(),
                       A.va(), prow, pcol, (const int*)dptr<int>(n2c), dptr<T>(rows), dptr<int>(bad));
    hipLaunchKernelGGL((cell_identity_kernel<T>), dim3(grid_for(ncells)), dim3(256), 0, st, ncells, (const int*)dptr<int>(c2n),
                       dptr<T>(rows), dptr<long long>(size0));
    check_launch("CSR -> cell space");
    if (read_int(dptr<int>(bad), st) != 0) return false;
    cellspace = true;
    n = ncells;
    nnz = nnz_api + (ncells - n_api);
    node2cell = std::move(n2c);
    cell2node = std::move(c2n);
    dia.n = ncells;
    dia.R = (int)R;
    dia.rows = std::move(rows);
    if (lattice_pipeline_hierarchy(size0, R, C)) {
      A = Csr<T>();  // (compact numbering: of no use to a cell-space handle; its CSR form is built on demand)
      return true;
    }
    cellspace = false;
    n = n_api;
    nnz = nnz_api;
    node2cell.release();
    cell2node.release();
    dia = Dia<T>();
    return false;
  }

  const Csr<T>& cg_matrix() const {
    if constexpr (MIXED) {
      return Aouter;
    } else {
      return H.levels[0].A;
    }
  }
  // CSR form of the fine-level matrix, built from the lattice form when the handle was set up without one
  void ensure_csr() {
    if (csr_ready) return;
    CS_REQUIRE(nnz < ((int64_t)1 << 31), CSGPU_BAD_ARGS,
               "this call needs the CSR form of the matrix, which a raster of this size does not have (2^31 stored "
               "entries); resistance-only csgpu_solve_pairs works without it");
    Csr<T> A;
    dia_to_csr(dia, A, st);
    CS_HIP(hipStreamSynchronize(st));
    if constexpr (MIXED) {
      Aouter = std::move(A);
    } else {
      H.levels[0].A = std::move(A);
    }
    csr_ready = true;
  }

  // known_period: raster height when the matrix was built here from an all-valid raster, 0 = detect, -1 = no lattice
  void finish_setup(Csr<T>&& A, const int* prow, const int* pcol, int known_period, const long long* size0 = nullptr) {
    detect_lattice(A, known_period);
    DBuf lrow, lcol;
    // A lattice detected from the matrix numbers ITS nodes column-major on an R x (n / R) raster; coordinates the caller
    // handed over are those of the full raster the component was cut out of (a connected component below an all-NODATA row
    // or right of an all-NODATA column: offset rows / columns) and would put the direct tiles outside the level's extent --
    // found by tools/fuzz_rasters.py at 60..220 cells a side on the device at the end of round 4 (out-of-bounds writes,
    // "rows without a stored diagonal entry"; the small cases of round 3 never reached a component that is an all-valid
    // rectangle with an offset). The lattice's own coordinates are the ones that match sp.grid_rows / grid_cols below.
    if (dia.n > 0) {
      lrow.alloc((size_t)n * sizeof(int));
      lcol.alloc((size_t)n * sizeof(int));
      hipLaunchKernelGGL(lattice_coords_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, dia.R, dptr<int>(lrow), dptr<int>(lcol));
      prow = dptr<int>(lrow);
      pcol = dptr<int>(lcol);
    }
    SetupParams sp = setup_params();
    sp.size0 = size0;
    sp.n_real = size0 ? n_api : 0;
    if (dia.n > 0) {  // raster extent known: left-over cells stay with their own 3x3 tile (agg_pass2_kernel)
      sp.grid_rows = dia.R;
      sp.grid_cols = (int)(n / dia.R);
    } else if (known_period > 0 && n % known_period == 0) {  // all-valid raster built here, lattice product switched off
      sp.grid_rows = known_period;
      sp.grid_cols = (int)(n / known_period);
    }
    sp.lattice_s = sp.two_product && dia.n > 0 && kn.lattice_s;
    // Chebyshev weights on the coarse levels need a clean restriction chain: measured on MI355X they save 4-17 % of the
    // iterations on rasters up to 5000^2 (7 levels) with an fp32 hierarchy and at 10000^2 (8 levels) with an fp64 one,
    // but the fp32 hierarchy of a 10000^2 raster lost with them (11.8 / 21 instead of 10.9 / 11 iterations, mean /
    // slowest column): the large weights of the polynomial amplify the spurious near-kernel component that eight fp32
    // restrictions put on the coarse right-hand sides (amg_setup.h, component_candidates). The coarse tail now projects
    // that component out (tail.h, tail_project) and with it the same hierarchy reaches the fp64 count (10.77 / 11,
    // profiles/r2_coarse_chebyshev.json) -- measured on 48 pairs at the very end of the round, after the full evidence
    // set had been taken with the rule below in place; CSGPU_COARSE_CHEBYSHEV=1 lifts it, and the next full evidence run
    // should.
    if (sizeof(TP) == 4 && n > 30000000 && kn.coarse_smoother != 1) sp.coarse_chebyshev = false;
    if constexpr (MIXED) {
      Csr<TP> Ap;
      convert_csr(A, Ap, st);
      Aouter = std::move(A);
      amg_setup(H, std::move(Ap), sp, prow, pcol, st);
    } else {
      amg_setup(H, std::move(A), sp, prow, pcol, st);
    }
    // The coarsest-level Dirichlet correction (pcg.h, DirichletCoarse) probes G_k = 1_f' A_g 1_f as the sum of the penalty
    // vector, which assumes that A annihilates the per-component candidate. A handle with finite ground conductances on
    // its diagonal (csgpu_raster_setup_grounded) leaks there and the correction would be over-weighted by (G + leak) / G
    // (SPD either way, but slower): such handles keep the plain deflated pseudo-inverse.
    if (ground_node.p) H.dir_ncomp = 0;
    Level<TP>& L0 = H.levels[0];
    if (sp.lattice_s && L0.agg0.p) {
      // two-product level from the lattice: b_c = Q^T b and out = S b + Q x_c with index-free Q (lattice.h) and S in
      // lattice form (stencil.h); when the aggregates are not the regular tiles the CSR forms are built instead
      hipEvent_t e0, e1;
      CS_HIP(hipEventCreate(&e0));
      CS_HIP(hipEventCreate(&e1));
      CS_HIP(hipEventRecord(e0, st));
      const bool ok = kn.lattice_q && lattice_q_from_csr(L0.Q, (const int*)dptr<int>(L0.agg0), dia.R, (int)(n / dia.R),
                                                          L0.Ql, st);
      L0.agg0.release();
      if (kn.verbose)
        fprintf(stderr, "csgpu: two-product level on a %d x %lld lattice: index-free Q %s\n", dia.R, (long long)(n / dia.R),
                ok ? "built" : "not applicable (CSR forms)");
      if (ok) {
        dia_build_s(dia, (const TP*)dptr<TP>(L0.dinv), L0.omega, L0.Sdia, st);
      } else {
        build_qt_matrix(L0, st);
        build_sq_matrix(L0, st);
      }
      CS_HIP(hipEventRecord(e1, st));
      CS_HIP(hipEventSynchronize(e1));
      float ms = 0;
      CS_HIP(hipEventElapsedTime(&ms, e0, e1));
      H.setup_ms += ms;
      hipEventDestroy(e0);
      hipEventDestroy(e1);
    }
  }

  // Cell space is chosen for rasters WITH NODATA cells when the index-free fine level applies (two-product V(1,1) level,
  // lattice product not switched off) and most cells are valid: the marching kernels stream every cell, so below
  // ~half-full rasters the compact CSR path moves fewer bytes (CSGPU_CELLSPACE_MIN_FRAC, default 0.5; CSGPU_NO_CELLSPACE=1
  // keeps the compact numbering of round 2).
  bool want_cellspace(int64_t nvalid, int64_t ncells, int64_t R, int64_t C) const {
    if (!kn.cellspace || !kn.stencil) return false;
    if (nvalid == ncells || R < 6 || C < 6) return false;
    if (!(opts.nu_pre == 1 && opts.nu_post == 1 && kn.two_product)) return false;
    if (opts.aggregation == CSGPU_AGG_MIS2 || opts.theta != 0.0) return false;
    const double minfrac = kn.cellspace_min_frac;
    return (double)nvalid >= minfrac * (double)ncells;
  }

  // The index-free pipeline needs what the lattice two-product level needs (and its A/B knobs off).
  bool want_lattice_pipeline(int64_t R, int64_t C) const {
    if (!(kn.direct_lattice && kn.stencil && kn.lattice_s && kn.lattice_q && kn.two_product && kn.direct_tiles)) return false;
    if (!(opts.nu_pre == 1 && opts.nu_post == 1)) return false;
    if (opts.aggregation == CSGPU_AGG_MIS2 || opts.theta != 0.0) return false;
    return R >= 6 && C >= 6 && R * C > opts.max_coarse && opts.max_levels >= 2;
  }

  // Hierarchy of a lattice-form matrix `dia` (R x C, every cell a row; size0: weights of a cell-space matrix, may be
  // empty) through the index-free pipeline of lattice_setup.h. False when the pipeline declined (nothing changed).
  bool lattice_pipeline_hierarchy(DBuf& size0, int64_t R, int64_t C) {
    hipEvent_t e0, e1;
    CS_HIP(hipEventCreate(&e0));
    CS_HIP(hipEventCreate(&e1));
    CS_HIP(hipEventRecord(e0, st));

    SetupParams sp = setup_params();
    sp.grid_rows = (int)R;
    sp.grid_cols = (int)C;
    sp.lattice_s = true;
    sp.size0 = cellspace ? (const long long*)dptr<long long>(size0) : (const long long*)nullptr;
    sp.n_real = cellspace ? n_api : 0;
    if (sizeof(TP) == 4 && n > 30000000 && kn.coarse_smoother != 1) sp.coarse_chebyshev = false;  // (see finish_setup)
    // polygon handles: the strength-aware tiles are what keeps the strengthened polygon interiors (poly.h) out of the
    // aggregates of their surroundings -- always on, however few cells the polygons cover
    if (poly_proj) sp.tile_split_min = -1.0;
    SetupCarry carry;
    const bool ok = lattice_level0_setup<T, TP>(H, dia, (int)R, (int)C, cellspace ? dptr<long long>(size0) : (long long*)nullptr,
                                                sp, carry, st);
    if (ok) {
      amg_setup_levels(H, sp, nullptr, nullptr, carry, st);
      CS_HIP(hipEventRecord(e1, st));
      CS_HIP(hipEventSynchronize(e1));
      float ms = 0;
      CS_HIP(hipEventElapsedTime(&ms, e0, e1));
      H.setup_ms = ms;
      if constexpr (MIXED) Aouter.nrows = Aouter.ncols = (int)n;
      csr_ready = false;
      if (kn.verbose)
        fprintf(stderr, "csgpu: %lld x %lld raster through the index-free pipeline (%lld rows, %lld nodes, %d levels)\n",
                (long long)R, (long long)C, (long long)n, (long long)n_api, (int)H.levels.size());
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ok;
  }

  // Raster -> lattice form -> hierarchy, no CSR (lattice_setup.h). dcond / dground: device rasters (row-major); node: the
  // exclusive scan of the valid flags (column-major). Returns false when the pipeline declined (the caller falls back).
  bool setup_lattice_direct(DBuf& dcond, DBuf& dground, DBuf& node, int64_t R, int64_t C, int four, int avg_res, int reg,
                            std::chrono::steady_clock::time_point t0, int colmajor = 0) {
    const int64_t ncells = R * C;
    const int gc = grid_for(ncells);
    if (cellspace) {
      cellmap.alloc((size_t)ncells * sizeof(int));
      node2cell.alloc((size_t)n_api * sizeof(int));
      cell2node.alloc((size_t)ncells * sizeof(int));
    }
    hipLaunchKernelGGL((raster_maps_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       (const int*)dptr<int>(node), dptr<int>(nodemap), cellspace ? dptr<int>(cellmap) : (int*)nullptr,
                       cellspace ? dptr<int>(node2cell) : (int*)nullptr, cellspace ? dptr<int>(cell2node) : (int*)nullptr,
                       colmajor);
    dia.n = ncells;
    dia.R = (int)R;
    dia.rows.alloc((size_t)ncells * 5 * sizeof(T));
    if (dground.p) ground_node.alloc((size_t)ncells * sizeof(T));
    DBuf part = dalloc<double>(gc), cnt = dalloc<unsigned long long>(gc), size0;
    hipLaunchKernelGGL((raster_dia_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                       (const T*)dptr<T>(dcond), dground.p ? (const T*)dptr<T>(dground) : (const T*)nullptr, dptr<T>(dia.rows),
                       dground.p ? dptr<T>(ground_node) : (T*)nullptr, dptr<double>(part), dptr<unsigned long long>(cnt),
                       colmajor);
    if (cellspace) size0.alloc((size_t)ncells * sizeof(long long));
    hipLaunchKernelGGL((raster_dia_finish_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       dptr<T>(dia.rows), (const double*)dptr<double>(part), gc,
                       reg ? (double)std::numeric_limits<T>::epsilon() : 0.0,
                       cellspace ? dptr<long long>(size0) : (long long*)nullptr, colmajor);
    std::vector<unsigned long long> hc((size_t)gc);
    CS_HIP(hipMemcpyAsync(hc.data(), cnt.p, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    check_launch("raster -> lattice form");
    CS_HIP(hipStreamSynchronize(st));
    nnz_api = 0;
    for (unsigned long long v : hc) nnz_api += (int64_t)v;
    nnz = nnz_api + (n - n_api);  // entries a CSR form of the device matrix would hold (NODATA rows: their diagonal)
    dcond.release();
    dground.release();
    node.release();
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (lattice_pipeline_hierarchy(size0, R, C)) return true;
    // declined (cannot happen for tile aggregates): CSR pipeline on the CSR form of the same matrix
    CS_REQUIRE(nnz < ((int64_t)1 << 31), CSGPU_BAD_ARGS, "raster too large for the CSR pipeline");
    Csr<T> A;
    dia_to_csr(dia, A, st);
    DBuf drow((size_t)n * sizeof(int)), dcol((size_t)n * sizeof(int));
    hipLaunchKernelGGL(lattice_coords_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, (int)R, dptr<int>(drow), dptr<int>(dcol));
    dia = Dia<T>();
    finish_setup(std::move(A), dptr<int>(drow), dptr<int>(dcol), (int)R,
                 cellspace ? (const long long*)dptr<long long>(size0) : (const long long*)nullptr);
    return true;
  }

  void setup_from_raster(const void* cond, int64_t R, int64_t C, int four, int avg_res, int reg,
                         const void* ground = nullptr) {
    KnobScope ks(&kn);
    auto t0 = std::chrono::steady_clock::now();
    const int64_t ncells = R * C;
    DBuf dcond((size_t)ncells * sizeof(T));
    CS_HIP(hipMemcpyAsync(dcond.p, cond, (size_t)ncells * sizeof(T), hipMemcpyHostToDevice, st));
    DBuf dground;
    if (ground) {
      dground.alloc((size_t)ncells * sizeof(T));
      CS_HIP(hipMemcpyAsync(dground.p, ground, (size_t)ncells * sizeof(T), hipMemcpyHostToDevice, st));
    }
    // node numbering: exclusive scan of the valid-cell flags in column-major order
    DBuf node = dalloc<int>((size_t)ncells + 1);
    CS_HIP(hipMemsetAsync(node.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
    const int gc = grid_for(ncells);
    // (the raster kernels of the index-free pipeline read the column-major copies: coalesced along raster columns)
    const bool transposed = kn.raster_transpose && R >= 32 && C >= 32;
    DBuf dcondT, dgroundT;
    if (transposed) {
      const int gt = (int)std::min<int64_t>(((R + 31) / 32) * ((C + 31) / 32), 65536);
      dcondT.alloc((size_t)ncells * sizeof(T));
      hipLaunchKernelGGL((raster_transpose_kernel<T>), dim3(gt), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                         dptr<T>(dcondT));
      if (ground) {
        dgroundT.alloc((size_t)ncells * sizeof(T));
        hipLaunchKernelGGL((raster_transpose_kernel<T>), dim3(gt), dim3(256), 0, st, (int)R, (int)C,
                           (const T*)dptr<T>(dground), dptr<T>(dgroundT));
      }
    }
    hipLaunchKernelGGL((raster_valid_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C,
                       transposed ? dptr<T>(dcondT) : dptr<T>(dcond), dptr<int>(node), transposed ? 1 : 0);
    DBuf total = dalloc<int>(1);
    exclusive_scan_i32(dptr<int>(node), ncells + 1, st, dptr<int>(total));
    n_api = read_int(dptr<int>(total), st);
    CS_REQUIRE(n_api > 0, CSGPU_BAD_ARGS, "raster has no cell with positive conductance");
    cellspace = want_cellspace(n_api, ncells, R, C);
    n = cellspace ? ncells : n_api;
    raster_rows = R;
    raster_cols = C;
    nodemap.alloc((size_t)ncells * sizeof(int));
    if (n == ncells && want_lattice_pipeline(R, C)) {
      // every cell a row: the whole level 0 is built from the lattice form, without a CSR matrix (lattice_setup.h)
      if (setup_lattice_direct(transposed ? dcondT : dcond, transposed ? dgroundT : dground, node, R, C, four, avg_res, reg, t0,
                               transposed ? 1 : 0))
        return;
    }
    CS_REQUIRE(ncells * 9 < ((int64_t)1 << 31), CSGPU_BAD_ARGS,
               "raster too large for the CSR pipeline (int32 entry offsets); only all-valid / mostly-valid rasters without "
               "polygons, with the default two-product fine level, go through the index-free pipeline");
    if (cellspace) {
      cellmap.alloc((size_t)ncells * sizeof(int));
      node2cell.alloc((size_t)n_api * sizeof(int));
      cell2node.alloc((size_t)ncells * sizeof(int));
    }
    Csr<T> A;
    A.nrows = A.ncols = (int)n;
    A.rowptr.alloc((size_t)(n + 1) * sizeof(int));
    CS_HIP(hipMemsetAsync(A.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
    hipLaunchKernelGGL((raster_count_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, dptr<T>(dcond),
                       dptr<int>(node), A.rp(), dptr<int>(nodemap), cellspace ? 1 : 0,
                       cellspace ? dptr<int>(cellmap) : (int*)nullptr, cellspace ? dptr<int>(node2cell) : (int*)nullptr,
                       cellspace ? dptr<int>(cell2node) : (int*)nullptr);
    exclusive_scan_i32(A.rp(), n + 1, st, dptr<int>(total));
    nnz = read_int(dptr<int>(total), st);
    nnz_api = nnz - (n - n_api);  // (the NODATA rows hold one entry each)
    A.nnz = nnz;
    A.col.alloc((size_t)nnz * sizeof(int));
    A.val.alloc((size_t)nnz * sizeof(T));
    DBuf drow((size_t)n * sizeof(int)), dcol((size_t)n * sizeof(int));
    if (ground) ground_node.alloc((size_t)n * sizeof(T));
    hipLaunchKernelGGL((raster_fill_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                       dptr<T>(dcond), dptr<int>(node), A.rp(), A.ci(), A.va(), dptr<int>(drow), dptr<int>(dcol),
                       ground ? (const T*)dptr<T>(dground) : (const T*)nullptr,
                       ground ? dptr<T>(ground_node) : (T*)nullptr, cellspace ? 1 : 0);
    if (reg) {
      // (cell space: the NODATA diagonals are still 0 here, so the norm is that of the real graph's entries; the shift
      // they receive is overwritten below)
      const int g = grid_for(nnz);
      DBuf part = dalloc<double>(g);
      hipLaunchKernelGGL((dot_kernel<T, 1, false>), dim3(g), dim3(256), 0, st, nnz, (const T*)A.va(), (const T*)A.va(),
                         dptr<double>(part), (const T*)nullptr, (const T*)nullptr, (double*)nullptr);
      hipLaunchKernelGGL((add_scalar_kernel<T>), dim3(g), dim3(256), 0, st, nnz, A.va(), dptr<double>(part), g,
                         (double)std::numeric_limits<T>::epsilon());
      CS_HIP(hipStreamSynchronize(st));
    }
    DBuf size0;
    if (cellspace) {
      size0.alloc((size_t)n * sizeof(long long));
      hipLaunchKernelGGL((raster_identity_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                         (const int*)A.rp(), A.va(), dptr<long long>(size0));
    }
    check_launch("raster build");
    CS_HIP(hipStreamSynchronize(st));
    dcond.release();
    node.release();
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // every cell a row: row i couples to i+-1, i+-(R-1), i+-R, i+-(R+1) -> lattice form (stencil.h)
    finish_setup(std::move(A), dptr<int>(drow), dptr<int>(dcol), n == ncells ? (int)R : -1,
                 cellspace ? (const long long*)dptr<long long>(size0) : (const long long*)nullptr);
    if (cellspace && !(dia.n > 0 && H.levels[0].lattice_two_product()) && kn.verbose)
      fprintf(stderr, "csgpu: cell-space raster without the index-free fine level (CSR kernels on %lld rows)\n", (long long)n);
  }

  // Polygons on the lattice path (poly.h). Returns false -- nothing changed -- when the handle does not qualify (options
  // that switch the index-free pipeline off, fewer than half of the cells carrying a row, no merged polygon) and the
  // merged CSR graph is to be built instead. CSGPU_NO_POLY_LATTICE=1: A/B knob; CSGPU_POLY_STRENGTH: factor on the
  // polygon-interior edges of the preconditioner's matrix (default 100; iterations on a 180^2 raster with 12 polygons: 23.6
  // at 1 -- the plain projection --, 14.4 at 10, 12.0 at 60-100, 12.6 at 300, 16.8 at 1000, 24 at 1e4; 9 without polygons,
  // 10.9 on the merged CSR graph).
  bool setup_poly_lattice(const void* cond, const int32_t* polymap, int64_t R, int64_t C, int four, int avg_res, int reg,
                          bool is_fallback) {
    if (is_fallback || kn.poly_lattice < 0 || !want_lattice_pipeline(R, C)) return false;
    if (!kn.cellspace || ground_node.p) return false;
    auto t0 = std::chrono::steady_clock::now();
    const int64_t ncells = R * C;
    const T* hc = (const T*)cond;
    // ---- host: which polygons own a node, their member cells (column-major ids, ascending), rows of the lattice
    int32_t maxid = 0;
    for (int64_t k = 0; k < ncells; ++k) maxid = std::max(maxid, polymap[k]);
    if (maxid <= 0 || maxid >= kMaxPolyId) return false;
    std::vector<char> has_valid((size_t)maxid + 1, 0);
    int64_t nvalid = 0;
    for (int64_t k = 0; k < ncells; ++k) {
      const bool v = hc[k] > T(0);
      nvalid += v ? 1 : 0;
      if (v && polymap[k] > 0) has_valid[(size_t)polymap[k]] = 1;
    }
    std::vector<int> dense((size_t)maxid + 1, -1);
    int npoly = 0;
    for (int32_t p = 1; p <= maxid; ++p)
      if (has_valid[(size_t)p]) dense[(size_t)p] = npoly++;
    if (npoly == 0) return false;
    std::vector<int> count((size_t)npoly, 0);
    int64_t nrows = 0;
    for (int64_t k = 0; k < ncells; ++k) {
      const int32_t p = polymap[k];
      const bool member = p > 0 && dense[(size_t)p] >= 0;
      if (member) ++count[(size_t)dense[(size_t)p]];
      if (member || hc[k] > T(0)) ++nrows;
    }
    const double minfrac = kn.cellspace_min_frac;
    if ((double)nrows < minfrac * (double)ncells) return false;
    // (polygons of a single cell are ordinary nodes: nothing to project)
    std::vector<int> hptr((size_t)npoly + 1, 0);
    for (int p = 0; p < npoly; ++p) hptr[(size_t)p + 1] = hptr[(size_t)p] + (count[(size_t)p] >= 2 ? count[(size_t)p] : 0);
    std::vector<int> hcells((size_t)std::max(hptr[(size_t)npoly], 1));
    {
      std::vector<int> cur(hptr.begin(), hptr.end() - 1);
      // row-major scan, column-major ids: every polygon's list is sorted afterwards (fixed summation order)
      for (int64_t i = 0; i < R; ++i)
        for (int64_t j = 0; j < C; ++j) {
          const int32_t p = polymap[i * C + j];
          if (p <= 0 || dense[(size_t)p] < 0) continue;
          const int d = dense[(size_t)p];
          if (count[(size_t)d] < 2) continue;
          hcells[(size_t)cur[(size_t)d]++] = (int)(j * R + i);
        }
      for (int p = 0; p < npoly; ++p) std::sort(hcells.begin() + hptr[(size_t)p], hcells.begin() + hptr[(size_t)p + 1]);
    }
    std::vector<int> hchunk_first, hchunk_poly, hpoly_chunk0((size_t)npoly + 1, 0);
    for (int p = 0; p < npoly; ++p) {
      hpoly_chunk0[(size_t)p] = (int)hchunk_first.size();
      for (int m = hptr[(size_t)p]; m < hptr[(size_t)p + 1]; m += kPolyChunk) {
        hchunk_first.push_back(m);
        hchunk_poly.push_back(p);
      }
    }
    hpoly_chunk0[(size_t)npoly] = (int)hchunk_first.size();
    // ---- host: shape of every polygon. CORE cells = members all of whose lattice neighbours (with a row) are members too.
    std::vector<int> core((size_t)npoly, 0);
    bool wire_like = false;
    for (int p = 0; p < npoly; ++p) {
      int64_t i0 = R, i1 = -1, j0 = C, j1 = -1;
      for (int m = hptr[(size_t)p]; m < hptr[(size_t)p + 1]; ++m) {
        const int64_t k = hcells[(size_t)m], ci = k % R, cj = k / R;
        i0 = std::min(i0, ci);
        i1 = std::max(i1, ci);
        j0 = std::min(j0, cj);
        j1 = std::max(j1, cj);
        bool all = true;
        for (int dj = -1; dj <= 1 && all; ++dj)
          for (int di = -1; di <= 1 && all; ++di) {
            if ((di == 0 && dj == 0) || (four && di != 0 && dj != 0)) continue;
            const int64_t ii = ci + di, jj = cj + dj;
            if (ii < 0 || ii >= R || jj < 0 || jj >= C) continue;
            const int32_t q = polymap[ii * C + jj];
            const bool row = hc[ii * C + jj] > T(0) || (q > 0 && dense[(size_t)q] >= 0);
            if (row && !(q > 0 && dense[(size_t)q] == p)) all = false;
          }
        core[(size_t)p] += all ? 1 : 0;
      }
      // A LONG, THIN polygon (a river, a road, the sliver one polygon leaves of another) short-circuits cells many tiles
      // apart, and its cells are minority pieces of their tiles: no strength of its interior edges makes the cell-space
      // hierarchy see that. Measured at 5000^2 with 200 strips of up to 80 cells (profiles/r4_polygon_strip_width_sweep.txt):
      // 2 cells wide 45 iterations against 13 on the merged graph (429 ms per batch against 201), 3-5 wide a tie
      // (19-20 iterations, ~200 ms either way), 8 wide 16 iterations and 163 ms against 217. Such rasters keep the
      // merged CSR graph: the lattice path requires every polygon that is at least 8 cells long to have 30 % core cells
      // (strips from 3 cells of width on: never worse than a tie with the merged graph).
      const int64_t extent = std::max(i1 - i0, j1 - j0) + 1;
      if (count[(size_t)p] >= 2 && extent >= 8 && (double)core[(size_t)p] < 0.30 * (double)count[(size_t)p]) wire_like = true;
    }
    // A polygon in SEVERAL PIECES (one id used in two places) may be the only link between two parts of the raster: the
    // merged graph is connected through its node, the cell-space lattice -- and with it the hierarchy, whose coarsest
    // pseudo-inverse treats every connected component of ITS graph as a singular block -- is not. Contiguous polygons
    // cannot change the connectivity (their interior edges, all positive, hold them together), so only rasters whose
    // polygons are all contiguous (in the raster's own 4- / 8-neighbourhood) take the lattice path.
    bool split_polygon = false;
    {
      std::vector<int> stack;
      std::vector<char> seen;
      for (int p = 0; p < npoly && !split_polygon; ++p) {
        const int lo = hptr[(size_t)p], hi = hptr[(size_t)p + 1];
        if (hi - lo < 2) continue;
        seen.assign((size_t)(hi - lo), 0);
        stack.clear();
        stack.push_back(0);
        seen[0] = 1;
        int reached = 1;
        while (!stack.empty()) {
          const int m = stack.back();
          stack.pop_back();
          const int64_t k = hcells[(size_t)(lo + m)], ci = k % R, cj = k / R;
          for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
              if ((di == 0 && dj == 0) || (four && di != 0 && dj != 0)) continue;
              const int64_t ii = ci + di, jj = cj + dj;
              if (ii < 0 || ii >= R || jj < 0 || jj >= C) continue;
              const int key = (int)(jj * R + ii);
              const auto it = std::lower_bound(hcells.begin() + lo, hcells.begin() + hi, key);
              if (it == hcells.begin() + hi || *it != key) continue;
              const int q = (int)(it - (hcells.begin() + lo));
              if (!seen[(size_t)q]) {
                seen[(size_t)q] = 1;
                ++reached;
                stack.push_back(q);
              }
            }
        }
        if (reached != hi - lo) split_polygon = true;
      }
    }
    if (split_polygon && kn.poly_lattice <= 0) {
      if (kn.verbose) fprintf(stderr, "csgpu: a polygon in several pieces: merged CSR graph instead of the lattice path\n");
      return false;
    }
    if (wire_like && kn.poly_lattice <= 0) {
      if (kn.verbose) fprintf(stderr, "csgpu: a long thin polygon: merged CSR graph instead of the lattice path\n");
      return false;
    }
    // ---- device: labels and node numbering exactly as the merged path computes them (raster.h)
    const int gc = grid_for(ncells);
    DBuf dcond((size_t)ncells * sizeof(T)), dpoly = dalloc<int>((size_t)ncells);
    CS_HIP(hipMemcpyAsync(dcond.p, cond, (size_t)ncells * sizeof(T), hipMemcpyHostToDevice, st));
    CS_HIP(hipMemcpyAsync(dpoly.p, polymap, (size_t)ncells * sizeof(int), hipMemcpyHostToDevice, st));
    DBuf rep = dalloc<int>((size_t)maxid + 2), present = dalloc<int>((size_t)maxid + 2);
    hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(maxid + 2)), dim3(256), 0, st, dptr<int>(rep), (int64_t)maxid + 2,
                       0x7fffffff);
    hipLaunchKernelGGL((poly_rep_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       (const int*)dpoly.p, dptr<int>(rep));
    DBuf label = dalloc<int>((size_t)ncells), flag = dalloc<int>((size_t)ncells + 1), total = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(flag.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
    hipLaunchKernelGGL((poly_label_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       (const int*)dpoly.p, (const int*)dptr<int>(rep), dptr<int>(label), dptr<int>(flag));
    exclusive_scan_i32(dptr<int>(flag), ncells + 1, st, dptr<int>(total));
    n_api = read_int(dptr<int>(total), st);
    CS_REQUIRE(n_api > 0, CSGPU_BAD_ARGS, "raster has no cell with positive conductance");
    hipLaunchKernelGGL(poly_present_kernel, dim3(grid_for(maxid + 1)), dim3(256), 0, st, maxid, (const int*)dptr<int>(rep),
                       dptr<int>(present));
    CS_HIP(hipMemsetAsync(dptr<int>(present) + maxid + 1, 0, sizeof(int), st));
    exclusive_scan_i32(dptr<int>(present), (int64_t)maxid + 2, st, dptr<int>(total));
    CS_REQUIRE(read_int(dptr<int>(total), st) == npoly, CSGPU_INTERNAL, "polygon count of the device and the host scan differ");
    raster_rows = R;
    raster_cols = C;
    n = ncells;
    cellspace = true;
    poly_proj = true;
    nodemap.alloc((size_t)ncells * sizeof(int));
    cellmap.alloc((size_t)ncells * sizeof(int));
    node2cell.alloc((size_t)n_api * sizeof(int));
    cell2node.alloc((size_t)ncells * sizeof(int));
    cell_poly.alloc((size_t)ncells * sizeof(int));
    {
      DBuf node = dalloc<int>((size_t)ncells), drow((size_t)n_api * sizeof(int)), dcol((size_t)n_api * sizeof(int));
      DBuf node_poly = dalloc<int>((size_t)n_api), poly_node = dalloc<int>((size_t)std::max(npoly, 1));
      hipLaunchKernelGGL(poly_node_kernel, dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const int*)dpoly.p,
                         (const int*)dptr<int>(rep), (const int*)dptr<int>(label), (const int*)dptr<int>(flag),
                         (const int*)dptr<int>(present), dptr<int>(node), dptr<int>(nodemap), dptr<int>(drow), dptr<int>(dcol),
                         dptr<int>(node_poly), dptr<int>(poly_node));
      hipLaunchKernelGGL(poly_cell_maps_kernel, dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const int*)dptr<int>(label),
                         (const int*)dptr<int>(node), (const int*)dptr<int>(node_poly), dptr<int>(cellmap),
                         dptr<int>(cell2node), dptr<int>(cell_poly));
      hipLaunchKernelGGL(poly_node2cell_kernel, dim3(grid_for(n_api)), dim3(256), 0, st, n_api, (int)R,
                         (const int*)dptr<int>(drow), (const int*)dptr<int>(dcol), dptr<int>(node2cell));
      check_launch("polygon maps");
      CS_HIP(hipStreamSynchronize(st));
    }
    // ---- lattice form with strengthened polygon interiors, regularisation, weights
    // Strength of the polygon-interior edges in the preconditioner's matrix, per polygon. Measured on MI355X at 5000^2
    // (profiles/r4_polygon_strength_sweep.txt): polygons up to 100 cells across want ~1000 (11.2 iterations against 10.4
    // without polygons; 18.5 at 100, 18.7 at 1e4), polygons up to 20 cells across ~100 (13.9; 33 at 1000), polygons of a
    // few cells less still -- the hierarchy copes with a stiff inclusion once it spans several 3x3 tiles, while a stiff
    // speck inside a tile only unbalances the smoother. Hence strength = coef * (member cells), clamped.
    // CSGPU_POLY_STRENGTH fixes one value for all polygons (A/B knob).
    const double s_fixed = kn.poly_strength, s_coef = kn.poly_coef, s_min = kn.poly_smin, s_max = kn.poly_smax;
    // ... of a BLOB: the member count is discounted by the polygon's share of core cells -- full weight from 40 % core
    // cells on (a square of 5 x 5 has 36 %), proportionally less below.
    std::vector<double> hstrength((size_t)npoly);
    for (int p = 0; p < npoly; ++p) {
      const double cnt_p = (double)count[(size_t)p];
      const double blob = std::min(1.0, 2.5 * (double)core[(size_t)p] / std::max(cnt_p, 1.0));
      hstrength[(size_t)p] = s_fixed > 0.0 ? s_fixed : std::min(s_max, std::max(s_min, s_coef * cnt_p * blob));
    }
    DBuf dstrength = dalloc<double>((size_t)npoly);
    CS_HIP(hipMemcpyAsync(dstrength.p, hstrength.data(), hstrength.size() * sizeof(double), hipMemcpyHostToDevice, st));
    const double strength = s_fixed;
    dia.n = ncells;
    dia.R = (int)R;
    dia.rows.alloc((size_t)ncells * 5 * sizeof(T));
    DBuf part = dalloc<double>(gc), cnt = dalloc<unsigned long long>(gc), size0;
    hipLaunchKernelGGL((raster_dia_poly_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                       (const T*)dptr<T>(dcond), (const int*)dptr<int>(label), (const int*)dptr<int>(cell_poly),
                       (const double*)dptr<double>(dstrength), dptr<T>(dia.rows), dptr<double>(part),
                       dptr<unsigned long long>(cnt));
    size0.alloc((size_t)ncells * sizeof(long long));
    hipLaunchKernelGGL((raster_dia_poly_finish_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C,
                       (const int*)dptr<int>(label), dptr<T>(dia.rows), (const double*)dptr<double>(part), gc,
                       reg ? (double)std::numeric_limits<T>::epsilon() : 0.0, dptr<long long>(size0));
    std::vector<unsigned long long> hcnt((size_t)gc);
    CS_HIP(hipMemcpyAsync(hcnt.data(), cnt.p, hcnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    check_launch("raster + polygons -> lattice form");
    CS_HIP(hipStreamSynchronize(st));
    nnz_api = 0;
    for (unsigned long long v : hcnt) nnz_api += (int64_t)v;  // (entries of the cell-space matrix outside the polygons' interiors)
    nnz = nnz_api + (n - nrows);
    // ---- member lists of the projection
    auto up = [&](DBuf& d, const std::vector<int>& h) {
      d.alloc(std::max<size_t>(h.size(), 1) * sizeof(int));
      if (!h.empty()) CS_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, st));
    };
    up(proj_ptr, hptr);
    up(proj_cells, hcells);
    up(proj_chunk_first, hchunk_first);
    up(proj_chunk_poly, hchunk_poly);
    up(proj_poly_chunk0, hpoly_chunk0);
    proj_chunk_sum.alloc(std::max<size_t>(hchunk_first.size(), 1) * kMaxK * sizeof(double));
    CS_HIP(hipStreamSynchronize(st));
    proj.npoly = npoly;
    proj.nchunks = (int)hchunk_first.size();
    proj.ptr = dptr<int>(proj_ptr);
    proj.cells = dptr<int>(proj_cells);
    proj.chunk_first = dptr<int>(proj_chunk_first);
    proj.chunk_poly = dptr<int>(proj_chunk_poly);
    proj.poly_chunk0 = dptr<int>(proj_poly_chunk0);
    proj.chunk_sum = dptr<double>(proj_chunk_sum);
    poly_size = count;
    dcond.release();
    dpoly.release();
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!lattice_pipeline_hierarchy(size0, R, C)) {
      // (cannot happen for tile aggregates) undo: the merged CSR path takes over
      poly_proj = cellspace = false;
      dia = Dia<T>();
      proj = PolyProj();
      return false;
    }
    // the fallback solver (merged CSR graph) is built from these on first use
    poly_cond_host.assign(hc, hc + ncells);
    poly_map_host.assign(polymap, polymap + ncells);
    poly_four = four;
    poly_avg = avg_res;
    poly_reg = reg;
    if (kn.verbose)
      fprintf(stderr, "csgpu: %d polygons (%d member cells, %d chunks) on the lattice path, interior strength %s %g\n", npoly,
              hptr[(size_t)npoly], proj.nchunks, strength > 0 ? "fixed at" : "per polygon, coefficient", strength > 0 ? strength : s_coef);
    return true;
  }

  // the merged-graph solver behind a polygon handle of the lattice path (entry points other than resistance-only pairs)
  Solver<T, TP>& poly_fallback() {
    std::lock_guard<std::mutex> lk(mu_fb);
    if (!poly_fb) {
      poly_fb.reset(new Solver<T, TP>(opts));
      poly_fb->setup_from_raster_poly(poly_cond_host.data(), poly_map_host.data(), raster_rows, raster_cols, poly_four,
                                      poly_avg, poly_reg, /*is_fallback=*/true);
    }
    return *poly_fb;
  }

  // csgpu_raster_setup_poly: raster with short-circuit polygons, graph built on the device (raster.h, second half)
  void setup_from_raster_poly(const void* cond, const int32_t* polymap, int64_t R, int64_t C, int four, int avg_res,
                              int reg, bool is_fallback = false) {
    KnobScope ks(&kn);
    if (setup_poly_lattice(cond, polymap, R, C, four, avg_res, reg, is_fallback)) return;
    auto t0 = std::chrono::steady_clock::now();
    const int64_t ncells = R * C;
    const int gc = grid_for(ncells);
    DBuf dcond((size_t)ncells * sizeof(T)), dpoly = dalloc<int>((size_t)ncells);
    CS_HIP(hipMemcpyAsync(dcond.p, cond, (size_t)ncells * sizeof(T), hipMemcpyHostToDevice, st));
    CS_HIP(hipMemcpyAsync(dpoly.p, polymap, (size_t)ncells * sizeof(int), hipMemcpyHostToDevice, st));
    DBuf dmax = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(dmax.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(poly_max_kernel, dim3(gc), dim3(256), 0, st, ncells, (const int*)dpoly.p, dptr<int>(dmax));
    const int maxid = read_int(dptr<int>(dmax), st);
    CS_REQUIRE(maxid < kMaxPolyId, CSGPU_BAD_ARGS, "polygon ids must be below 2^26");
    // 1. representative cell of every polygon, 2. labels, flags, node ids
    DBuf rep = dalloc<int>((size_t)maxid + 2), present = dalloc<int>((size_t)maxid + 2);
    hipLaunchKernelGGL(fill_int_kernel, dim3(grid_for(maxid + 2)), dim3(256), 0, st, dptr<int>(rep), (int64_t)maxid + 2,
                       0x7fffffff);
    hipLaunchKernelGGL((poly_rep_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       (const int*)dpoly.p, dptr<int>(rep));
    DBuf label = dalloc<int>((size_t)ncells), flag = dalloc<int>((size_t)ncells + 1), total = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(flag.p, 0, ((size_t)ncells + 1) * sizeof(int), st));
    hipLaunchKernelGGL((poly_label_kernel<T>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const T*)dptr<T>(dcond),
                       (const int*)dpoly.p, (const int*)dptr<int>(rep), dptr<int>(label), dptr<int>(flag));
    exclusive_scan_i32(dptr<int>(flag), ncells + 1, st, dptr<int>(total));
    n = n_api = read_int(dptr<int>(total), st);
    CS_REQUIRE(n > 0, CSGPU_BAD_ARGS, "raster has no cell with positive conductance");
    hipLaunchKernelGGL(poly_present_kernel, dim3(grid_for(maxid + 1)), dim3(256), 0, st, maxid, (const int*)dptr<int>(rep),
                       dptr<int>(present));
    CS_HIP(hipMemsetAsync(dptr<int>(present) + maxid + 1, 0, sizeof(int), st));
    exclusive_scan_i32(dptr<int>(present), (int64_t)maxid + 2, st, dptr<int>(total));
    const int npoly = read_int(dptr<int>(total), st);  // polygons that own a node; `present` now holds their dense index
    raster_rows = R;
    raster_cols = C;
    nodemap.alloc((size_t)ncells * sizeof(int));
    DBuf node = dalloc<int>((size_t)ncells), drow((size_t)n * sizeof(int)), dcol((size_t)n * sizeof(int));
    DBuf node_poly = dalloc<int>((size_t)n), poly_node = dalloc<int>((size_t)std::max(npoly, 1));
    hipLaunchKernelGGL(poly_node_kernel, dim3(gc), dim3(256), 0, st, (int)R, (int)C, (const int*)dpoly.p,
                       (const int*)dptr<int>(rep), (const int*)dptr<int>(label), (const int*)dptr<int>(flag),
                       (const int*)dptr<int>(present), dptr<int>(node), dptr<int>(nodemap), dptr<int>(drow), dptr<int>(dcol),
                       dptr<int>(node_poly), dptr<int>(poly_node));
    // 3./4. row lengths: ordinary cells directly, polygon nodes from their sorted candidate lists
    Csr<T> A;
    A.nrows = A.ncols = (int)n;
    A.rowptr.alloc((size_t)(n + 1) * sizeof(int));
    CS_HIP(hipMemsetAsync(A.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
    hipLaunchKernelGGL((poly_cell_rows_kernel<T, false>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                       (const T*)dptr<T>(dcond), (const int*)dptr<int>(node), (const int*)dptr<int>(node_poly),
                       (const int*)dptr<int>(label), A.rp(), (const int*)nullptr, (int*)nullptr, (T*)nullptr);
    DBuf cnt = dalloc<int>((size_t)npoly + 1), padded = dalloc<int>((size_t)npoly + 2), seg = dalloc<int64_t>((size_t)npoly + 2);
    DBuf ckey, cval;
    std::vector<int64_t> hseg((size_t)npoly + 1, 0);
    if (npoly > 0) {
      CS_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)npoly + 1) * sizeof(int), st));
      hipLaunchKernelGGL((poly_candidates_kernel<T, true>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                         (const T*)dptr<T>(dcond), (const int*)dptr<int>(node), (const int*)dptr<int>(node_poly),
                         (const int*)dptr<int>(label), dptr<int>(cnt), (const int64_t*)nullptr,
                         (unsigned long long*)nullptr, (double*)nullptr);
      hipLaunchKernelGGL(poly_pad_kernel, dim3(grid_for(npoly + 1)), dim3(256), 0, st, npoly, (const int*)dptr<int>(cnt),
                         dptr<int>(padded));
      // offsets of the padded segments (64-bit on the host: a handful of integers)
      std::vector<int> hpad((size_t)npoly + 1);
      CS_HIP(hipMemcpyAsync(hpad.data(), padded.p, ((size_t)npoly + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      for (int p = 0; p < npoly; ++p) hseg[p + 1] = hseg[p] + hpad[p];
      CS_HIP(hipMemcpyAsync(seg.p, hseg.data(), ((size_t)npoly + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
      const int64_t tot = std::max<int64_t>(hseg[npoly], 1);
      ckey.alloc((size_t)tot * sizeof(unsigned long long));
      cval.alloc((size_t)tot * sizeof(double));
      CS_HIP(hipMemsetAsync(ckey.p, 0xff, (size_t)tot * sizeof(unsigned long long), st));  // padding sorts last
      CS_HIP(hipMemsetAsync(cval.p, 0, (size_t)tot * sizeof(double), st));
      CS_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)npoly + 1) * sizeof(int), st));
      hipLaunchKernelGGL((poly_candidates_kernel<T, false>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                         (const T*)dptr<T>(dcond), (const int*)dptr<int>(node), (const int*)dptr<int>(node_poly),
                         (const int*)dptr<int>(label), dptr<int>(cnt), (const int64_t*)dptr<int64_t>(seg),
                         dptr<unsigned long long>(ckey), dptr<double>(cval));
      hipLaunchKernelGGL(poly_sort_kernel, dim3(npoly), dim3(256), 0, st, (const int64_t*)dptr<int64_t>(seg),
                         dptr<unsigned long long>(ckey), dptr<double>(cval));
      hipLaunchKernelGGL((poly_rows_kernel<T, true>), dim3(npoly), dim3(64), 0, st, (const int64_t*)dptr<int64_t>(seg),
                         (const int*)dptr<int>(cnt), (const unsigned long long*)dptr<unsigned long long>(ckey),
                         (const double*)dptr<double>(cval), (const int*)dptr<int>(poly_node), A.rp(), (const int*)nullptr,
                         (int*)nullptr, (T*)nullptr);
    }
    exclusive_scan_i32(A.rp(), n + 1, st, dptr<int>(total));
    nnz = nnz_api = read_int(dptr<int>(total), st);
    A.nnz = nnz;
    A.col.alloc((size_t)nnz * sizeof(int));
    A.val.alloc((size_t)nnz * sizeof(T));
    hipLaunchKernelGGL((poly_cell_rows_kernel<T, true>), dim3(gc), dim3(256), 0, st, (int)R, (int)C, four, avg_res,
                       (const T*)dptr<T>(dcond), (const int*)dptr<int>(node), (const int*)dptr<int>(node_poly),
                       (const int*)dptr<int>(label), (int*)nullptr, (const int*)A.rp(), A.ci(), A.va());
    if (npoly > 0)
      hipLaunchKernelGGL((poly_rows_kernel<T, false>), dim3(npoly), dim3(64), 0, st, (const int64_t*)dptr<int64_t>(seg),
                         (const int*)dptr<int>(cnt), (const unsigned long long*)dptr<unsigned long long>(ckey),
                         (const double*)dptr<double>(cval), (const int*)dptr<int>(poly_node), (int*)nullptr,
                         (const int*)A.rp(), A.ci(), A.va());
    if (reg) {
      const int g = grid_for(nnz);
      DBuf part = dalloc<double>(g);
      hipLaunchKernelGGL((dot_kernel<T, 1, false>), dim3(g), dim3(256), 0, st, nnz, (const T*)A.va(), (const T*)A.va(),
                         dptr<double>(part), (const T*)nullptr, (const T*)nullptr, (double*)nullptr);
      hipLaunchKernelGGL((add_scalar_kernel<T>), dim3(g), dim3(256), 0, st, nnz, A.va(), dptr<double>(part), g,
                         (double)std::numeric_limits<T>::epsilon());
    }
    check_launch("raster build (polygons)");
    CS_HIP(hipStreamSynchronize(st));
    upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    finish_setup(std::move(A), dptr<int>(drow), dptr<int>(dcol), -1);
  }

  void raster_nodemap(int32_t* out, int64_t* rows, int64_t* cols) override {
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    CS_REQUIRE(nodemap.p != nullptr, CSGPU_BAD_ARGS, "handle was not built by csgpu_raster_setup");
    if (rows) *rows = raster_rows;
    if (cols) *cols = raster_cols;
    if (out) CS_HIP(hipMemcpy(out, nodemap.p, (size_t)raster_rows * raster_cols * sizeof(int), hipMemcpyDeviceToHost));
  }

  void ensure_components() {
    if (ncomp >= 0) return;
    ensure_csr();
    const Csr<T>& A = cg_matrix();
    comp_label.alloc((size_t)n * sizeof(int));
    ncomp = connected_components<T>((int)n, A.rp(), A.ci(), A.va(), dptr<int>(comp_label), st);
  }

  int64_t components(int32_t* out) override {
    if (poly_proj) return poly_fallback().components(out);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    ensure_components();
    if (!cellspace) {
      if (out) CS_HIP(hipMemcpy(out, comp_label.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
      return ncomp;
    }
    // cell space: every NODATA row is a component of its own in comp_label; the caller sees the components of the real
    // graph, numbered by their smallest node id (the order of the surviving labels is already that order)
    if (ncomp_api < 0) {
      DBuf keep = dalloc<int>((size_t)ncomp + 1), total = dalloc<int>(1);
      CS_HIP(hipMemsetAsync(keep.p, 0, keep.bytes, st));
      hipLaunchKernelGGL(comp_keep_kernel, dim3(grid_for(n_api)), dim3(256), 0, st, n_api, (const int*)dptr<int>(node2cell),
                         (const int*)dptr<int>(comp_label), dptr<int>(keep));
      exclusive_scan_i32(dptr<int>(keep), ncomp + 1, st, dptr<int>(total));
      ncomp_api = read_int(dptr<int>(total), st);
      comp_label_api.alloc((size_t)n_api * sizeof(int));
      hipLaunchKernelGGL(comp_compact_kernel, dim3(grid_for(n_api)), dim3(256), 0, st, n_api, (const int*)dptr<int>(node2cell),
                         (const int*)dptr<int>(comp_label), (const int*)dptr<int>(keep), dptr<int>(comp_label_api));
      check_launch("components (cell space)");
      CS_HIP(hipStreamSynchronize(st));
    }
    if (out) CS_HIP(hipMemcpy(out, comp_label_api.p, (size_t)n_api * sizeof(int), hipMemcpyDeviceToHost));
    return ncomp_api;
  }

  // Advanced-mode solve on a raster-built handle, rasters in and out (compute_omniscape_current, utils.jl:145-257, for
  // one raster or for many windows stacked into one raster with NODATA separators -- every window is a component of
  // ONE block-diagonal system, solved by ONE PCG). curr_out / volt_out: row-major rasters, 0 where there is no node.
  void solve_raster(const void* source, void* curr_out, void* volt_out, csgpu_stats* stats) override {
    if (poly_proj) return poly_fallback().solve_raster(source, curr_out, volt_out, stats);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    CS_REQUIRE(nodemap.p != nullptr, CSGPU_BAD_ARGS, "handle was not built by csgpu_raster_setup");
    const int64_t ncells = raster_rows * raster_cols;
    ensure_csr();
    ensure_components();
    W.ensure(n, 1, H.levels.size() > 1 && H.levels[0].two_product() ? H.levels[1].A.nrows : 0);
    if (stats) {
      stats->nrhs = 1;
      stats->batch = 1;
    }
    DBuf dsrc((size_t)ncells * sizeof(T)), has = dalloc<int>((size_t)2 * ncomp);
    CS_HIP(hipMemcpyAsync(dsrc.p, source, (size_t)ncells * sizeof(T), hipMemcpyHostToDevice, st));
    CS_HIP(hipMemsetAsync(has.p, 0, (size_t)2 * ncomp * sizeof(int), st));
    CS_HIP(hipMemsetAsync(W.rhs(), 0, (size_t)n * sizeof(T), st));
    const T* gnode = ground_node.p ? (const T*)dptr<T>(ground_node) : (const T*)nullptr;
    hipLaunchKernelGGL((raster_rhs_kernel<T>), dim3(grid_for(ncells)), dim3(256), 0, st, ncells,
                       raster_rowmap(), (const T*)dptr<T>(dsrc), gnode, (const int*)dptr<int>(comp_label),
                       W.rhs(), dptr<int>(has));
    hipLaunchKernelGGL((raster_rhs_mask_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                       (const int*)dptr<int>(comp_label), (const int*)dptr<int>(has), W.rhs());
    // every component weighs alike in the one stopping rule: exact power-of-two normalisation per component
    DBuf absmax = dalloc<unsigned long long>((size_t)ncomp);
    CS_HIP(hipMemsetAsync(absmax.p, 0, absmax.bytes, st));
    hipLaunchKernelGGL((comp_absmax_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                       (const int*)dptr<int>(comp_label), (const T*)W.rhs(), dptr<unsigned long long>(absmax));
    hipLaunchKernelGGL((comp_scale_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                       (const int*)dptr<int>(comp_label), (const unsigned long long*)dptr<unsigned long long>(absmax), 1,
                       W.rhs());
    PcgParams pp = pcg_params(1);
    pp.need_x = true;
    pp.comp_label = dptr<int>(comp_label);
    pp.ncomp = (int)ncomp;
    PcgBatchResult r = pcg_solve<T, TP, 1>(cg_matrix(), H, W, pp, 1, st, dia_ptr());
    accumulate(stats, r, 1);
    hipLaunchKernelGGL((comp_scale_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                       (const int*)dptr<int>(comp_label), (const unsigned long long*)dptr<unsigned long long>(absmax), -1,
                       dptr<T>(W.x));
    DBuf draster((size_t)ncells * sizeof(T));
    if (volt_out) {
      hipLaunchKernelGGL((raster_scatter_kernel<T>), dim3(grid_for(ncells)), dim3(256), 0, st, ncells,
                         raster_rowmap(), (const T*)dptr<T>(W.x), dptr<T>(draster));
      CS_HIP(hipMemcpyAsync(volt_out, draster.p, (size_t)ncells * sizeof(T), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
    }
    if (curr_out) {
      const Csr<T>& A = cg_matrix();
      const int gc = grid_for(n);
      // the drop threshold of every component refers to that component's own largest branch current
      DBuf dcurr((size_t)n * sizeof(T)), cmax = dalloc<unsigned long long>((size_t)2 * ncomp);
      hipLaunchKernelGGL(compmax_init_kernel, dim3(grid_for(2 * ncomp)), dim3(256), 0, st, (int64_t)2 * ncomp,
                         dptr<unsigned long long>(cmax));
      hipLaunchKernelGGL((branch_max_comp_kernel<T>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(), A.va(),
                         (const T*)dptr<T>(W.x), (const int*)dptr<int>(comp_label), dptr<unsigned long long>(cmax));
      hipLaunchKernelGGL((node_current_kernel<T, 1>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(), A.va(),
                         (const T*)dptr<T>(W.x), (const double*)nullptr, dptr<T>(dcurr), gnode,
                         (const int*)dptr<int>(comp_label), (const unsigned long long*)dptr<unsigned long long>(cmax));
      hipLaunchKernelGGL((raster_scatter_kernel<T>), dim3(grid_for(ncells)), dim3(256), 0, st, ncells,
                         raster_rowmap(), (const T*)dptr<T>(dcurr), dptr<T>(draster));
      CS_HIP(hipMemcpyAsync(curr_out, draster.p, (size_t)ncells * sizeof(T), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
    }
    check_launch("solve_raster");
    if (stats) stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  int pick_k(int64_t ncols) const {
    int kmax = opts.batch;
    if (kmax < 1) kmax = 1;
    if (kmax > kMaxK) kmax = kMaxK;
    // Batches of 32 pay where the wide passes run in the marching kernels: handles whose level 0 is in lattice form. The
    // general CSR SpMM of the coarser levels processes the 32 columns of an fp64 hierarchy as two halves of 16 (spmv.h);
    // the level-0 kernels of a CSR-path handle (networks, rasters with thin polygons: spmv_launch_cg / _wide, the long-row
    // kernel) have no such form and run 1.75x slower per column with 8 rows per lane, so fp64 hierarchies without a
    // lattice level 0 stay at 16 columns. fp32 hierarchies (8 lanes per row at K = 32) are fine. CSGPU_WIDE_CSR=1 lifts it.
    if (kmax > 16 && sizeof(TP) == 8 && !kn.wide_csr) {
      if (!(H.levels.size() > 1 && H.levels[0].lattice_two_product())) kmax = 16;
    }
    int k = 1;
    while (k < kmax && k < ncols) k <<= 1;
    return k;
  }

  template <int K>
  PcgBatchResult run_batch(int ncols, bool need_x, const double* bb_host = nullptr, const int* pair_src = nullptr,
                           const int* pair_dst = nullptr) {
    PcgParams pp = pcg_params(K);
    pp.need_x = need_x;
    pp.rhs_in_r = !need_x;  // (only solve_pairs runs without the whole solution; it hands the pairs over instead of b:
    pp.rp_ready = false;    //  pcg_solve writes -- or synthesises -- the +-1 entries of r0, and ||b||^2 = 2 is known)
    pp.bb_host = need_x ? nullptr : bb_host;
    if (!need_x) {
      pp.pair_src = pair_src;
      pp.pair_dst = pair_dst;
      pp.pair_cols = ncols;
    }
    return pcg_solve<T, TP, K>(cg_matrix(), H, W, pp, ncols, st, dia_ptr());
  }
  PcgBatchResult run_batch_k(int K, int ncols, bool need_x = true, const double* bb_host = nullptr,
                             const int* pair_src = nullptr, const int* pair_dst = nullptr) {
    switch (K) {
      case 1: return run_batch<1>(ncols, need_x, bb_host, pair_src, pair_dst);
      case 2: return run_batch<2>(ncols, need_x, bb_host, pair_src, pair_dst);
      case 4: return run_batch<4>(ncols, need_x, bb_host, pair_src, pair_dst);
      case 8: return run_batch<8>(ncols, need_x, bb_host, pair_src, pair_dst);
      case 32: return run_batch<32>(ncols, need_x, bb_host, pair_src, pair_dst);
      default: return run_batch<16>(ncols, need_x, bb_host, pair_src, pair_dst);
    }
  }

  // streaming pair solves (pcg_stream_pairs, pcg.h) for the wide batches
  PcgStreamResult run_stream_k(int K, const int64_t* src, const int64_t* dst, int64_t npairs, const int64_t* gather,
                               int64_t ngather, T* resist_out, T* gathered_out) {
    PcgParams pp = pcg_params(K);
    switch (K) {
      case 8: return pcg_stream_pairs<T, TP, 8>(H, W, pp, dia, src, dst, npairs, gather, ngather, resist_out, gathered_out, st);
      case 16: return pcg_stream_pairs<T, TP, 16>(H, W, pp, dia, src, dst, npairs, gather, ngather, resist_out, gathered_out, st);
      case 32: return pcg_stream_pairs<T, TP, 32>(H, W, pp, dia, src, dst, npairs, gather, ngather, resist_out, gathered_out, st);
      default: return PcgStreamResult();
    }
  }

  void accumulate(csgpu_stats* s, const PcgBatchResult& r, int ncols) {
    if (!s) return;
    for (int c = 0; c < ncols; ++c) {
      s->total_iters += r.s.iters[c];
      s->max_iters = std::max(s->max_iters, r.s.iters[c]);
      s->max_relres = std::max(s->max_relres, r.s.relres[c]);
      // the reference's ONLY acceptance test is the residual (src/core.jl:639-641: Krylov.cg's stats are not looked at): a
      // column that stopped on itmax or on a breakdown of the recurrence with ||Ax-b||/||b|| < 1e-4 is a success there, and
      // here (fuzz finding of round 5: ten-decade mazes at rtol 1e-10 stagnate at 1e-7 and were reported as failures)
      // (... with two decades of margin where the figure is the recurrence residual of a column that did not stop on the
      // rule: column_accepted, pcg.h)
      const bool bad = !column_accepted(r.s.relres[c], r.s.done[c], r.explicit_relres);
      if (bad) s->not_converged += 1;
    }
    s->device_ms += r.device_ms;
    s->cg_spmv_ms += r.spmv_ms;
    s->cg_spmv_calls += r.spmv_calls;
    s->graph_launches += r.graph_launches;
    s->polished_batches += r.polished;
    s->cg_spmv_bytes = r.spmv_bytes;
    s->resid_ms += r.resid_ms;
    s->resid_calls += r.resid_calls;
    s->resid_bytes = r.resid_bytes;
    s->resid_fused = r.resid_fused;
  }

#define CS_DISPATCH_K(K, ...)                               \
  switch (K) {                                              \
    case 1: { constexpr int KK = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int KK = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int KK = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int KK = 8; __VA_ARGS__; } break;   \
    case 32: { constexpr int KK = 32; __VA_ARGS__; } break; \
    default: { constexpr int KK = 16; __VA_ARGS__; } break; \
  }

  void solve_pairs(const int64_t* src, const int64_t* dst, int64_t npairs, void* volt_out, const int64_t* gather,
                   int64_t ngather, void* gathered_out, void* resist_out, csgpu_stats* stats, const int32_t* weights,
                   void* curr_out, void* cum_inout, void* max_inout, void* branch_out) override {
    // polygon handle of the lattice path: resistances (and gathered focal voltages) here, everything else -- voltage /
    // current maps, explicit checks of whole solutions -- on the merged graph
    if (poly_proj && (volt_out || curr_out || cum_inout || max_inout || branch_out || opts.explicit_check > 0))
      return poly_fallback().solve_pairs(src, dst, npairs, volt_out, gather, ngather, gathered_out, resist_out, stats, weights,
                                         curr_out, cum_inout, max_inout, branch_out);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int64_t p = 0; p < npairs; ++p)
      CS_REQUIRE(src[p] >= 0 && src[p] < n_api && dst[p] >= 0 && dst[p] < n_api, CSGPU_BAD_ARGS, "pair node id out of range");
    for (int64_t g = 0; g < ngather; ++g)
      CS_REQUIRE(gather[g] >= 0 && gather[g] < n_api, CSGPU_BAD_ARGS, "gather node id out of range");
    CS_REQUIRE(!(cellspace && branch_out), CSGPU_BAD_ARGS,
               "branch currents are indexed by the stored entries of a network's matrix (out.jl:209-290); this handle was "
               "built from a raster");
    // from here on: row ids of the device matrix (cell space: the cells of the nodes)
    const Ids src_r = rows_of(src, npairs), dst_r = rows_of(dst, npairs), gather_r = rows_of(gather, ngather);
    src = src_r.p;
    dst = dst_r.p;
    gather = gather_r.p;
    if (ncomp > 1) {
      // a pair across two components is an inconsistent singular system (the reference only pairs points of one
      // component, core.jl:146-153): refuse it instead of iterating to itmax
      DBuf ids = dalloc<int64_t>((size_t)2 * npairs), labs = dalloc<int>((size_t)2 * npairs);
      CS_HIP(hipMemcpyAsync(ids.p, src, (size_t)npairs * sizeof(int64_t), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(dptr<int64_t>(ids) + npairs, dst, (size_t)npairs * sizeof(int64_t), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(gather_int_kernel, dim3(grid_for(2 * npairs)), dim3(256), 0, st, 2 * npairs,
                         (const int64_t*)dptr<int64_t>(ids), (const int*)dptr<int>(comp_label), dptr<int>(labs));
      std::vector<int> lab((size_t)2 * npairs);
      CS_HIP(hipMemcpyAsync(lab.data(), labs.p, lab.size() * sizeof(int), hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      for (int64_t p = 0; p < npairs; ++p)
        CS_REQUIRE(lab[p] == lab[npairs + p], CSGPU_BAD_ARGS, "pair spans two connected components");
    }
    // Kmax: the width of the call's full batches; the work arena is sized for it. The batch width is then picked PER BATCH:
    // the ragged tail of a pair list runs at the width ITS column count asks for (100 pairs at batch 32 = 32 + 32 + 32 + 4:
    // the last four pairs run through the K = 4 kernels in the same buffers, instead of dragging 28 idle columns through
    // every K = 32 pass -- 0.6 s of the 2.66 s job at 10000^2; the reference's batched driver handles a
    // short last batch too, src/core.jl:448-463).
    const int Kmax = pick_k(npairs);
    const int64_t tail_rows = H.levels.size() > 1 && H.levels[0].two_product() ? H.levels[1].A.nrows : 0;
    W.ensure(n, Kmax, tail_rows);
    if (stats) {
      stats->nrhs = (int)npairs;
      stats->batch = Kmax;
    }
    // Streaming ("continuous batching", pcg.h): a resistance-only call with more pairs than columns on the lattice path
    // can keep every column busy -- a column takes the next pair of the list when its own has converged -- instead of
    // solving batch after batch at the pace of each batch's slowest column. A pair then costs its own iterations + 1
    // K-wide iterations (the + 1 is its initial V-cycle, during which the column's CG product and residual update idle)
    // against the batch's slowest column + an initial V-cycle: a gain when the iteration counts of a batch are spread
    // (NODATA / heterogeneous rasters: 13.9 mean against 16 at 10000^2 with 15 % holes), a small loss when they are not
    // (the bench raster: 10.76 against 11). The first batch of a call therefore always runs as a batch and its iteration
    // counts decide for the rest of the list (re-evaluated after every further batch). Used where an iteration takes
    // milliseconds (the host looks at the slots after every iteration): CSGPU_STREAM_MIN (vector elements n * K, default
    // 2^25) moves that bound; CSGPU_STREAM=1 streams from the first pair on (tests, A/B), CSGPU_NO_STREAM=1 never.
    bool stream_eligible = false, stream_now = false;
    if (!(volt_out || curr_out || cum_inout || max_inout || branch_out || opts.explicit_check > 0) && dia_ptr() && npairs > Kmax &&
        Kmax >= 8 && !poly_proj && kn.stream >= 0) {
      stream_eligible = (int64_t)n * Kmax >= kn.stream_min;
      stream_now = stream_eligible && kn.stream > 0;
    }
    // the rest of the list [p0, npairs) as a stream; false when the stream declined (the batches go on)
    auto stream_rest = [&](int64_t p0) -> bool {
      PcgStreamResult sr = run_stream_k(Kmax, src + p0, dst + p0, npairs - p0, gather, ngather,
                                        resist_out ? (T*)resist_out + p0 : (T*)nullptr,
                                        gathered_out ? (T*)gathered_out + (size_t)p0 * ngather : (T*)nullptr);
      if (!sr.applicable) return false;
      if (stats) {
        for (size_t p = 0; p < sr.iters.size(); ++p) {
          stats->total_iters += sr.iters[p];
          stats->max_iters = std::max(stats->max_iters, sr.iters[p]);
          stats->max_relres = std::max(stats->max_relres, sr.relres[p]);
          if (!column_accepted(sr.relres[p], sr.status[p], false)) stats->not_converged += 1;   // (as in accumulate())
        }
        stats->device_ms += sr.device_ms;
        stats->cg_spmv_ms += sr.spmv_ms;
        stats->cg_spmv_calls += sr.spmv_calls;
        stats->cg_spmv_bytes = sr.spmv_bytes;
        stats->polished_batches += sr.polished;
        stats->stream_slots += sr.slots;
      }
      return true;
    };
    DBuf dsrc = dalloc<int>(Kmax), ddst = dalloc<int>(Kmax);
    DBuf dgather = dalloc<int>((size_t)std::max<int64_t>(ngather, 1));
    if (ngather > 0) {
      std::vector<int> g32(ngather);
      for (int64_t g = 0; g < ngather; ++g) g32[g] = (int)gather[g];
      CS_HIP(hipMemcpyAsync(dgather.p, g32.data(), (size_t)ngather * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipStreamSynchronize(st));
    }
    DBuf dres = dalloc<T>(Kmax), dgath = dalloc<T>((size_t)std::max<int64_t>(ngather, 1) * Kmax);
    DBuf dvolt;
    if (volt_out || curr_out) dvolt.alloc((size_t)n * Kmax * sizeof(T));
    // N1: node currents of every pair, optional cumulative / maximum accumulation over the pairs of this call
    const bool want_curr = curr_out || cum_inout || max_inout || branch_out;
    // resistance-only calls consume x at the pairs' nodes and the gathered focal nodes only (core.jl:231-232,
    // 685-703): accumulate just those entries instead of carrying the n x K solution through every iteration
    const bool need_x = volt_out || want_curr || opts.explicit_check > 0;
    if (need_x) ensure_csr();  // (the explicit residual check and the current kernels walk the CSR form)
    std::vector<int> focal;
    if (!need_x) focal.resize((size_t)ngather + 2 * Kmax);
    DBuf dcurr, dcum, dmax, dweight, dbpart, dbmax, dbranch, dbranch2;
    if (branch_out) {
      dbranch.alloc((size_t)std::max<int64_t>(nnz, 1) * Kmax * sizeof(T));
      dbranch2.alloc((size_t)std::max<int64_t>(nnz, 1) * Kmax * sizeof(T));
    }
    if (want_curr) {
      dcurr.alloc((size_t)n * Kmax * sizeof(T));
      dweight.alloc((size_t)Kmax * sizeof(int));
      dbpart.alloc((size_t)kMaxGrid * Kmax * 2 * sizeof(double));
      dbmax.alloc((size_t)Kmax * 2 * sizeof(double));
      if (cum_inout) {
        dcum.alloc((size_t)n * sizeof(T));
        CS_HIP(hipMemsetAsync(dcum.p, 0, dcum.bytes, st));
      }
      if (max_inout) {
        dmax.alloc((size_t)n * sizeof(T));
        CS_HIP(hipMemsetAsync(dmax.p, 0, dmax.bytes, st));
      }
    }
    std::vector<int> s32(Kmax), d32(Kmax), w32(Kmax);
    const bool fixed_k = kn.fixed_k;  // A/B knob (csgpu_opts.fixed_k): every batch at the call's width (round 5)
    int K = Kmax;
    for (int64_t p0 = 0; p0 < npairs; p0 += K) {
      if (stream_now && npairs - p0 > Kmax) {
        if (stream_rest(p0)) break;
        stream_now = stream_eligible = false;  // (declined: not asked again in this call)
      }
      K = fixed_k ? Kmax : pick_k(npairs - p0);
      W.ensure(n, K, tail_rows);  // (a narrower batch: same buffers, laid out for its own width)
      const int ncols = (int)std::min<int64_t>(K, npairs - p0);
      for (int c = 0; c < K; ++c) {
        s32[c] = (int)src[p0 + std::min(c, ncols - 1)];
        d32[c] = (int)dst[p0 + std::min(c, ncols - 1)];
      }
      CS_HIP(hipMemcpyAsync(dsrc.p, s32.data(), K * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(ddst.p, d32.data(), K * sizeof(int), hipMemcpyHostToDevice, st));
      // right-hand side: into b, or -- focal path -- the pairs themselves go down (r0 = b; nothing reads b later: pcg_solve
      // writes r0, or on the fused lattice path never stores it)
      if (need_x) {
        T* rhs = W.rhs();
        CS_HIP(hipMemsetAsync(rhs, 0, (size_t)n * K * sizeof(T), st));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((pairs_rhs_kernel<T, KK>), dim3(1), dim3(64), 0, st, rhs, dptr<int>(dsrc),
                                             dptr<int>(ddst), ncols));
      }
      double bb[kMaxK];
      for (int c = 0; c < kMaxK; ++c) bb[c] = (c < K && s32[c] != d32[c]) ? 2.0 : 0.0;
      // (polygon handles: norms are taken in NODE space -- the merged system's ||b_m||^2 = 2 whatever the polygons' sizes; a
      // unit current into a polygon's node is spread evenly over its cells by the first projection; pcg.h, poly.h)
      if (!need_x) {  // focal list: [gathered nodes..., src of every column..., dst of every column...]
        focal.resize((size_t)ngather + 2 * K);
        for (int64_t g = 0; g < ngather; ++g) focal[g] = (int)gather[g];
        for (int c = 0; c < K; ++c) {
          focal[ngather + c] = s32[c];
          focal[ngather + K + c] = d32[c];
        }
        W.set_focal(focal, st);
      }
      PcgBatchResult r = run_batch_k(K, ncols, need_x, bb, dptr<int>(dsrc), dptr<int>(ddst));
      accumulate(stats, r, ncols);
      if (stream_eligible && ncols == K) {
        // spread of this batch's iteration counts: stream the rest when (mean + 1) slots per pair beat (max + 1/2)
        double sum = 0;
        int mx = 0;
        for (int c = 0; c < K; ++c) {
          sum += r.s.iters[c];
          mx = std::max(mx, r.s.iters[c]);
        }
        // (measured on MI355X, profiles/r4_stream_*.jsonl: at 10000^2 with 15 % NODATA a batch's slowest column is only
        // ~0.5 iterations above the batch's mean -- the 13.9 against 16 of round 3 compared the mean with the maximum over
        // ALL batches -- and the stream's 91 slots for 96 pairs buy nothing against 6 batches; on a sigma = 3 raster, 82
        // against ~90 per batch, streaming is 4 % faster)
        stream_now = (sum / K + 1.0) * 1.04 < mx + 0.5;
      }
      const int ge = grid_for((int64_t)ncols * (ngather + 1));
      if (need_x) {
        CS_DISPATCH_K(K, hipLaunchKernelGGL((pairs_extract_kernel<T, KK>), dim3(ge), dim3(256), 0, st,
                                             (const T*)dptr<T>(W.x), dptr<int>(dsrc), dptr<int>(ddst), ncols,
                                             dptr<int>(dgather), (int)ngather, dptr<T>(dres), dptr<T>(dgath)));
      } else {
        CS_DISPATCH_K(K, hipLaunchKernelGGL((pairs_extract_focal_kernel<T, KK>), dim3(ge), dim3(256), 0, st,
                                             (const T*)dptr<T>(W.xf), ncols, (int)ngather, dptr<T>(dres),
                                             dptr<T>(dgath)));
      }
      if (resist_out)
        CS_HIP(hipMemcpyAsync((T*)resist_out + p0, dres.p, (size_t)ncols * sizeof(T), hipMemcpyDeviceToHost, st));
      if (gathered_out && ngather > 0)
        CS_HIP(hipMemcpyAsync((T*)gathered_out + (size_t)p0 * ngather, dgath.p, (size_t)ncols * ngather * sizeof(T),
                              hipMemcpyDeviceToHost, st));
      if (volt_out) {
        CS_DISPATCH_K(K, hipLaunchKernelGGL((pairs_volt_kernel<T, KK>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n,
                                             (const T*)dptr<T>(W.x), dptr<int>(dsrc), ncols, dptr<T>(dvolt)));
        download_cols((const T*)dptr<T>(dvolt), ncols, (T*)volt_out + (size_t)p0 * n_api);
      }
      if (want_curr) {
        const Csr<T>& A = cg_matrix();
        const int gc = grid_for(n * K);
        CS_DISPATCH_K(K, hipLaunchKernelGGL((branch_max_kernel<T, KK>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(),
                                            A.va(), (const T*)dptr<T>(W.x), dptr<double>(dbpart)));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((branch_max_final_kernel<KK>), dim3(1), dim3(256), 0, st,
                                            (const double*)dptr<double>(dbpart), gc, dptr<double>(dbmax)));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((node_current_kernel<T, KK>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(),
                                            A.va(), (const T*)dptr<T>(W.x), (const double*)dptr<double>(dbmax),
                                            dptr<T>(dcurr), (const T*)nullptr, (const int*)nullptr,
                                            (const unsigned long long*)nullptr));
        if (branch_out) {
          CS_DISPATCH_K(K, hipLaunchKernelGGL((branch_current_kernel<T, KK>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(),
                                              A.va(), (const T*)dptr<T>(W.x), (const double*)dptr<double>(dbmax),
                                              dptr<T>(dbranch)));
          CS_DISPATCH_K(K, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(nnz * ncols)), dim3(256), 0, st, nnz,
                                              (const T*)dptr<T>(dbranch), ncols, dptr<T>(dbranch2)));
          CS_HIP(hipMemcpyAsync((T*)branch_out + (size_t)p0 * nnz, dbranch2.p, (size_t)nnz * ncols * sizeof(T),
                                hipMemcpyDeviceToHost, st));
        }
        if (curr_out) {
          CS_DISPATCH_K(K, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n,
                                              (const T*)dptr<T>(dcurr), ncols, dptr<T>(dvolt)));
          download_cols((const T*)dptr<T>(dvolt), ncols, (T*)curr_out + (size_t)p0 * n_api);
        }
        if (cum_inout || max_inout) {
          for (int c = 0; c < K; ++c) w32[c] = c < ncols ? (weights ? weights[p0 + c] : 1) : 0;
          CS_HIP(hipMemcpyAsync(dweight.p, w32.data(), K * sizeof(int), hipMemcpyHostToDevice, st));
          CS_DISPATCH_K(K, hipLaunchKernelGGL((current_accumulate_kernel<T, KK>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                                              (const T*)dptr<T>(dcurr), ncols, (const int*)dptr<int>(dweight),
                                              cum_inout ? dptr<T>(dcum) : (T*)nullptr, max_inout ? dptr<T>(dmax) : (T*)nullptr));
        }
      }
      check_launch("solve_pairs batch");
      CS_HIP(hipStreamSynchronize(st));
    }
    if (cum_inout || max_inout) {
      std::vector<T> tmp((size_t)n_api);
      if (cum_inout) {
        download_cols((const T*)dptr<T>(dcum), 1, tmp.data());
        T* h = (T*)cum_inout;
        for (int64_t i = 0; i < n_api; ++i) h[i] += tmp[i];
      }
      if (max_inout) {
        download_cols((const T*)dptr<T>(dmax), 1, tmp.data());
        T* h = (T*)max_inout;
        for (int64_t i = 0; i < n_api; ++i) h[i] = tmp[i] > h[i] ? tmp[i] : h[i];
      }
    }
    if (stats) stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  void solve_rhs(const void* rhs, int64_t nrhs, void* x_out, csgpu_stats* stats) override {
    if (poly_proj) return poly_fallback().solve_rhs(rhs, nrhs, x_out, stats);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    const int K = pick_k(nrhs);
    ensure_csr();
    W.ensure(n, K, H.levels.size() > 1 && H.levels[0].two_product() ? H.levels[1].A.nrows : 0);
    if (stats) {
      stats->nrhs = (int)nrhs;
      stats->batch = K;
    }
    DBuf stage((size_t)n * K * sizeof(T));
    for (int64_t p0 = 0; p0 < nrhs; p0 += K) {
      const int ncols = (int)std::min<int64_t>(K, nrhs - p0);
      upload_cols((const T*)rhs + (size_t)p0 * n_api, ncols, dptr<T>(stage));
      CS_DISPATCH_K(K, hipLaunchKernelGGL((interleave_kernel<T, KK>), dim3(grid_for(n * K)), dim3(256), 0, st, n,
                                           (const T*)dptr<T>(stage), ncols, W.rhs()));
      PcgBatchResult r = run_batch_k(K, ncols);
      accumulate(stats, r, ncols);
      CS_DISPATCH_K(K, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n,
                                           (const T*)dptr<T>(W.x), ncols, dptr<T>(stage)));
      check_launch("solve_rhs batch");
      download_cols((const T*)dptr<T>(stage), ncols, (T*)x_out + (size_t)p0 * n_api);
    }
    if (stats) stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  // N2: per-column Dirichlet sets on one hierarchy (see csgpu_solve_grounded / csgpu_solve_sources in csgpu.h)
  void solve_grounded(const GroundedJob& J, int64_t nrhs, csgpu_stats* stats) override {
    if (poly_proj) return poly_fallback().solve_grounded(J, nrhs, stats);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    const int64_t* gptr = J.gptr;
    const int64_t* gidx = J.gidx;
    const int64_t* sptr = J.sptr;
    const int64_t* sidx = J.sidx;
    const bool sparse = J.rhs == nullptr;
    for (int64_t c = 0; c < nrhs; ++c) {
      CS_REQUIRE(gptr[c] <= gptr[c + 1], CSGPU_BAD_ARGS, "ground_ptr must be non-decreasing");
      for (int64_t e = gptr[c]; e < gptr[c + 1]; ++e)
        CS_REQUIRE(gidx[e] >= 0 && gidx[e] < n_api, CSGPU_BAD_ARGS, "ground node id out of range");
      if (sparse) {
        CS_REQUIRE(sptr[c] <= sptr[c + 1], CSGPU_BAD_ARGS, "source_ptr must be non-decreasing");
        for (int64_t e = sptr[c]; e < sptr[c + 1]; ++e)
          CS_REQUIRE(sidx[e] >= 0 && sidx[e] < n_api, CSGPU_BAD_ARGS, "source node id out of range");
      }
      if (J.check) CS_REQUIRE(J.check[c] < n_api, CSGPU_BAD_ARGS, "check node id out of range");
    }
    const Ids gidx_r = rows_of(gidx + gptr[0], gptr[nrhs] - gptr[0]);  // row ids of the device matrix (cell space)
    gidx = gidx_r.p - gptr[0];
    Ids sidx_r;
    if (sparse) {
      sidx_r = rows_of(sidx + sptr[0], sptr[nrhs] - sptr[0]);
      sidx = sidx_r.p - sptr[0];
    }
    std::vector<int64_t> chk_in;  // (a negative id = no check node: mapped as node 0, masked again afterwards)
    Ids chk_r;
    if (J.check) {
      chk_in.resize((size_t)nrhs);
      for (int64_t c = 0; c < nrhs; ++c) chk_in[(size_t)c] = std::max<int64_t>(J.check[c], 0);
      chk_r = rows_of(chk_in.data(), nrhs);
    }
    const int Kmax = pick_k(nrhs);  // (the arena's width; the ragged last batch runs at its own, as in solve_pairs)
    const bool want_curr = J.curr_out || J.cum_inout || J.max_inout;
    ensure_csr();
    const int64_t tail_rows = H.levels.size() > 1 && H.levels[0].two_product() ? H.levels[1].A.nrows : 0;
    W.ensure(n, Kmax, tail_rows);
    W.drop_graphs();  // captured chunks hold the ground-set buffers of an earlier call
    if (stats) {
      stats->nrhs = (int)nrhs;
      stats->batch = Kmax;
    }
    int64_t maxg = 1, maxs = 1;
    for (int64_t p0 = 0; p0 < nrhs; p0 += Kmax) {
      maxg = std::max(maxg, gptr[std::min(nrhs, p0 + Kmax)] - gptr[p0]);
      if (sparse) maxs = std::max(maxs, sptr[std::min(nrhs, p0 + Kmax)] - sptr[p0]);
    }
    DBuf stage, dgp = dalloc<int>(Kmax + 1), dgi = dalloc<int>((size_t)maxg);
    if (!sparse || J.x_out || J.curr_out) stage.alloc((size_t)n * Kmax * sizeof(T));
    DBuf dsrow, dscol, dsval, dchk, dchkv;
    if (sparse) {
      dsrow = dalloc<int>((size_t)maxs);
      dscol = dalloc<int>((size_t)maxs);
      dsval = dalloc<T>((size_t)maxs);
    }
    if (J.check && J.check_out) {
      dchk = dalloc<int>(Kmax);
      dchkv = dalloc<T>(Kmax);
    }
    DBuf dcurr, dbpart, dbmax, dcum, dmax, dweight;
    if (want_curr) {
      dcurr.alloc((size_t)n * Kmax * sizeof(T));
      dbpart.alloc((size_t)kMaxGrid * Kmax * 2 * sizeof(double));
      dbmax.alloc((size_t)Kmax * 2 * sizeof(double));
      if (J.cum_inout || J.max_inout) dweight.alloc((size_t)Kmax * sizeof(int));
      if (J.cum_inout) {
        dcum.alloc((size_t)n * sizeof(T));
        CS_HIP(hipMemsetAsync(dcum.p, 0, dcum.bytes, st));
      }
      if (J.max_inout) {
        dmax.alloc((size_t)n * sizeof(T));
        CS_HIP(hipMemsetAsync(dmax.p, 0, dmax.bytes, st));
      }
    }
    std::vector<int> hp(Kmax + 1), hi, srow, scol, w32(Kmax), c32(Kmax);
    std::vector<T> sval, chkv(Kmax);
    struct Ent {
      int col, row;
      T val;
    };
    std::vector<Ent> ents;
    const bool fixed_k = kn.fixed_k;  // A/B knob, as in solve_pairs
    int K = Kmax;
    for (int64_t p0 = 0; p0 < nrhs; p0 += K) {
      K = fixed_k ? Kmax : pick_k(nrhs - p0);
      if (K != W.K) {
        W.ensure(n, K, tail_rows);
        W.drop_graphs();
      }
      const int ncols = (int)std::min<int64_t>(K, nrhs - p0);
      hi.clear();
      for (int c = 0; c < K; ++c) {
        hp[c] = (int)hi.size();
        if (c < ncols)
          for (int64_t e = gptr[p0 + c]; e < gptr[p0 + c + 1]; ++e) hi.push_back((int)gidx[e]);
      }
      hp[K] = (int)hi.size();
      CS_HIP(hipMemcpyAsync(dgp.p, hp.data(), (size_t)(K + 1) * sizeof(int), hipMemcpyHostToDevice, st));
      if (!hi.empty()) CS_HIP(hipMemcpyAsync(dgi.p, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice, st));
      if (!sparse) {
        upload_cols((const T*)J.rhs + (size_t)p0 * n_api, ncols, dptr<T>(stage));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((interleave_kernel<T, KK>), dim3(grid_for(n * K)), dim3(256), 0, st, n,
                                             (const T*)dptr<T>(stage), ncols, W.rhs()));
      } else {
        // the batch's entries, duplicates of one (column, node) summed on the host in the caller's order (a few entries per
        // column: a one-to-all column is a single +1 and does not cross PCIe as 8 n bytes of zeros)
        ents.clear();
        for (int c = 0; c < ncols; ++c)
          for (int64_t e = sptr[p0 + c]; e < sptr[p0 + c + 1]; ++e)
            ents.push_back(Ent{c, (int)sidx[e], J.sval ? ((const T*)J.sval)[e] : T(1)});
        std::stable_sort(ents.begin(), ents.end(),
                         [](const Ent& a, const Ent& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
        srow.clear();
        scol.clear();
        sval.clear();
        for (const Ent& e : ents) {
          if (!srow.empty() && scol.back() == e.col && srow.back() == e.row) {
            sval.back() += e.val;
          } else {
            srow.push_back(e.row);
            scol.push_back(e.col);
            sval.push_back(e.val);
          }
        }
        CS_HIP(hipMemsetAsync(W.rhs(), 0, (size_t)n * K * sizeof(T), st));
        if (!srow.empty()) {
          CS_HIP(hipMemcpyAsync(dsrow.p, srow.data(), srow.size() * sizeof(int), hipMemcpyHostToDevice, st));
          CS_HIP(hipMemcpyAsync(dscol.p, scol.data(), scol.size() * sizeof(int), hipMemcpyHostToDevice, st));
          CS_HIP(hipMemcpyAsync(dsval.p, sval.data(), sval.size() * sizeof(T), hipMemcpyHostToDevice, st));
          CS_DISPATCH_K(K, hipLaunchKernelGGL((sparse_rhs_kernel<T, KK>), dim3(ceil_div((int)srow.size(), 256)), dim3(256), 0,
                                               st, (int)srow.size(), (const int*)dptr<int>(dsrow),
                                               (const int*)dptr<int>(dscol), (const T*)dptr<T>(dsval), W.rhs()));
        }
      }
      const int gtotal = (int)hi.size();
      if (gtotal > 0)
        CS_DISPATCH_K(K, hipLaunchKernelGGL((mask_grounds_kernel<T, T, KK>), dim3(ceil_div(gtotal, 256)), dim3(256), 0, st,
                                             (const int*)dptr<int>(dgp), (const int*)dptr<int>(dgi), W.rhs(),
                                             (T*)nullptr, (const int*)nullptr));
      CS_HIP(hipStreamSynchronize(st));  // hp / hi / the entry lists are reused by the next batch
      PcgBatchResult r;
      {
        PcgParams pp = pcg_params(K);
        pp.need_x = true;
        pp.gptr = dptr<int>(dgp);
        pp.gidx = dptr<int>(dgi);
        pp.gtotal = gtotal;
        CS_DISPATCH_K(K, r = (pcg_solve<T, TP, KK>(cg_matrix(), H, W, pp, ncols, st, dia_ptr())));
      }
      accumulate(stats, r, ncols);
      if (J.check && J.check_out) {
        for (int c = 0; c < K; ++c) c32[c] = (c < ncols && J.check[p0 + c] >= 0) ? (int)chk_r[p0 + c] : -1;
        CS_HIP(hipMemcpyAsync(dchk.p, c32.data(), (size_t)K * sizeof(int), hipMemcpyHostToDevice, st));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((gather_columns_kernel<T, KK>), dim3(1), dim3(64), 0, st,
                                             (const T*)dptr<T>(W.x), (const int*)dptr<int>(dchk), ncols, dptr<T>(dchkv)));
        CS_HIP(hipMemcpyAsync(chkv.data(), dchkv.p, (size_t)K * sizeof(T), hipMemcpyDeviceToHost, st));
        CS_HIP(hipStreamSynchronize(st));
        for (int c = 0; c < ncols; ++c) ((T*)J.check_out)[p0 + c] = chkv[c];
      }
      if (J.x_out) {
        CS_DISPATCH_K(K, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n,
                                             (const T*)dptr<T>(W.x), ncols, dptr<T>(stage)));
        download_cols((const T*)dptr<T>(stage), ncols, (T*)J.x_out + (size_t)p0 * n_api);
      }
      if (want_curr) {
        const Csr<T>& A = cg_matrix();
        const int gc = grid_for(n * K);
        CS_DISPATCH_K(K, hipLaunchKernelGGL((branch_max_kernel<T, KK>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(),
                                            A.va(), (const T*)dptr<T>(W.x), dptr<double>(dbpart)));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((branch_max_final_kernel<KK>), dim3(1), dim3(256), 0, st,
                                            (const double*)dptr<double>(dbpart), gc, dptr<double>(dbmax)));
        CS_DISPATCH_K(K, hipLaunchKernelGGL((node_current_kernel<T, KK>), dim3(gc), dim3(256), 0, st, (int)n, A.rp(), A.ci(),
                                            A.va(), (const T*)dptr<T>(W.x), (const double*)dptr<double>(dbmax),
                                            dptr<T>(dcurr), (const T*)nullptr, (const int*)nullptr,
                                            (const unsigned long long*)nullptr));
        if (J.curr_out) {
          CS_HIP(hipStreamSynchronize(st));  // stage still feeds the copy of x
          CS_DISPATCH_K(K, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n * ncols)), dim3(256), 0, st, n,
                                              (const T*)dptr<T>(dcurr), ncols, dptr<T>(stage)));
          download_cols((const T*)dptr<T>(stage), ncols, (T*)J.curr_out + (size_t)p0 * n_api);
        }
        if (J.cum_inout || J.max_inout) {
          // the reference's serial merge after the fan-out over the focal points (src/raster/onetoall.jl:153-158)
          for (int c = 0; c < K; ++c) w32[c] = c < ncols ? 1 : 0;
          CS_HIP(hipMemcpyAsync(dweight.p, w32.data(), K * sizeof(int), hipMemcpyHostToDevice, st));
          CS_DISPATCH_K(K, hipLaunchKernelGGL((current_accumulate_kernel<T, KK>), dim3(grid_for(n)), dim3(256), 0, st, (int)n,
                                              (const T*)dptr<T>(dcurr), ncols, (const int*)dptr<int>(dweight),
                                              J.cum_inout ? dptr<T>(dcum) : (T*)nullptr,
                                              J.max_inout ? dptr<T>(dmax) : (T*)nullptr));
        }
      }
      check_launch("solve_grounded batch");
      CS_HIP(hipStreamSynchronize(st));
    }
    if (J.cum_inout || J.max_inout) {
      std::vector<T> tmp((size_t)n_api);
      if (J.cum_inout) {
        download_cols((const T*)dptr<T>(dcum), 1, tmp.data());
        T* h = (T*)J.cum_inout;
        for (int64_t i = 0; i < n_api; ++i) h[i] += tmp[i];
      }
      if (J.max_inout) {
        download_cols((const T*)dptr<T>(dmax), 1, tmp.data());
        T* h = (T*)J.max_inout;
        for (int64_t i = 0; i < n_api; ++i) h[i] = tmp[i] > h[i] ? tmp[i] : h[i];
      }
    }
    if (stats) stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  // Effective resistance between short-circuited node sets on ONE hierarchy (see csgpu_solve_region_pairs in csgpu.h):
  // column c solves A_ff y = A_fI 1 with the rows / columns of I u J masked; v = 1_I - y is the potential with I at 1 and
  // J at 0, and R = 1 / v'Av (the energy form: its error is second order in the error of the iterate).
  void solve_region_pairs(const int64_t* set_ptr, const int64_t* set_nodes, int64_t nsets, const int64_t* src_set,
                          const int64_t* dst_set, int64_t npairs, double* resistances, csgpu_stats* stats) override {
    if (poly_proj) return poly_fallback().solve_region_pairs(set_ptr, set_nodes, nsets, src_set, dst_set, npairs, resistances, stats);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    auto t0 = std::chrono::steady_clock::now();
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int64_t q = 0; q < nsets; ++q) {
      CS_REQUIRE(set_ptr[q] <= set_ptr[q + 1], CSGPU_BAD_ARGS, "set_ptr must be non-decreasing");
      for (int64_t e = set_ptr[q]; e < set_ptr[q + 1]; ++e)
        CS_REQUIRE(set_nodes[e] >= 0 && set_nodes[e] < n_api, CSGPU_BAD_ARGS, "set node id out of range");
    }
    const Ids set_r = rows_of(set_nodes + set_ptr[0], set_ptr[nsets] - set_ptr[0]);  // row ids of the device matrix
    set_nodes = set_r.p - set_ptr[0];
    for (int64_t p = 0; p < npairs; ++p)
      CS_REQUIRE(src_set[p] >= 0 && src_set[p] < nsets && dst_set[p] >= 0 && dst_set[p] < nsets && src_set[p] != dst_set[p],
                 CSGPU_BAD_ARGS, "pair refers to a set that does not exist (or to the same set twice)");
    // Two sets that share a node are one equipotential: R = 0 without a solve (the reference's bookkeeping does the same
    // for focal points on one node, core.jl:189,209-211). The remaining pairs are solved; `slot` maps them back.
    std::vector<int64_t> act_src, act_dst, slot;
    {
      std::vector<std::vector<int64_t>> sorted_sets((size_t)nsets);
      auto sorted = [&](int64_t q) -> const std::vector<int64_t>& {
        std::vector<int64_t>& v = sorted_sets[(size_t)q];
        if (v.empty() && set_ptr[q + 1] > set_ptr[q]) {
          v.assign(set_nodes + set_ptr[q], set_nodes + set_ptr[q + 1]);
          std::sort(v.begin(), v.end());
        }
        return v;
      };
      for (int64_t p = 0; p < npairs; ++p) {
        const std::vector<int64_t>&a = sorted(src_set[p]), &b = sorted(dst_set[p]);
        CS_REQUIRE(!a.empty() && !b.empty(), CSGPU_BAD_ARGS, "pair with an empty source or destination set");
        bool shared = false;
        for (size_t i = 0, j = 0; i < a.size() && j < b.size() && !shared;) {
          if (a[i] == b[j]) shared = true;
          else if (a[i] < b[j]) ++i;
          else ++j;
        }
        if (shared) {
          resistances[p] = 0.0;
        } else {
          act_src.push_back(src_set[p]);
          act_dst.push_back(dst_set[p]);
          slot.push_back(p);
        }
      }
    }
    if (stats) stats->nrhs = (int)npairs;
    if (slot.empty()) return;
    double* const res_all = resistances;
    src_set = act_src.data();
    dst_set = act_dst.data();
    npairs = (int64_t)slot.size();
    const int K = pick_k(npairs);
    ensure_csr();
    W.ensure(n, K, H.levels.size() > 1 && H.levels[0].two_product() ? H.levels[1].A.nrows : 0);
    W.drop_graphs();  // captured chunks hold the ground-set buffers of an earlier call
    if (stats) stats->batch = K;
    const Csr<T>& A = cg_matrix();
    DBuf volt((size_t)n * K * sizeof(T)), av((size_t)n * K * sizeof(T));
    // the set lists of the largest batch, allocated once: a captured PCG chunk bakes the list pointers in, so they must
    // not move between the batches of a call (a re-allocation here let a later batch replay a chunk that
    // masked the residual at the nodes of an earlier one)
    size_t maxg = 1, maxs = 1;
    for (int64_t p0 = 0; p0 < npairs; p0 += K) {
      size_t g = 0, s = 0;
      for (int64_t p = p0; p < std::min<int64_t>(npairs, p0 + K); ++p) {
        const size_t la = (size_t)(set_ptr[src_set[p] + 1] - set_ptr[src_set[p]]);
        const size_t lb = (size_t)(set_ptr[dst_set[p] + 1] - set_ptr[dst_set[p]]);
        g += la + lb;
        s += la;
      }
      maxg = std::max(maxg, g);
      maxs = std::max(maxs, s);
    }
    CS_REQUIRE(maxg < ((size_t)1 << 31), CSGPU_BAD_ARGS, "set lists of one batch exceed 2^31 entries");
    DBuf dgp = dalloc<int>(K + 1), dsp = dalloc<int>(K + 1), dgi = dalloc<int>(maxg), dsi = dalloc<int>(maxs);
    int sg = 1;
    CS_DISPATCH_K(K, sg = (spmv_grid<T, KK>((int)n)));
    DBuf part = dalloc<double>((size_t)sg * K);
    std::vector<double> hpart((size_t)sg * K);
    std::vector<int> gp(K + 1), sp(K + 1), gi, si;
    for (int64_t p0 = 0; p0 < npairs; p0 += K) {
      const int ncols = (int)std::min<int64_t>(K, npairs - p0);
      gi.clear();
      si.clear();
      for (int c = 0; c < K; ++c) {
        gp[c] = (int)gi.size();
        sp[c] = (int)si.size();
        if (c < ncols) {
          const int64_t a = src_set[p0 + c], b = dst_set[p0 + c];
          for (int64_t e = set_ptr[a]; e < set_ptr[a + 1]; ++e) {
            si.push_back((int)set_nodes[e]);
            gi.push_back((int)set_nodes[e]);
          }
          for (int64_t e = set_ptr[b]; e < set_ptr[b + 1]; ++e) gi.push_back((int)set_nodes[e]);
        }
      }
      gp[K] = (int)gi.size();
      sp[K] = (int)si.size();
      CS_HIP(hipMemcpyAsync(dgp.p, gp.data(), (size_t)(K + 1) * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(dsp.p, sp.data(), (size_t)(K + 1) * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(dgi.p, gi.data(), gi.size() * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemcpyAsync(dsi.p, si.data(), si.size() * sizeof(int), hipMemcpyHostToDevice, st));
      CS_HIP(hipMemsetAsync(volt.p, 0, volt.bytes, st));
      CS_DISPATCH_K(K, hipLaunchKernelGGL((scatter_ones_kernel<T, KK>), dim3(ceil_div((int64_t)si.size(), 256)), dim3(256), 0,
                                           st, (const int*)dptr<int>(dsp), (const int*)dptr<int>(dsi), dptr<T>(volt)));
      {  // b = A 1_I, masked on I u J
        SpmvArgs<T> a = spmv_args(A, (const T*)dptr<T>(volt), W.rhs());
        a.order = H.levels[0].orderA.p ? dptr<int>(H.levels[0].orderA) : nullptr;
        CS_DISPATCH_K(K, (spmv_launch<T, KK>(a, EPI_PLAIN, false, st)));
      }
      CS_DISPATCH_K(K, hipLaunchKernelGGL((mask_grounds_kernel<T, T, KK>), dim3(ceil_div((int64_t)gi.size(), 256)), dim3(256),
                                           0, st, (const int*)dptr<int>(dgp), (const int*)dptr<int>(dgi), W.rhs(),
                                           (T*)nullptr, (const int*)nullptr));
      CS_HIP(hipStreamSynchronize(st));  // the host lists are reused by the next batch
      PcgBatchResult r;
      {
        PcgParams pp = pcg_params(K);
        pp.need_x = true;
        pp.gptr = dptr<int>(dgp);
        pp.gidx = dptr<int>(dgi);
        pp.gtotal = (int)gi.size();
        CS_DISPATCH_K(K, r = (pcg_solve<T, TP, KK>(A, H, W, pp, ncols, st, dia_ptr())));
      }
      accumulate(stats, r, ncols);
      hipLaunchKernelGGL((subtract_kernel<T>), dim3(grid_for(n * K)), dim3(256), 0, st, (int64_t)n * K, dptr<T>(volt),
                         (const T*)dptr<T>(W.x));
      {  // v'Av per column: the product's fused dot with its own input
        SpmvArgs<T> a = spmv_args(A, (const T*)dptr<T>(volt), dptr<T>(av));
        a.order = H.levels[0].orderA.p ? dptr<int>(H.levels[0].orderA) : nullptr;
        a.dotw = nullptr;
        a.partials = dptr<double>(part);
        CS_DISPATCH_K(K, (spmv_launch<T, KK>(a, EPI_PLAIN, true, st)));
      }
      CS_HIP(hipMemcpyAsync(hpart.data(), part.p, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      check_launch("solve_region_pairs batch");
      CS_HIP(hipStreamSynchronize(st));
      for (int c = 0; c < ncols; ++c) {
        double e = 0;
        for (int b = 0; b < sg; ++b) e += hpart[(size_t)b * K + c];
        res_all[slot[(size_t)(p0 + c)]] = e > 0 ? 1.0 / e : -1.0;
      }
    }
    if (stats) stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  template <class U>
  static int64_t spmv_bytes(const Csr<U>& A, int k) {
    return A.nnz * (int64_t)(sizeof(U) + 4) + ((int64_t)A.nrows + 1) * 4 +
           ((int64_t)A.nrows + (int64_t)A.ncols) * k * (int64_t)sizeof(U);
  }

  void get_info(csgpu_info* info) const override {
    KnobScope ks(&kn);
    memset(info, 0, sizeof(*info));
    info->n = n_api;      // (cell space: the caller's node count; level_n[0] / level_nnz[0] are the device matrix's)
    info->nnz = nnz_api;
    info->levels = (int)H.levels.size();
    info->val_bytes = (int)sizeof(T);
    info->precond_bytes = (int)sizeof(TP);
    info->lattice_period = dia.n > 0 ? dia.R : 0;
    double nnz_sum = 0, n_sum = 0;
    int64_t bytes = (int64_t)(Aouter.device_bytes() + dia.device_bytes() + W.p2.bytes);
    const int tail_first = tail_first_level_peek(H);
    info->hierarchy_rebuilt_fp64 = rebuilt_fp64 ? 1 : 0;
    info->enrich_vectors = H.enr.nvec;
    info->host_blocks = host_blocks;
    info->batch_width = pick_k(opts.batch);
    info->stream_mode = kn.stream;
    info->tail_first_level = tail_first;
    {
      const bool sweeps_level = !H.coarse_dense;  // the last level runs Jacobi sweeps instead of a dense pseudo-inverse
      const int want = kn.last_level_sweeps;
      info->last_level_sweeps = !sweeps_level ? -1 : (want == 0 ? (H.levels.size() == 1 ? kSingleLevelSweeps : 8) : std::max(want, 0));
    }
    info->coarse_chebyshev = (H.levels.size() > 1 && !H.levels[1].weights.empty()) ? 1 : 0;
    info->cellspace = cellspace ? 1 : 0;
    info->poly_lattice = poly_proj ? 1 : 0;
    info->enrich_on = kn.enrich ? 1 : 0;
    info->expander_probe_hit = H.expander_probe_hit ? 1 : 0;
    info->fused_restrict_solves = H.fused_restrict_solves;
    info->virtual_rhs_solves = H.virtual_rhs_solves;
    info->reserved_info3 = 0;
    info->enrich_tau = kn.enrich ? kn.enrich_tau : 0.0;
    for (size_t l = 0; l < H.levels.size(); ++l) {
      const Level<TP>& L = H.levels[l];
      if (l < 32) {
        int form = CSGPU_FORM_CSR;
        if (l == 0) form = (dia.n > 0 && L.lattice_two_product()) ? CSGPU_FORM_LATTICE9 : CSGPU_FORM_CSR;
        else if (tail_first >= 1 && (int)l >= tail_first) form = CSGPU_FORM_TAIL;
        else if (L.lattice_v22()) form = CSGPU_FORM_LATTICE9;
        else if (L.A25.n > 0 && L.A25.n == L.A.nrows && pick_k(opts.batch) >= 8) form = CSGPU_FORM_LATTICE25;  // (vcycle: K >= 8)
        info->level_form[l] = form;
      }
      const int64_t lnnz = l == 0 ? nnz : L.A.nnz;  // (level 0 may hold no CSR form: lattice pipeline)
      nnz_sum += (double)lnnz;
      n_sum += (double)L.A.nrows;
      if (l < 32) {
        info->level_n[l] = L.A.nrows;
        info->level_nnz[l] = lnnz;
      }
      bytes += (int64_t)(L.A.device_bytes() + L.P.device_bytes() + L.R.device_bytes() + L.Q.device_bytes() + L.dinv.bytes +
                         L.xa.bytes + L.rb.bytes + L.b.bytes + L.qs.bytes + L.orderA.bytes + L.QT.device_bytes() +
                         L.M.device_bytes() + L.orderQT.bytes + L.Sdia.device_bytes() + L.Ql.device_bytes() + L.Adia.device_bytes() + L.A25.device_bytes());
    }
    bytes += (int64_t)H.enr.device_bytes();
    bytes += (int64_t)(H.coarse_inv.bytes + W.x.bytes + W.r.bytes + W.z.bytes + W.rp.bytes + W.p.bytes + W.Ap.bytes + W.b.bytes);
    info->operator_complexity = nnz_sum / std::max(1.0, (double)nnz);
    info->grid_complexity = n_sum / std::max(1.0, (double)H.levels[0].A.nrows);
    info->setup_ms = H.setup_ms;
    info->upload_ms = upload_ms;
    info->device_bytes = bytes;
    const int64_t csr_bytes_k1 = nnz * (int64_t)(sizeof(T) + 4) + (n + 1) * 4 + 2 * n * (int64_t)sizeof(T);
    info->spmv_bytes_fine = csr_bytes_k1;
    // B_iter = B_spmv(A0) + 10 n sizeof(T) + sum_l [(nu1+nu2+1) B_spmv(A_l) + B_spmv(P_l) + B_spmv(R_l) + 4 n_l sizeof(TP)]
    // for the CSR path; on a lattice level 0 the four marching kernels of DESIGN.md 4a (batch width 1):
    //   CG product n(5T + 3P), residual update n(5T + 2T + 2P), restriction n(9P + P) + n_c P, second product
    //   n(5P + 9P + 2P) + n_c P          (T, P = sizeof of the CG / preconditioner precision)
    const int64_t sT = (int64_t)sizeof(T), sP = (int64_t)sizeof(TP);
    const bool lat = dia.n > 0 && !H.levels.empty() && H.levels[0].lattice_two_product();
    int64_t bi = csr_bytes_k1 + 10 * n * sT;
    if (lat) {
      const int64_t nc = H.levels.size() > 1 ? H.levels[1].A.nrows : 0;
      bi = n * (5 * sT + 3 * sP) + n * (7 * sT + 2 * sP) + (n * 10 * sP + nc * sP) + (n * 16 * sP + nc * sP);
    }
    for (size_t l = lat ? 1 : 0; l + 1 < H.levels.size(); ++l) {
      const Level<TP>& L = H.levels[l];
      // sweeps: nu_pre / nu_post on level 0, nu_coarse on level 1, nu_coarse + 1 below; the first pre-sweep from a zero
      // guess needs no product, the first post-sweep is fused with the prolongation into one product with Q
      const int nuc = opts.nu_coarse > 0 ? opts.nu_coarse : 1;
      const int nup = l == 0 ? opts.nu_pre : (l == 1 ? nuc : nuc + 1), nuq = l == 0 ? opts.nu_post : (l == 1 ? nuc : nuc + 1);
      const int prods = std::max(nup - 1, 0) + 1 + std::max(nuq - 1, 0);
      bi += prods * spmv_bytes(L.A, 1) + spmv_bytes(L.Q, 1) + spmv_bytes(L.R, 1) + 4 * (int64_t)L.A.nrows * sP;
    }
    info->bytes_per_iteration = bi;
  }

  double spmv_bench(int k, int reps) override {
    if (poly_proj) return poly_fallback().spmv_bench(k, reps);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    ensure_csr();
    const Csr<T>& A = cg_matrix();
    DBuf x((size_t)n * k * sizeof(T)), y((size_t)n * k * sizeof(T));
    fill<T>(dptr<T>(x), n * k, T(1), st);
    hipEvent_t e0, e1;
    CS_HIP(hipEventCreate(&e0));
    CS_HIP(hipEventCreate(&e1));
    auto launch = [&]() {
      SpmvArgs<T> a = spmv_args(A, (const T*)dptr<T>(x), dptr<T>(y));
      a.order = H.levels[0].orderA.p ? dptr<int>(H.levels[0].orderA) : nullptr;
      CS_DISPATCH_K(k, spmv_launch<T, KK>(a, EPI_PLAIN, false, st));
    };
    launch();  // warm-up
    CS_HIP(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch();
    CS_HIP(hipEventRecord(e1, st));
    CS_HIP(hipEventSynchronize(e1));
    check_launch("spmv bench");
    float ms = 0;
    CS_HIP(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return (double)ms / std::max(reps, 1);
  }

  void spmv_host(const void* xh, void* yh, int k) override {
    if (poly_proj) return poly_fallback().spmv_host(xh, yh, k);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    ensure_csr();
    const Csr<T>& A = cg_matrix();
    DBuf x((size_t)n * k * sizeof(T)), y((size_t)n * k * sizeof(T));
    if (cellspace) {
      // host vectors are interleaved [n_api][k] in the caller's numbering: (node, column) pairs are "columns" of length k
      // for the translation kernels when the roles of the two indices are swapped -- simplest: go through column-major
      DBuf xc((size_t)n * k * sizeof(T)), t((size_t)n_api * k * sizeof(T)), t2((size_t)n_api * k * sizeof(T));
      CS_HIP(hipMemcpyAsync(t.p, xh, t.bytes, hipMemcpyHostToDevice, st));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n_api * k)), dim3(256), 0, st, n_api,
                                           (const T*)dptr<T>(t), k, dptr<T>(t2)));
      hipLaunchKernelGGL((cells_from_nodes_kernel<T>), dim3(grid_for(n * k)), dim3(256), 0, st, n, n_api,
                         (const int*)dptr<int>(cell2node), (const T*)dptr<T>(t2), k, dptr<T>(xc));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((interleave_kernel<T, KK>), dim3(grid_for(n * k)), dim3(256), 0, st, n,
                                           (const T*)dptr<T>(xc), k, dptr<T>(x)));
      SpmvArgs<T> a = spmv_args(A, (const T*)dptr<T>(x), dptr<T>(y));
      a.order = H.levels[0].orderA.p ? dptr<int>(H.levels[0].orderA) : nullptr;
      CS_DISPATCH_K(k, spmv_launch<T, KK>(a, EPI_PLAIN, false, st));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((deinterleave_kernel<T, KK>), dim3(grid_for(n * k)), dim3(256), 0, st, n,
                                           (const T*)dptr<T>(y), k, dptr<T>(xc)));
      hipLaunchKernelGGL((nodes_from_cells_kernel<T>), dim3(grid_for(n_api * k)), dim3(256), 0, st, n, n_api,
                         (const int*)dptr<int>(node2cell), (const T*)dptr<T>(xc), k, dptr<T>(t2));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((interleave_kernel<T, KK>), dim3(grid_for(n_api * k)), dim3(256), 0, st, n_api,
                                           (const T*)dptr<T>(t2), k, dptr<T>(t)));
      check_launch("spmv_host (cell space)");
      CS_HIP(hipMemcpyAsync(yh, t.p, t.bytes, hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      return;
    }
    CS_HIP(hipMemcpyAsync(x.p, xh, x.bytes, hipMemcpyHostToDevice, st));
    SpmvArgs<T> a = spmv_args(A, (const T*)dptr<T>(x), dptr<T>(y));
    a.order = H.levels[0].orderA.p ? dptr<int>(H.levels[0].orderA) : nullptr;
    CS_DISPATCH_K(k, spmv_launch<T, KK>(a, EPI_PLAIN, false, st));
    check_launch("spmv_host");
    CS_HIP(hipMemcpyAsync(yh, y.p, y.bytes, hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
  }

  // y = (level matrix) x through the launcher the V-cycle uses for that operator (test hook). Host arrays in the
  // hierarchy's precision; for which == 5 ([S Q]) the fused dot x[0:n] . y is returned too.
  // level whose CSR operators the test hooks read: H.levels[lvl], or -- level 0 of a lattice-pipeline handle -- level 0
  // of a hierarchy built by the CSR pipeline from the CSR form of the same matrix (see Href)
  Level<TP>& hook_level(int lvl) {
    Level<TP>& L = H.levels[lvl];
    if (lvl != 0 || csr_ready_at_setup()) return L;
    if (!Href) {
      ensure_csr();
      Csr<TP> Ap;
      if constexpr (MIXED) {
        convert_csr(Aouter, Ap, st);
      } else {
        const Csr<T>& A = H.levels[0].A;  // copy: the hook hierarchy owns its matrix
        convert_csr(A, Ap, st);
      }
      DBuf drow((size_t)n * sizeof(int)), dcol((size_t)n * sizeof(int)), w;
      hipLaunchKernelGGL(lattice_coords_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, dia.R, dptr<int>(drow), dptr<int>(dcol));
      SetupParams sp = setup_params();
      sp.grid_rows = dia.R;
      sp.grid_cols = (int)(n / dia.R);
      sp.max_levels = 2;
      sp.lattice_s = false;
      if (cellspace) {
        w.alloc((size_t)n * sizeof(long long));
        hipLaunchKernelGGL(weights_from_map_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, (const int*)dptr<int>(cell2node),
                           dptr<long long>(w));
        sp.size0 = dptr<long long>(w);
        sp.n_real = n_api;
      }
      Href.reset(new Hierarchy<TP>());
      amg_setup(*Href, std::move(Ap), sp, dptr<int>(drow), dptr<int>(dcol), st);
      CS_HIP(hipStreamSynchronize(st));
    }
    return Href->levels[0];
  }
  bool csr_ready_at_setup() const { return H.levels[0].Q.nnz > 0 || !H.levels[0].lattice_two_product() || H.levels.size() < 2; }

  void level_spmv_host(int lvl, int which, const void* xh, void* yh, int k, double* dots) override {
    if (which == 6) {
      // test hook of the polygon lattice path (poly.h): y = Pi x for a cell-space vector x ([R*C][k], column-major cell ids)
      // and dots[c] = ||Pi x||^2 in NODE space -- the merged system's norm -- from the very kernels the PCG loop uses
      CS_REQUIRE(poly_proj && proj.nchunks > 0, CSGPU_BAD_ARGS, "not a polygon handle on the lattice path");
      std::lock_guard<std::mutex> lk(mu);
      KnobScope ks(&kn);
      CS_HIP(hipSetDevice(device));
      DBuf x((size_t)n * k * sizeof(TP));
      const int gv = grid_for(n * k);
      DBuf part = dalloc<double>((size_t)(gv + 1) * kMaxK);
      CS_HIP(hipMemcpyAsync(x.p, xh, x.bytes, hipMemcpyHostToDevice, st));
      CS_DISPATCH_K(k, (poly_project<TP, TP, KK>(proj, dptr<TP>(x), (TP*)nullptr, (const int*)nullptr, st)));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((dot_kernel<TP, KK, false>), dim3(gv), dim3(256), 0, st, n, (const TP*)dptr<TP>(x),
                                          (const TP*)dptr<TP>(x), dptr<double>(part), (const TP*)nullptr, (const TP*)nullptr,
                                          (double*)nullptr));
      CS_DISPATCH_K(k, hipLaunchKernelGGL((poly_norm_corr_kernel<KK>), dim3(1), dim3(256), 0, st, proj,
                                          dptr<double>(part) + (size_t)gv * KK, (const int*)nullptr));
      check_launch("polygon projection hook");
      std::vector<double> hp((size_t)(gv + 1) * k);
      CS_HIP(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
      CS_HIP(hipMemcpyAsync(yh, x.p, x.bytes, hipMemcpyDeviceToHost, st));
      CS_HIP(hipStreamSynchronize(st));
      if (dots)
        for (int c = 0; c < k; ++c) {
          double t = 0.0;
          for (int r = 0; r <= gv; ++r) t += hp[(size_t)r * k + c];
          dots[c] = t;
        }
      return;
    }
    if (poly_proj) return poly_fallback().level_spmv_host(lvl, which, xh, yh, k, dots);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    CS_REQUIRE(lvl >= 0 && lvl < (int)H.levels.size(), CSGPU_BAD_ARGS, "level out of range");
    Level<TP>& L = H.levels[lvl];   // the forms the solve phase uses (lattice kernels)
    Level<TP>& Lc = hook_level(lvl);  // the CSR operators
    // lattice level: the CSR forms of Q^T / [S Q] are not kept; build them for the size query only
    if (which == 5 && Lc.M.nnz == 0 && L.lattice_two_product()) build_sq_matrix(Lc, st);
    if (which == 4 && Lc.QT.nnz == 0 && L.lattice_two_product()) build_qt_matrix(Lc, st);
    const Csr<TP>& M = which == 0 ? Lc.A : which == 1 ? Lc.P : which == 2 ? Lc.R : which == 3 ? Lc.Q : which == 4 ? Lc.QT : Lc.M;
    CS_REQUIRE(M.nnz > 0, CSGPU_BAD_ARGS, "level has no such operator");
    DBuf x((size_t)M.ncols * k * sizeof(TP)), y((size_t)M.nrows * k * sizeof(TP));
    DBuf part = dalloc<double>(std::max<size_t>(16384, spmv_grid_upper(M.nrows)) * kMaxK);
    CS_HIP(hipMemcpyAsync(x.p, xh, x.bytes, hipMemcpyHostToDevice, st));
    SpmvArgs<TP> a = spmv_args(M, (const TP*)dptr<TP>(x), dptr<TP>(y));
    const bool sq = which == 5;
    if (which == 0 || sq) a.order = Lc.orderA.p ? dptr<int>(Lc.orderA) : nullptr;
    if (which == 4) a.order_lr = Lc.orderQT.p ? dptr<int>(Lc.orderQT) : nullptr;
    if (sq) a.partials = dptr<double>(part);
    const bool sq_lattice = sq && L.lattice_two_product();
    if (sq_lattice) {  // the kernel the V-cycle runs on a lattice level: S b + Q x_c with x = [b; x_c]
      CS_DISPATCH_K(k, dia_sq_product<TP, KK>(L.Sdia, L.Ql, (const TP*)dptr<TP>(x), (const TP*)dptr<TP>(x) + (size_t)M.nrows * k,
                                             dptr<TP>(y), dptr<double>(part), nullptr, st));
    } else if (which == 4 && L.lattice_two_product()) {  // index-free restriction
      CS_DISPATCH_K(k, lattice_restrict<TP, KK>(L.Ql, (const TP*)dptr<TP>(x), dptr<TP>(y), nullptr, st));
    } else if (which == 0 && L.A25.n == M.nrows && k >= 8) {  // refined-tile lattice level: the 25-point marching product
      CS_DISPATCH_K(k, dia25_launch<TP, (KK >= 8 ? KK : 8)>(L.A25, D25_PLAIN, (const TP*)dptr<TP>(x), dptr<TP>(y), (const TP*)nullptr,
                                                           (const TP*)nullptr, TP(0), nullptr, st));
    } else if (sq) {
      CS_DISPATCH_K(k, spmv_launch_wide<TP, KK>(a, true, st));
    } else {
      CS_DISPATCH_K(k, spmv_launch<TP, KK>(a, EPI_PLAIN, false, st));
    }
    check_launch("level_spmv_host");
    CS_HIP(hipMemcpyAsync(yh, y.p, y.bytes, hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
    if (sq && dots) {
      int g = 1;
      CS_DISPATCH_K(k, g = sq_lattice ? dia_grid<TP, TP, KK>(L.Sdia)
                             : spmv_grid<TP, KK>(M.nrows));
      std::vector<double> ph((size_t)g * k);
      CS_HIP(hipMemcpy(ph.data(), part.p, ph.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int c = 0; c < k; ++c) {
        double s = 0;
        for (int b = 0; b < g; ++b) s += ph[(size_t)b * k + c];
        dots[c] = s;
      }
    }
  }

  void get_level_matrix(int lvl, int which, int64_t* nrows, int64_t* ncols, int64_t* nnz_out, int32_t* rowptr,
                        int32_t* colidx, void* vals) const override {
    KnobScope ks(&kn);
    if (poly_proj)
      return const_cast<Solver<T, TP>*>(this)->poly_fallback().get_level_matrix(lvl, which, nrows, ncols, nnz_out, rowptr, colidx, vals);
    CS_REQUIRE(lvl >= 0 && lvl < (int)H.levels.size(), CSGPU_BAD_ARGS, "level out of range");
    auto* self = const_cast<Solver<T, TP>*>(this);  // (hooks may build the CSR forms they inspect)
    if (lvl == 0 && which == 0) self->ensure_csr();
    const Level<TP>& L = (lvl == 0 && which != 0) ? self->hook_level(0) : H.levels[lvl];
    if (cellspace && lvl == 0 && which == 0) {
      // the matrix of the real graph in the reference's node numbering: the NODATA rows dropped, columns renumbered
      // (test / measurement hook: compacted on the host)
      const Csr<T>& A = cg_matrix();
      if (nrows) *nrows = n_api;
      if (ncols) *ncols = n_api;
      if (nnz_out) *nnz_out = nnz_api;
      if (!rowptr && !colidx && !vals) return;
      std::vector<int> rp((size_t)n + 1), ci((size_t)nnz), c2n((size_t)n);
      std::vector<T> va((size_t)nnz);
      CS_HIP(hipMemcpy(rp.data(), A.rp(), rp.size() * sizeof(int), hipMemcpyDeviceToHost));
      CS_HIP(hipMemcpy(ci.data(), A.ci(), ci.size() * sizeof(int), hipMemcpyDeviceToHost));
      CS_HIP(hipMemcpy(va.data(), A.va(), va.size() * sizeof(T), hipMemcpyDeviceToHost));
      CS_HIP(hipMemcpy(c2n.data(), cell2node.p, c2n.size() * sizeof(int), hipMemcpyDeviceToHost));
      int64_t o = 0;
      for (int64_t cell = 0; cell < n; ++cell) {
        const int nd = c2n[(size_t)cell] - 1;
        if (nd < 0) continue;
        if (rowptr) rowptr[nd] = (int32_t)o;
        for (int k = rp[(size_t)cell]; k < rp[(size_t)cell + 1]; ++k, ++o) {
          if (colidx) colidx[o] = c2n[(size_t)ci[(size_t)k]] - 1;
          if (vals) ((T*)vals)[o] = va[(size_t)k];
        }
      }
      if (rowptr) rowptr[n_api] = (int32_t)o;
      return;
    }
    if (lvl == 0 && which == 0) {  // the CG matrix itself, in the handle's value type
      const Csr<T>& A = cg_matrix();
      if (nrows) *nrows = A.nrows;
      if (ncols) *ncols = A.ncols;
      if (nnz_out) *nnz_out = A.nnz;
      if (rowptr && A.rowptr.p) CS_HIP(hipMemcpy(rowptr, A.rp(), (size_t)(A.nrows + 1) * sizeof(int), hipMemcpyDeviceToHost));
      if (colidx && A.nnz > 0) CS_HIP(hipMemcpy(colidx, A.ci(), (size_t)A.nnz * sizeof(int), hipMemcpyDeviceToHost));
      if (vals && A.nnz > 0) CS_HIP(hipMemcpy(vals, A.va(), (size_t)A.nnz * sizeof(T), hipMemcpyDeviceToHost));
      return;
    }
    if (H.levels[lvl].lattice_two_product() && ((which == 5 && L.M.nnz == 0) || (which == 4 && L.QT.nnz == 0))) {
      // lattice level: the CSR forms of Q^T / [S Q] are not kept; build them (independent CSR builders) for inspection
      if (which == 5) build_sq_matrix(const_cast<Level<TP>&>(L), st);
      if (which == 4) build_qt_matrix(const_cast<Level<TP>&>(L), st);
      CS_HIP(hipStreamSynchronize(st));
    }
    const Csr<TP>& M = which == 0 ? L.A : which == 1 ? L.P : which == 2 ? L.R : which == 3 ? L.Q : which == 4 ? L.QT : L.M;
    if (nrows) *nrows = M.nrows;
    if (ncols) *ncols = M.ncols;
    if (nnz_out) *nnz_out = M.nnz;
    if (rowptr && M.rowptr.p) CS_HIP(hipMemcpy(rowptr, M.rp(), (size_t)(M.nrows + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (colidx && M.nnz > 0) CS_HIP(hipMemcpy(colidx, M.ci(), (size_t)M.nnz * sizeof(int), hipMemcpyDeviceToHost));
    if (vals && M.nnz > 0) {
      std::vector<TP> tmp((size_t)M.nnz);
      CS_HIP(hipMemcpy(tmp.data(), M.va(), (size_t)M.nnz * sizeof(TP), hipMemcpyDeviceToHost));
      T* out = (T*)vals;
      for (size_t i = 0; i < tmp.size(); ++i) out[i] = (T)tmp[i];
    }
  }

  void dia_product_host(const void* zh, const void* ph, const double* beta, void* pout_h, void* yh, int k,
                        double* dots) override {
    if (poly_proj) return poly_fallback().dia_product_host(zh, ph, beta, pout_h, yh, k, dots);
    std::lock_guard<std::mutex> lk(mu);
    KnobScope ks(&kn);
    CS_HIP(hipSetDevice(device));
    CS_REQUIRE(dia.n > 0, CSGPU_BAD_ARGS, "matrix has no lattice form");
    const size_t xb = (size_t)n * k * sizeof(TP), yb = (size_t)n * k * sizeof(T);
    DBuf z(xb), pin(xb), pout(xb), y(yb), dbeta = dalloc<double>(kMaxK);
    DBuf part = dalloc<double>(std::max<size_t>(16384, spmv_grid_upper(n)) * kMaxK);
    CS_HIP(hipMemcpyAsync(z.p, zh, xb, hipMemcpyHostToDevice, st));
    CS_HIP(hipMemcpyAsync(pin.p, ph, xb, hipMemcpyHostToDevice, st));
    CS_HIP(hipMemsetAsync(dbeta.p, 0, kMaxK * sizeof(double), st));
    CS_HIP(hipMemcpyAsync(dbeta.p, beta, (size_t)k * sizeof(double), hipMemcpyHostToDevice, st));
    CS_HIP(hipMemsetAsync(pout.p, 0, xb, st));
    int g = 1;
    switch (k) {
      case 1: g = dia_grid<T, TP, 1>(dia); dia_cg_product<T, TP, 1>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
      case 2: g = dia_grid<T, TP, 2>(dia); dia_cg_product<T, TP, 2>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
      case 4: g = dia_grid<T, TP, 4>(dia); dia_cg_product<T, TP, 4>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
      case 8: g = dia_grid<T, TP, 8>(dia); dia_cg_product<T, TP, 8>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
      case 32: g = dia_grid<T, TP, 32>(dia); dia_cg_product<T, TP, 32>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
      default: g = dia_grid<T, TP, 16>(dia); dia_cg_product<T, TP, 16>(dia, nullptr, dptr<TP>(z), dptr<TP>(pin), dptr<TP>(pout), dptr<T>(y), dptr<double>(part), st, dptr<double>(dbeta)); break;
    }
    check_launch("dia_product_host");
    CS_HIP(hipMemcpyAsync(pout_h, pout.p, xb, hipMemcpyDeviceToHost, st));
    CS_HIP(hipMemcpyAsync(yh, y.p, yb, hipMemcpyDeviceToHost, st));
    std::vector<double> phost((size_t)g * k);
    CS_HIP(hipMemcpyAsync(phost.data(), part.p, phost.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    CS_HIP(hipStreamSynchronize(st));
    if (dots)
      for (int c = 0; c < k; ++c) {
        double s = 0;
        for (int b = 0; b < g; ++b) s += phost[(size_t)b * k + c];
        dots[c] = s;
      }
  }
};

}  // namespace csgpu

struct csgpu_handle {
  std::unique_ptr<csgpu::ISolver> solver;
};

using csgpu::Error;
using csgpu::g_last_error;

#define CS_API_BEGIN try {
#define CS_API_END                                   \
  }                                                  \
  catch (const Error& e) {                           \
    g_last_error = e.what();                         \
    return e.code;                                   \
  }                                                  \
  catch (const std::bad_alloc&) {                    \
    g_last_error = "host allocation failed";         \
    return CSGPU_OOM;                                \
  }                                                  \
  catch (const std::exception& e) {                  \
    g_last_error = e.what();                         \
    return CSGPU_INTERNAL;                           \
  }

extern "C" {

const char* csgpu_version(void) { return "csgpu 0.2 (gfx950)"; }
const char* csgpu_last_error(void) { return g_last_error.c_str(); }

int csgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void csgpu_default_opts(csgpu_opts* o) {
  memset(o, 0, sizeof(*o));
  o->struct_size = (int32_t)sizeof(csgpu_opts);
  o->device = -1;
  o->max_levels = 16;
  o->max_coarse = 100;
  o->aggregation = CSGPU_AGG_AUTO;
  o->nu_pre = 1;
  o->nu_post = 1;
  o->criterion = CSGPU_CRIT_KRYLOV;
  o->itmax = 100000;
  o->batch = 8;
  o->check_every = 0;
  o->nu_coarse = 2;
  o->theta = 0.0;
  o->omega_p = 1.6;
  o->omega_s = 1.7;
  o->rtol = 1e-6;
  o->atol = -1.0;
  o->node_row = nullptr;
  o->node_col = nullptr;
  o->precond_bytes = 0;
  o->use_graph = 0;
  o->two_product = 0;
  o->stencil = 0;
  o->explicit_check = 0;
  o->reserved3 = 0;
}

// host_matrix: csgpu_setup / csgpu_multi_setup -- 2^31 stored entries and more are streamed (setup_from_host_streamed),
// which decides for itself whether the matrix qualifies
static int check_common(int64_t n, int64_t nnz, int val_bytes, const csgpu_opts* opts, bool host_matrix = false) {
  if (n <= 0 || nnz < 0 || (val_bytes != 4 && val_bytes != 8)) {
    g_last_error = "bad arguments: n, nnz or val_bytes";
    return CSGPU_BAD_ARGS;
  }
  if ((nnz >= ((int64_t)1 << 31) && !host_matrix) || n >= ((int64_t)1 << 31) - 1) {
    g_last_error = "matrix too large for int32 device indexing (need nnz < 2^31 and n < 2^31 - 1)";
    return CSGPU_BAD_ARGS;
  }
  if (opts && (opts->precond_bytes != 0 && opts->precond_bytes != 4 && opts->precond_bytes != 8)) {
    g_last_error = "csgpu_opts.precond_bytes must be 0, 4 or 8";
    return CSGPU_BAD_ARGS;
  }
  if (opts && opts->struct_size != (int32_t)sizeof(csgpu_opts)) {
    g_last_error = "csgpu_opts.struct_size mismatch (call csgpu_default_opts first)";
    return CSGPU_BAD_ARGS;
  }
  return CSGPU_OK;
}

// A problem that will not be coarsened (n <= max_coarse: the preconditioner IS the dense pseudo-inverse) gains nothing from
// an fp32 preconditioner and loses a lot: the pseudo-inverse's cutoff n eps(fp32) lambda_max sits inside the spectrum of a
// heterogeneous component, and sqrt(r'z) of an fp32 z is noise once r is small (fuzz findings: a 75-node path with
// conductances over three decades did not converge, a 94-node graph stopped at ||Ax-b||/||b|| = 3e-7 for rtol = 1e-10).
// Such handles compute in the matrix precision whatever precond_bytes says.
static void single_level_precision(csgpu_opts& o, int64_t n) {
  if (n <= (int64_t)o.max_coarse) o.precond_bytes = 0;
}

// fp32 hierarchies on strongly heterogeneous rasters: conductance ratios of e^+-9 eat the fp32
// mantissa of the Galerkin operators -- a coarse row whose entries are 1e4 and whose couplings to the next region are
// 1e-4 loses those couplings to the rounding of the large ones -- and the fp32 hierarchy of a log-normal sigma = 3 raster
// needs 135 iterations where the fp64 one needs 82 (3000^2; 460 ms vs 418 ms per batch, so the fp64 hierarchy is also
// the faster one). At sigma = 2 both need 25 and fp32 is 1.5x faster. The strength test of level 0 measures which regime
// a raster is in: the fraction of cells the 0.03 sqrt(a_ii a_jj) filter moves out of their 3x3 tile is 0.0 % at
// sigma = 1, 0.3 % at 1.5, 1.7 % at 2, 4.4 % at 2.5, 7.8 % at 3. Above CSGPU_HETERO_FP64_FRAC (default 3 %) a handle
// asked for with precond_bytes = 4 is rebuilt with an fp64 hierarchy (csgpu_get_info then reports the precision in effect);
// CSGPU_HETERO_FP64_FRAC=1 switches the fallback off.
static bool hetero_wants_fp64(const csgpu::ISolver& s) {
  const double lim = s.kn.hetero_fp64_frac;
  const bool yes = s.hetero_frac() > lim;
  if (yes && s.kn.verbose)
    fprintf(stderr, "csgpu: heterogeneous raster (%.1f %% of the cells leave their tile): fp64 hierarchy instead of fp32\n",
            100.0 * s.hetero_frac());
  return yes;
}

// Raster entry points: whether the handle is coarsened depends on the number of NODES, not of cells (a raster
// with few valid cells stays in compact numbering with n = valid cells). Upper bound of the node count -- valid cells
// outside polygons + polygons holding a valid cell -- with an early exit once it exceeds max_coarse, so the scan of the
// host raster costs nothing on real landscapes.
static void single_level_precision_raster(csgpu_opts& o, const void* cond_, const int32_t* poly, int64_t cells) {
  typedef double T;  // (only fp64 matrices can carry an fp32 preconditioner)
  if (o.precond_bytes == 0) return;
  const T* cond = (const T*)cond_;
  const int64_t limit = (int64_t)o.max_coarse;
  int64_t nodes = 0;
  std::vector<int32_t> seen;  // polygon ids counted so far (at most `limit` + 1 of them)
  for (int64_t i = 0; i < cells && nodes <= limit; ++i) {
    if (!(cond[i] > T(0))) continue;
    if (poly && poly[i] > 0) {
      if (std::find(seen.begin(), seen.end(), poly[i]) != seen.end()) continue;
      seen.push_back(poly[i]);
    }
    ++nodes;
  }
  single_level_precision(o, nodes);
}

int csgpu_setup(const void* rowptr, const void* colidx, const void* vals, int64_t n, int64_t nnz, int idx_bytes,
                int val_bytes, int index_base, const csgpu_opts* opts, csgpu_handle** out) {
  CS_API_BEGIN
  if (!rowptr || !colidx || !vals || !out || (idx_bytes != 4 && idx_bytes != 8) || (index_base != 0 && index_base != 1)) {
    g_last_error = "bad arguments: null pointer, idx_bytes or index_base";
    return CSGPU_BAD_ARGS;
  }
  int rc = check_common(n, nnz, val_bytes, opts, /*host_matrix=*/true);
  if (rc) return rc;
  csgpu_opts o;
  if (opts) o = *opts; else csgpu_default_opts(&o);
  single_level_precision(o, n);
  std::unique_ptr<csgpu_handle> h(new csgpu_handle());
  if (val_bytes == 8 && o.precond_bytes == 4) {
    auto* s = new csgpu::Solver<double, float>(o);
    h->solver.reset(s);
    s->setup_from_host(rowptr, colidx, vals, n, nnz, idx_bytes, index_base);
    if (hetero_wants_fp64(*s)) {
      h->solver.reset();
      o.precond_bytes = 0;
      auto* s2 = new csgpu::Solver<double, double>(o);
      h->solver.reset(s2);
      s2->rebuilt_fp64 = true;
      s2->setup_from_host(rowptr, colidx, vals, n, nnz, idx_bytes, index_base);
    }
  } else if (val_bytes == 8) {
    auto* s = new csgpu::Solver<double, double>(o);
    h->solver.reset(s);
    s->setup_from_host(rowptr, colidx, vals, n, nnz, idx_bytes, index_base);
  } else {
    auto* s = new csgpu::Solver<float, float>(o);
    h->solver.reset(s);
    s->setup_from_host(rowptr, colidx, vals, n, nnz, idx_bytes, index_base);
  }
  *out = h.release();
  return CSGPU_OK;
  CS_API_END
}

int csgpu_raster_setup_grounded(const void* cond, const void* ground, int64_t nrows, int64_t ncols, int val_bytes,
                                int four_neighbors, int avg_resistances, int reg, const csgpu_opts* opts,
                                csgpu_handle** out) {
  CS_API_BEGIN
  if (!cond || !out || nrows <= 0 || ncols <= 0) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  int rc = check_common(nrows * ncols, 0, val_bytes, opts);
  if (rc) return rc;
  // (rasters of 2^31 / 9 = 238 M cells and more have no int32 CSR form; the index-free pipeline takes them when it
  // applies -- checked inside)
  csgpu_opts o;
  if (opts) o = *opts; else csgpu_default_opts(&o);
  o.node_row = o.node_col = nullptr;
  if (val_bytes == 8) single_level_precision_raster(o, cond, nullptr, nrows * ncols);
  std::unique_ptr<csgpu_handle> h(new csgpu_handle());
  if (val_bytes == 8 && o.precond_bytes == 4) {
    auto* s = new csgpu::Solver<double, float>(o);
    h->solver.reset(s);
    s->setup_from_raster(cond, nrows, ncols, four_neighbors, avg_resistances, reg, ground);
    if (hetero_wants_fp64(*s)) {
      h->solver.reset();
      o.precond_bytes = 0;
      auto* s2 = new csgpu::Solver<double, double>(o);
      h->solver.reset(s2);
      s2->rebuilt_fp64 = true;
      s2->setup_from_raster(cond, nrows, ncols, four_neighbors, avg_resistances, reg, ground);
    }
  } else if (val_bytes == 8) {
    auto* s = new csgpu::Solver<double, double>(o);
    h->solver.reset(s);
    s->setup_from_raster(cond, nrows, ncols, four_neighbors, avg_resistances, reg, ground);
  } else {
    auto* s = new csgpu::Solver<float, float>(o);
    h->solver.reset(s);
    s->setup_from_raster(cond, nrows, ncols, four_neighbors, avg_resistances, reg, ground);
  }
  *out = h.release();
  return CSGPU_OK;
  CS_API_END
}

int csgpu_raster_setup_poly(const void* cond, const int32_t* polymap, int64_t nrows, int64_t ncols, int val_bytes,
                            int four_neighbors, int avg_resistances, int reg, const csgpu_opts* opts,
                            csgpu_handle** out) {
  if (!polymap)
    return csgpu_raster_setup_grounded(cond, nullptr, nrows, ncols, val_bytes, four_neighbors, avg_resistances, reg, opts,
                                       out);
  CS_API_BEGIN
  if (!cond || !out || nrows <= 0 || ncols <= 0) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  int rc = check_common(nrows * ncols, 0, val_bytes, opts);
  if (rc) return rc;
  if (nrows * ncols * 9 >= ((int64_t)1 << 31)) {
    g_last_error = "raster too large for int32 device indexing";
    return CSGPU_BAD_ARGS;
  }
  csgpu_opts o;
  if (opts) o = *opts; else csgpu_default_opts(&o);
  o.node_row = o.node_col = nullptr;
  if (val_bytes == 8) single_level_precision_raster(o, cond, polymap, nrows * ncols);
  std::unique_ptr<csgpu_handle> h(new csgpu_handle());
  if (val_bytes == 8 && o.precond_bytes == 4) {
    auto* s = new csgpu::Solver<double, float>(o);
    h->solver.reset(s);
    s->setup_from_raster_poly(cond, polymap, nrows, ncols, four_neighbors, avg_resistances, reg);
    // (no heterogeneity fallback here: on a polygon handle the strength test also counts the cells the STRENGTHENED
    // polygon interiors cut off at the polygons' rims, so hetero_frac does not measure the raster's heterogeneity)
  } else if (val_bytes == 8) {
    auto* s = new csgpu::Solver<double, double>(o);
    h->solver.reset(s);
    s->setup_from_raster_poly(cond, polymap, nrows, ncols, four_neighbors, avg_resistances, reg);
  } else {
    auto* s = new csgpu::Solver<float, float>(o);
    h->solver.reset(s);
    s->setup_from_raster_poly(cond, polymap, nrows, ncols, four_neighbors, avg_resistances, reg);
  }
  *out = h.release();
  return CSGPU_OK;
  CS_API_END
}

int csgpu_raster_setup(const void* cond, int64_t nrows, int64_t ncols, int val_bytes, int four_neighbors,
                       int avg_resistances, int reg, const csgpu_opts* opts, csgpu_handle** out) {
  return csgpu_raster_setup_grounded(cond, nullptr, nrows, ncols, val_bytes, four_neighbors, avg_resistances, reg, opts,
                                     out);
}

int csgpu_solve_raster(csgpu_handle* h, const void* source, void* curr_out, void* volt_out, csgpu_stats* stats) {
  CS_API_BEGIN
  if (!h || !source) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  h->solver->solve_raster(source, curr_out, volt_out, s);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

int csgpu_raster_nodemap(csgpu_handle* h, int32_t* nodemap_out, int64_t* nrows, int64_t* ncols) {
  CS_API_BEGIN
  if (!h) {
    g_last_error = "null handle";
    return CSGPU_BAD_ARGS;
  }
  h->solver->raster_nodemap(nodemap_out, nrows, ncols);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_components(csgpu_handle* h, int32_t* component_out, int64_t* ncomponents) {
  CS_API_BEGIN
  if (!h) {
    g_last_error = "null handle";
    return CSGPU_BAD_ARGS;
  }
  const int64_t nc = h->solver->components(component_out);
  if (ncomponents) *ncomponents = nc;
  return CSGPU_OK;
  CS_API_END
}

int csgpu_get_info(const csgpu_handle* h, csgpu_info* info) {
  CS_API_BEGIN
  if (!h || !info) {
    g_last_error = "null handle";
    return CSGPU_BAD_ARGS;
  }
  h->solver->get_info(info);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_solve_pairs(csgpu_handle* h, const int64_t* src, const int64_t* dst, int64_t npairs, void* volt_out,
                      const int64_t* gather_idx, int64_t ngather, void* gathered_out, void* resist_out,
                      csgpu_stats* stats) {
  CS_API_BEGIN
  if (!h || npairs < 0 || (npairs > 0 && (!src || !dst)) || ngather < 0 || (ngather > 0 && !gather_idx)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  if (npairs == 0) return CSGPU_OK;
  h->solver->solve_pairs(src, dst, npairs, volt_out, gather_idx, ngather, gathered_out, resist_out, s);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

int csgpu_solve_pairs_currents(csgpu_handle* h, const int64_t* src, const int64_t* dst, int64_t npairs,
                               const int32_t* weights, void* volt_out, void* curr_out, void* cum_curr_inout,
                               void* max_curr_inout, void* branch_out, void* resist_out, csgpu_stats* stats) {
  CS_API_BEGIN
  if (!h || npairs < 0 || (npairs > 0 && (!src || !dst))) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  if (npairs == 0) return CSGPU_OK;
  h->solver->solve_pairs(src, dst, npairs, volt_out, nullptr, 0, nullptr, resist_out, s, weights, curr_out, cum_curr_inout,
                         max_curr_inout, branch_out);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

int csgpu_solve_rhs(csgpu_handle* h, const void* rhs, int64_t nrhs, void* x_out, csgpu_stats* stats) {
  CS_API_BEGIN
  if (!h || nrhs < 0 || (nrhs > 0 && (!rhs || !x_out))) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  if (nrhs == 0) return CSGPU_OK;
  h->solver->solve_rhs(rhs, nrhs, x_out, s);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

int csgpu_solve_region_pairs(csgpu_handle* h, const int64_t* set_ptr, const int64_t* set_nodes, int64_t nsets,
                             const int64_t* src_set, const int64_t* dst_set, int64_t npairs, double* resistances,
                             csgpu_stats* stats) {
  CS_API_BEGIN
  if (!h || nsets < 0 || npairs < 0 || (npairs > 0 && (!set_ptr || !set_nodes || !src_set || !dst_set || !resistances))) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  if (npairs == 0) return CSGPU_OK;
  h->solver->solve_region_pairs(set_ptr, set_nodes, nsets, src_set, dst_set, npairs, resistances, s);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

static int grounded_call(csgpu_handle* h, const csgpu::GroundedJob& job, int64_t nrhs, csgpu_stats* stats) {
  CS_API_BEGIN
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  if (nrhs == 0) return CSGPU_OK;
  h->solver->solve_grounded(job, nrhs, s);
  if (s->not_converged > 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "CG solver did not converge: relative residual %g exceeds tolerance 1e-4 (%d of %d right-hand sides)",
             s->max_relres, s->not_converged, s->nrhs);
    g_last_error = buf;
    return CSGPU_NOT_CONVERGED;
  }
  return CSGPU_OK;
  CS_API_END
}

int csgpu_solve_grounded(csgpu_handle* h, const void* rhs, int64_t nrhs, const int64_t* ground_ptr,
                         const int64_t* ground_idx, void* x_out, void* curr_out, csgpu_stats* stats) {
  if (!h || nrhs < 0 || (nrhs > 0 && (!rhs || !x_out || !ground_ptr)) ||
      (nrhs > 0 && ground_ptr[nrhs] > ground_ptr[0] && !ground_idx)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu::GroundedJob job;
  job.rhs = rhs;
  job.gptr = ground_ptr;
  job.gidx = ground_idx;
  job.x_out = x_out;
  job.curr_out = curr_out;
  return grounded_call(h, job, nrhs, stats);
}

static bool sources_args_ok(int64_t nrhs, const int64_t* source_ptr, const int64_t* source_idx, const int64_t* ground_ptr,
                            const int64_t* ground_idx, const int64_t* check_node, const void* check_out) {
  if (nrhs < 0) return false;
  if (nrhs == 0) return true;
  if (!source_ptr || !ground_ptr) return false;
  if (source_ptr[nrhs] > source_ptr[0] && !source_idx) return false;
  if (ground_ptr[nrhs] > ground_ptr[0] && !ground_idx) return false;
  if ((check_out != nullptr) != (check_node != nullptr)) return false;
  return true;
}

int csgpu_solve_sources(csgpu_handle* h, int64_t nrhs, const int64_t* source_ptr, const int64_t* source_idx,
                        const void* source_val, const int64_t* ground_ptr, const int64_t* ground_idx,
                        const int64_t* check_node, void* check_out, void* x_out, void* curr_out, void* cum_curr_inout,
                        void* max_curr_inout, csgpu_stats* stats) {
  if (!h || !sources_args_ok(nrhs, source_ptr, source_idx, ground_ptr, ground_idx, check_node, check_out)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu::GroundedJob job;
  job.sptr = source_ptr;
  job.sidx = source_idx;
  job.sval = source_val;
  job.gptr = ground_ptr;
  job.gidx = ground_idx;
  job.check = check_node;
  job.check_out = check_out;
  job.x_out = x_out;
  job.curr_out = curr_out;
  job.cum_inout = cum_curr_inout;
  job.max_inout = max_curr_inout;
  return grounded_call(h, job, nrhs, stats);
}

int csgpu_spmv_bench(csgpu_handle* h, int k, int reps, double* avg_ms) {
  CS_API_BEGIN
  if (!h || !avg_ms || reps < 1 || !(k == 1 || k == 2 || k == 4 || k == 8 || k == 16 || k == 32)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  *avg_ms = h->solver->spmv_bench(k, reps);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_spmv_host(csgpu_handle* h, const void* x, void* y, int k) {
  CS_API_BEGIN
  if (!h || !x || !y || !(k == 1 || k == 2 || k == 4 || k == 8 || k == 16 || k == 32)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  h->solver->spmv_host(x, y, k);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_level_spmv_host(csgpu_handle* h, int lvl, int which, const void* x, void* y, int k, double* dots) {
  CS_API_BEGIN
  if (!h || !x || !y || which < 0 || which > 6 || !(k == 1 || k == 2 || k == 4 || k == 8 || k == 16 || k == 32)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  h->solver->level_spmv_host(lvl, which, x, y, k, dots);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_get_level_matrix(const csgpu_handle* h, int lvl, int which, int64_t* nrows, int64_t* ncols, int64_t* nnz,
                           int32_t* rowptr, int32_t* colidx, void* vals) {
  CS_API_BEGIN
  if (!h || which < 0 || which > 5) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  h->solver->get_level_matrix(lvl, which, nrows, ncols, nnz, rowptr, colidx, vals);
  return CSGPU_OK;
  CS_API_END
}

int csgpu_dia_product_host(csgpu_handle* h, const void* z, const void* p_in, const double* beta, void* p_out, void* y,
                           int k, double* dots) {
  CS_API_BEGIN
  if (!h || !z || !p_in || !beta || !p_out || !y || !(k == 1 || k == 2 || k == 4 || k == 8 || k == 16 || k == 32)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  h->solver->dia_product_host(z, p_in, beta, p_out, y, k, dots);
  return CSGPU_OK;
  CS_API_END
}

void csgpu_free(csgpu_handle* h) { delete h; }

int64_t csgpu_trim_memory(int device) { return (int64_t)csgpu::device_pool().trim(device); }

// ---- several devices behind one handle ---------------------------------------------------------------------------
}  // extern "C"

struct csgpu_multi {
  std::vector<csgpu_handle*> handles;  // one per device slot
  std::vector<int> devices;
  int val_bytes = 8;
  int batch = 8;
  std::vector<double> busy_s;
  std::vector<int64_t> pairs_done;
  ~csgpu_multi() {
    for (csgpu_handle* h : handles) delete h;
  }
};

namespace {

// device list of a multi handle: explicit list, or the first ndevices (<= 0: all) visible devices
int multi_devices(const int32_t* devices, int ndevices, std::vector<int>& out) {
  const int visible = csgpu_device_count();
  if (visible < 1) {
    g_last_error = "no HIP device visible";
    return CSGPU_HIP_ERROR;
  }
  out.clear();
  if (devices) {
    for (int i = 0; i < ndevices; ++i) {
      if (devices[i] < 0 || devices[i] >= visible) {
        g_last_error = "device ordinal out of range";
        return CSGPU_BAD_ARGS;
      }
      out.push_back(devices[i]);
    }
  } else {
    const int nd = ndevices <= 0 ? visible : std::min(ndevices, visible);
    for (int i = 0; i < nd; ++i) out.push_back(i);
  }
  if (out.empty()) {
    g_last_error = "empty device list";
    return CSGPU_BAD_ARGS;
  }
  return CSGPU_OK;
}

// one host thread per device runs `build(device, &handle)`; the first failure is reported
template <class F>
int multi_build(const csgpu_opts* opts, const int32_t* devices, int ndevices, int val_bytes, csgpu_multi** out, F build) {
  std::vector<int> devs;
  int rc = multi_devices(devices, ndevices, devs);
  if (rc) return rc;
  std::unique_ptr<csgpu_multi> m(new csgpu_multi());
  m->devices = devs;
  m->val_bytes = val_bytes;
  csgpu_opts o;
  if (opts) o = *opts; else csgpu_default_opts(&o);
  m->batch = std::max(1, std::min(o.batch, (int)csgpu::kMaxK));
  m->handles.assign(devs.size(), nullptr);
  m->busy_s.assign(devs.size(), 0.0);
  m->pairs_done.assign(devs.size(), 0);
  std::vector<int> codes(devs.size(), CSGPU_OK);
  std::vector<std::string> msgs(devs.size());
  std::vector<std::thread> th;
  for (size_t i = 0; i < devs.size(); ++i)
    th.emplace_back([&, i] {
      csgpu_opts oi = o;
      oi.device = devs[i];
      codes[i] = build(&oi, &m->handles[i]);
      if (codes[i]) msgs[i] = csgpu_last_error();  // thread-local message of this worker
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < devs.size(); ++i)
    if (codes[i]) {
      g_last_error = "device " + std::to_string(devs[i]) + ": " + msgs[i];
      return codes[i];
    }
  *out = m.release();
  return CSGPU_OK;
}

// the per-device cumulative / maximum node-current vectors of a job combined into the caller's arrays in slot order (sum /
// max: deterministic), the index range split over host threads
void multi_combine_maps(size_t nd, int64_t n, size_t vb, const std::vector<std::vector<char>>& cum,
                        const std::vector<std::vector<char>>& mx, void* cum_curr_inout, void* max_curr_inout) {
  auto combine = [&](int64_t lo, int64_t hi) {
    for (size_t i = 0; i < nd; ++i) {
      if (cum_curr_inout && !cum[i].empty()) {
        if (vb == 8) {
          double* o = (double*)cum_curr_inout;
          const double* a = (const double*)cum[i].data();
          for (int64_t k = lo; k < hi; ++k) o[k] += a[k];
        } else {
          float* o = (float*)cum_curr_inout;
          const float* a = (const float*)cum[i].data();
          for (int64_t k = lo; k < hi; ++k) o[k] += a[k];
        }
      }
      if (max_curr_inout && !mx[i].empty()) {
        if (vb == 8) {
          double* o = (double*)max_curr_inout;
          const double* a = (const double*)mx[i].data();
          for (int64_t k = lo; k < hi; ++k) o[k] = a[k] > o[k] ? a[k] : o[k];
        } else {
          float* o = (float*)max_curr_inout;
          const float* a = (const float*)mx[i].data();
          for (int64_t k = lo; k < hi; ++k) o[k] = a[k] > o[k] ? a[k] : o[k];
        }
      }
    }
  };
  const int nt = (int)std::max<size_t>(1, std::min<size_t>(nd, (size_t)(n / 1000000 + 1)));
  std::vector<std::thread> ct;
  for (int t = 1; t < nt; ++t) ct.emplace_back(combine, n * t / nt, n * (t + 1) / nt);
  combine(0, n / nt);
  for (auto& t : ct) t.join();
}

}  // namespace

extern "C" {

int csgpu_multi_setup(const void* rowptr, const void* colidx, const void* vals, int64_t n, int64_t nnz, int idx_bytes,
                      int val_bytes, int index_base, const csgpu_opts* opts, const int32_t* devices, int ndevices,
                      csgpu_multi** out) {
  CS_API_BEGIN
  if (!out) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  return multi_build(opts, devices, ndevices, val_bytes, out, [&](const csgpu_opts* o, csgpu_handle** h) {
    return csgpu_setup(rowptr, colidx, vals, n, nnz, idx_bytes, val_bytes, index_base, o, h);
  });
  CS_API_END
}

int csgpu_multi_raster_setup(const void* cond, int64_t nrows, int64_t ncols, int val_bytes, int four_neighbors,
                             int avg_resistances, int reg, const csgpu_opts* opts, const int32_t* devices, int ndevices,
                             csgpu_multi** out) {
  CS_API_BEGIN
  if (!out) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  return multi_build(opts, devices, ndevices, val_bytes, out, [&](const csgpu_opts* o, csgpu_handle** h) {
    return csgpu_raster_setup(cond, nrows, ncols, val_bytes, four_neighbors, avg_resistances, reg, o, h);
  });
  CS_API_END
}

int csgpu_multi_solve_pairs(csgpu_multi* m, const int64_t* src, const int64_t* dst, int64_t npairs,
                            const int64_t* gather_idx, int64_t ngather, void* gathered_out, void* resist_out,
                            csgpu_stats* stats) {
  CS_API_BEGIN
  if (!m || npairs < 0 || (npairs > 0 && (!src || !dst)) || ngather < 0 || (ngather > 0 && !gather_idx)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  const size_t nd = m->handles.size();
  std::fill(m->busy_s.begin(), m->busy_s.end(), 0.0);
  std::fill(m->pairs_done.begin(), m->pairs_done.end(), 0);
  if (npairs == 0) return CSGPU_OK;
  auto t0 = std::chrono::steady_clock::now();
  // chunk = one batch; fewer batches than devices: shrink the chunk so that every device gets one
  int64_t chunk = m->batch;
  if ((npairs + chunk - 1) / chunk < (int64_t)nd) chunk = std::max<int64_t>(1, (npairs + (int64_t)nd - 1) / (int64_t)nd);
  const int64_t nchunks = (npairs + chunk - 1) / chunk;
  std::atomic<int64_t> next{0};
  std::vector<int> codes(nd, CSGPU_OK);
  std::vector<std::string> msgs(nd);
  std::vector<csgpu_stats> st(nd);
  for (auto& x : st) memset(&x, 0, sizeof(x));
  const size_t vb = (size_t)m->val_bytes;
  auto worker = [&](size_t slot) {
    auto w0 = std::chrono::steady_clock::now();
    for (;;) {
      const int64_t c = next.fetch_add(1);
      if (c >= nchunks) break;
      const int64_t off = c * chunk, cnt = std::min(chunk, npairs - off);
      csgpu_stats cs;
      const int rc = csgpu_solve_pairs(m->handles[slot], src + off, dst + off, cnt, nullptr, gather_idx, ngather,
                                       gathered_out ? (char*)gathered_out + (size_t)off * ngather * vb : nullptr,
                                       resist_out ? (char*)resist_out + (size_t)off * vb : nullptr, &cs);
      st[slot].total_iters += cs.total_iters;
      st[slot].max_iters = std::max(st[slot].max_iters, cs.max_iters);
      st[slot].max_relres = std::max(st[slot].max_relres, cs.max_relres);
      st[slot].device_ms += cs.device_ms;
      st[slot].cg_spmv_ms += cs.cg_spmv_ms;
      st[slot].cg_spmv_calls += cs.cg_spmv_calls;
      st[slot].not_converged += cs.not_converged;
      st[slot].graph_launches += cs.graph_launches;
      st[slot].polished_batches += cs.polished_batches;
      st[slot].cg_spmv_bytes = cs.cg_spmv_bytes;
      st[slot].resid_ms += cs.resid_ms;
      st[slot].resid_calls += cs.resid_calls;
      st[slot].resid_bytes = cs.resid_bytes;
      st[slot].resid_fused = cs.resid_fused;
      st[slot].batch = std::max(st[slot].batch, cs.batch);
      m->pairs_done[slot] += cnt;
      if (rc != CSGPU_OK && codes[slot] == CSGPU_OK) {
        codes[slot] = rc;
        msgs[slot] = csgpu_last_error();
        if (rc != CSGPU_NOT_CONVERGED) break;  // hard error: stop this device; a non-converged chunk is only reported
      }
    }
    m->busy_s[slot] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < nd; ++i) th.emplace_back(worker, i);
  worker(0);
  for (auto& t : th) t.join();
  s->nrhs = (int)npairs;
  int rc_out = CSGPU_OK;
  for (size_t i = 0; i < nd; ++i) {
    s->total_iters += st[i].total_iters;
    s->max_iters = std::max(s->max_iters, st[i].max_iters);
    s->max_relres = std::max(s->max_relres, st[i].max_relres);
    s->device_ms = std::max(s->device_ms, st[i].device_ms);
    s->cg_spmv_ms += st[i].cg_spmv_ms;
    s->cg_spmv_calls += st[i].cg_spmv_calls;
    s->not_converged += st[i].not_converged;
    s->graph_launches += st[i].graph_launches;
    s->polished_batches += st[i].polished_batches;
    s->cg_spmv_bytes = std::max(s->cg_spmv_bytes, st[i].cg_spmv_bytes);
    s->resid_ms += st[i].resid_ms;
    s->resid_calls += st[i].resid_calls;
    s->resid_bytes = std::max(s->resid_bytes, st[i].resid_bytes);
    s->resid_fused = std::max(s->resid_fused, st[i].resid_fused);
    s->batch = std::max(s->batch, st[i].batch);
    if (codes[i] != CSGPU_OK && (rc_out == CSGPU_OK || rc_out == CSGPU_NOT_CONVERGED)) {
      rc_out = codes[i];
      g_last_error = "device " + std::to_string(m->devices[i]) + ": " + msgs[i];
    }
  }
  s->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc_out;
  CS_API_END
}

// csgpu_multi_solve_pairs with the reference's cumulative / maximum current maps (src/out.jl:96-107 merged serially in
// src/core.jl:262-285): batches are dealt round-robin to the devices (batch b -> slot b % ndevices), every device runs ITS
// pairs as ONE csgpu_solve_pairs_currents call, so its cumulative / maximum node-current vectors stay in its HBM for the
// whole job and cross PCIe once; the per-device vectors are then combined on the host in slot order (sum / max of
// ndevices n-vectors: deterministic).
int csgpu_multi_solve_pairs_currents(csgpu_multi* m, const int64_t* src, const int64_t* dst, int64_t npairs,
                                     const int32_t* weights, void* cum_curr_inout, void* max_curr_inout, void* resist_out,
                                     csgpu_stats* stats) {
  CS_API_BEGIN
  if (!m || npairs < 0 || (npairs > 0 && (!src || !dst))) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  const size_t nd = m->handles.size();
  std::fill(m->busy_s.begin(), m->busy_s.end(), 0.0);
  std::fill(m->pairs_done.begin(), m->pairs_done.end(), 0);
  if (npairs == 0) return CSGPU_OK;
  auto t0 = std::chrono::steady_clock::now();
  csgpu_info info;
  int rc = csgpu_get_info(m->handles[0], &info);
  if (rc) return rc;
  const int64_t n = info.n;
  const size_t vb = (size_t)m->val_bytes;
  int64_t chunk = m->batch;
  if ((npairs + chunk - 1) / chunk < (int64_t)nd) chunk = std::max<int64_t>(1, (npairs + (int64_t)nd - 1) / (int64_t)nd);
  // the pairs of every slot, in the caller's order
  std::vector<std::vector<int64_t>> idx(nd);
  for (int64_t p = 0; p < npairs; ++p) idx[(size_t)((p / chunk) % (int64_t)nd)].push_back(p);
  std::vector<int> codes(nd, CSGPU_OK);
  std::vector<std::string> msgs(nd);
  std::vector<csgpu_stats> st(nd);
  for (auto& x : st) memset(&x, 0, sizeof(x));
  std::vector<std::vector<char>> cum(nd), mx(nd);
  auto worker = [&](size_t slot) {
    auto w0 = std::chrono::steady_clock::now();
    const std::vector<int64_t>& mine = idx[slot];
    const int64_t cnt = (int64_t)mine.size();
    if (cnt > 0) {
      std::vector<int64_t> s_(cnt), d_(cnt);
      std::vector<int32_t> w_(cnt);
      for (int64_t k = 0; k < cnt; ++k) {
        s_[k] = src[mine[k]];
        d_[k] = dst[mine[k]];
        w_[k] = weights ? weights[mine[k]] : 1;
      }
      if (cum_curr_inout) cum[slot].assign((size_t)n * vb, 0);
      if (max_curr_inout) mx[slot].assign((size_t)n * vb, 0);
      std::vector<char> res((size_t)cnt * vb);
      codes[slot] = csgpu_solve_pairs_currents(m->handles[slot], s_.data(), d_.data(), cnt, w_.data(), nullptr, nullptr,
                                               cum_curr_inout ? cum[slot].data() : nullptr,
                                               max_curr_inout ? mx[slot].data() : nullptr, nullptr, res.data(), &st[slot]);
      if (codes[slot]) msgs[slot] = csgpu_last_error();
      if (resist_out && (codes[slot] == CSGPU_OK || codes[slot] == CSGPU_NOT_CONVERGED))
        for (int64_t k = 0; k < cnt; ++k) memcpy((char*)resist_out + (size_t)mine[k] * vb, res.data() + (size_t)k * vb, vb);
      m->pairs_done[slot] = cnt;
    }
    m->busy_s[slot] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < nd; ++i) th.emplace_back(worker, i);
  worker(0);
  for (auto& t : th) t.join();
  s->nrhs = (int)npairs;
  int rc_out = CSGPU_OK;
  for (size_t i = 0; i < nd; ++i) {
    s->total_iters += st[i].total_iters;
    s->max_iters = std::max(s->max_iters, st[i].max_iters);
    s->max_relres = std::max(s->max_relres, st[i].max_relres);
    s->device_ms = std::max(s->device_ms, st[i].device_ms);
    s->cg_spmv_ms += st[i].cg_spmv_ms;
    s->cg_spmv_calls += st[i].cg_spmv_calls;
    s->not_converged += st[i].not_converged;
    s->graph_launches += st[i].graph_launches;
    s->polished_batches += st[i].polished_batches;
    s->cg_spmv_bytes = std::max(s->cg_spmv_bytes, st[i].cg_spmv_bytes);
    s->resid_ms += st[i].resid_ms;
    s->resid_calls += st[i].resid_calls;
    s->resid_bytes = std::max(s->resid_bytes, st[i].resid_bytes);
    s->resid_fused = std::max(s->resid_fused, st[i].resid_fused);
    s->batch = std::max(s->batch, st[i].batch);
    if (codes[i] != CSGPU_OK && (rc_out == CSGPU_OK || rc_out == CSGPU_NOT_CONVERGED)) {
      rc_out = codes[i];
      g_last_error = "device " + std::to_string(m->devices[i]) + ": " + msgs[i];
    }
  }
  if (rc_out == CSGPU_OK || rc_out == CSGPU_NOT_CONVERGED)
    multi_combine_maps(nd, n, vb, cum, mx, cum_curr_inout, max_curr_inout);
  s->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc_out;
  CS_API_END
}

}  // extern "C"

namespace {

// Columns of a Dirichlet-masked job dealt over the devices of a set: slot i takes the contiguous range [lo_i, hi_i) of the
// caller's columns (sizes differ by at most one column; the pointer arrays of the caller are used as they are, offset by
// the range) and runs it as ONE call on its handle, so that the cumulative / maximum node-current vectors of its columns
// stay in its HBM for the whole job; the per-device vectors are combined on the host in slot order (deterministic).
// job_of(lo, hi) returns the GroundedJob of that range without the cumulative / maximum pointers.
template <class F>
int multi_grounded(csgpu_multi* m, int64_t nrhs, void* cum_curr_inout, void* max_curr_inout, csgpu_stats* stats, F job_of) {
  csgpu_stats local;
  csgpu_stats* s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  const size_t nd = m->handles.size();
  std::fill(m->busy_s.begin(), m->busy_s.end(), 0.0);
  std::fill(m->pairs_done.begin(), m->pairs_done.end(), 0);
  if (nrhs == 0) return CSGPU_OK;
  auto t0 = std::chrono::steady_clock::now();
  csgpu_info info;
  int rc = csgpu_get_info(m->handles[0], &info);
  if (rc) return rc;
  const int64_t n = info.n;
  const size_t vb = (size_t)m->val_bytes;
  std::vector<int> codes(nd, CSGPU_OK);
  std::vector<std::string> msgs(nd);
  std::vector<csgpu_stats> st(nd);
  for (auto& x : st) memset(&x, 0, sizeof(x));
  std::vector<std::vector<char>> cum(nd), mx(nd);
  auto worker = [&](size_t slot) {
    auto w0 = std::chrono::steady_clock::now();
    const int64_t base = nrhs / (int64_t)nd, rem = nrhs % (int64_t)nd;
    const int64_t lo = (int64_t)slot * base + std::min<int64_t>((int64_t)slot, rem);
    const int64_t hi = lo + base + ((int64_t)slot < rem ? 1 : 0);
    if (hi > lo) {
      csgpu::GroundedJob job = job_of(lo, hi);
      if (cum_curr_inout) {
        cum[slot].assign((size_t)n * vb, 0);
        job.cum_inout = cum[slot].data();
      }
      if (max_curr_inout) {
        mx[slot].assign((size_t)n * vb, 0);
        job.max_inout = mx[slot].data();
      }
      codes[slot] = grounded_call(m->handles[slot], job, hi - lo, &st[slot]);
      if (codes[slot]) msgs[slot] = csgpu_last_error();
      m->pairs_done[slot] = hi - lo;
    }
    m->busy_s[slot] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < nd; ++i) th.emplace_back(worker, i);
  worker(0);
  for (auto& t : th) t.join();
  s->nrhs = (int)nrhs;
  int rc_out = CSGPU_OK;
  for (size_t i = 0; i < nd; ++i) {
    s->total_iters += st[i].total_iters;
    s->max_iters = std::max(s->max_iters, st[i].max_iters);
    s->max_relres = std::max(s->max_relres, st[i].max_relres);
    s->device_ms = std::max(s->device_ms, st[i].device_ms);
    s->cg_spmv_ms += st[i].cg_spmv_ms;
    s->cg_spmv_calls += st[i].cg_spmv_calls;
    s->not_converged += st[i].not_converged;
    s->graph_launches += st[i].graph_launches;
    s->polished_batches += st[i].polished_batches;
    s->cg_spmv_bytes = std::max(s->cg_spmv_bytes, st[i].cg_spmv_bytes);
    s->resid_ms += st[i].resid_ms;
    s->resid_calls += st[i].resid_calls;
    s->resid_bytes = std::max(s->resid_bytes, st[i].resid_bytes);
    s->resid_fused = std::max(s->resid_fused, st[i].resid_fused);
    s->batch = std::max(s->batch, st[i].batch);
    if (codes[i] != CSGPU_OK && (rc_out == CSGPU_OK || rc_out == CSGPU_NOT_CONVERGED)) {
      rc_out = codes[i];
      g_last_error = "device " + std::to_string(m->devices[i]) + ": " + msgs[i];
    }
  }
  if ((rc_out == CSGPU_OK || rc_out == CSGPU_NOT_CONVERGED) && (cum_curr_inout || max_curr_inout))
    multi_combine_maps(nd, n, vb, cum, mx, cum_curr_inout, max_curr_inout);
  s->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc_out;
}

}  // namespace

extern "C" {

int csgpu_multi_solve_grounded(csgpu_multi* m, const void* rhs, int64_t nrhs, const int64_t* ground_ptr,
                               const int64_t* ground_idx, void* x_out, void* curr_out, csgpu_stats* stats) {
  CS_API_BEGIN
  if (!m || nrhs < 0 || (nrhs > 0 && (!rhs || !x_out || !ground_ptr)) ||
      (nrhs > 0 && ground_ptr[nrhs] > ground_ptr[0] && !ground_idx)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_info info;
  int rc = csgpu_get_info(m->handles[0], &info);
  if (rc) return rc;
  const size_t colb = (size_t)info.n * (size_t)m->val_bytes;
  return multi_grounded(m, nrhs, nullptr, nullptr, stats, [&](int64_t lo, int64_t) {
    csgpu::GroundedJob job;
    job.rhs = (const char*)rhs + (size_t)lo * colb;
    job.gptr = ground_ptr + lo;
    job.gidx = ground_idx;
    job.x_out = (char*)x_out + (size_t)lo * colb;
    job.curr_out = curr_out ? (char*)curr_out + (size_t)lo * colb : nullptr;
    return job;
  });
  CS_API_END
}

int csgpu_multi_solve_sources(csgpu_multi* m, int64_t nrhs, const int64_t* source_ptr, const int64_t* source_idx,
                              const void* source_val, const int64_t* ground_ptr, const int64_t* ground_idx,
                              const int64_t* check_node, void* check_out, void* x_out, void* curr_out,
                              void* cum_curr_inout, void* max_curr_inout, csgpu_stats* stats) {
  CS_API_BEGIN
  if (!m || !sources_args_ok(nrhs, source_ptr, source_idx, ground_ptr, ground_idx, check_node, check_out)) {
    g_last_error = "bad arguments";
    return CSGPU_BAD_ARGS;
  }
  csgpu_info info;
  int rc = csgpu_get_info(m->handles[0], &info);
  if (rc) return rc;
  const size_t vb = (size_t)m->val_bytes, colb = (size_t)info.n * vb;
  return multi_grounded(m, nrhs, cum_curr_inout, max_curr_inout, stats, [&](int64_t lo, int64_t) {
    csgpu::GroundedJob job;
    job.sptr = source_ptr + lo;
    job.sidx = source_idx;
    job.sval = source_val;
    job.gptr = ground_ptr + lo;
    job.gidx = ground_idx;
    job.check = check_node ? check_node + lo : nullptr;
    job.check_out = check_out ? (char*)check_out + (size_t)lo * vb : nullptr;
    job.x_out = x_out ? (char*)x_out + (size_t)lo * colb : nullptr;
    job.curr_out = curr_out ? (char*)curr_out + (size_t)lo * colb : nullptr;
    return job;
  });
  CS_API_END
}

int csgpu_multi_device_count(const csgpu_multi* m) { return m ? (int)m->handles.size() : 0; }

csgpu_handle* csgpu_multi_handle(csgpu_multi* m, int slot) {
  if (!m || slot < 0 || slot >= (int)m->handles.size()) return nullptr;
  return m->handles[slot];
}

int csgpu_multi_last_busy(const csgpu_multi* m, double* busy_s, int64_t* pairs_done) {
  if (!m) return CSGPU_BAD_ARGS;
  for (size_t i = 0; i < m->handles.size(); ++i) {
    if (busy_s) busy_s[i] = m->busy_s[i];
    if (pairs_done) pairs_done[i] = m->pairs_done[i];
  }
  return CSGPU_OK;
}

void csgpu_multi_free(csgpu_multi* m) { delete m; }

}  // extern "C"
