// stencil.h -- lattice ("symmetric diagonal") form of a raster Laplacian and the CG product evaluated from it.
//
// GPU counterpart of `A*p` inside Krylov.cg (reference call site src/core.jl:639) for the matrices the reference's
// raster path produces (construct_graph, src/raster/pairwise.jl:316-362): with every cell of an R x C raster valid,
// node i (column-major numbering, pairwise.jl:273-275) is coupled to i +- 1, i +- (R-1), i +- R, i +- (R+1) only, and
// the matrix is symmetric. Such a matrix needs no column indices and only half of its off-diagonal values:
//
//   rows[i] = { A[i,i], A[i,i+1], A[i,i+R-1], A[i,i+R], A[i,i+R+1] }          (5 values per node, 0 where absent)
//   y[i]    = rows[i].d x[i] + sum_o ( rows[i].a_o x[i+o] + rows[i-o].a_o x[i-o] ),   o in {1, R-1, R, R+1}
//
// i.e. 5*sizeof(T) matrix bytes per row instead of 9*(sizeof(T)+4)+4 for CSR (fp64: 40 B vs 112 B). The form is
// DETECTED from the CSR matrix the host hands over (dia_from_csr: every entry must sit on one of the nine diagonals
// and the matrix must be bit-symmetric), so it serves csgpu_setup (Julia-built graphs) and csgpu_raster_setup alike;
// anything else (NODATA holes, polygons, networks) keeps the CSR product.
//
// Kernel shape (dia_cg_kernel): a workgroup owns a 2-D tile of the raster -- TI consecutive rows (TI = 256 / lanes per
// node) of SEG consecutive raster columns -- and marches through its columns. Per column it streams the column's
// TI*K vector entries and TI*5 matrix values with fully coalesced, contiguous loads (a raster column segment is
// contiguous in node numbering) into a 4-slot LDS ring; the nine-point product of the previous column is then taken
// out of LDS (x) and registers (sliding 3x3 window). Every x entry is fetched from HBM once per tile plus a one-cell
// halo ((TI+2)(SEG+2)/(TI*SEG) ~ 1.06 at 64 x 64), independent of cache behaviour.
//
// Fusion: the search-direction update p = z + beta p of Krylov.cg is evaluated while the column is staged (halo
// included), so the product reads z and the old p and writes the new p (to a second buffer: neighbouring tiles still
// read the old values) and A p, plus the partials of p'Ap -- one pass instead of an update pass and a product pass.
//
// Algorithmic bytes per launch: n*5*sizeof(T) + n*K*(2*sizeof(XT) [z, p_in] + sizeof(XT) [p_out] + sizeof(T) [Ap]).
#pragma once
#include "blas1.h"
#include "spmv.h"

namespace csgpu {

// Streams that are written once / read once per launch can bypass cache allocation (-DCSGPU_DIA_NT=1: the stores,
// =2: also the loads of r and z). A/B: profiles/r2_nontemporal_ab.json.
#ifndef CSGPU_DIA_NT
#define CSGPU_DIA_NT 0
#endif
template <class T, int N>
__device__ __forceinline__ void dia_store(SpmvVec<T, N>* p, const SpmvVec<T, N>& v) {
#if CSGPU_DIA_NT >= 1 && defined(__HIPCC__)
  typedef T VT __attribute__((ext_vector_type(N)));
  VT t;
#pragma unroll
  for (int q = 0; q < N; ++q) t[q] = v.e[q];
  __builtin_nontemporal_store(t, reinterpret_cast<VT*>(p));
#else
  *p = v;
#endif
}
template <class T, int N>
__device__ __forceinline__ SpmvVec<T, N> dia_load(const SpmvVec<T, N>* p) {
#if CSGPU_DIA_NT >= 2 && defined(__HIPCC__)
  typedef T VT __attribute__((ext_vector_type(N)));
  const VT t = __builtin_nontemporal_load(reinterpret_cast<const VT*>(p));
  SpmvVec<T, N> v;
#pragma unroll
  for (int q = 0; q < N; ++q) v.e[q] = t[q];
  return v;
#else
  return *p;
#endif
}


// slot of a diagonal offset (0 = main diagonal), -1 when the offset is not one of the lattice's
__device__ __forceinline__ int dia_slot(int64_t d, int R) {
  if (d == 0) return 0;
  if (d == 1) return 1;
  if (d == R - 1) return 2;
  if (d == R) return 3;
  if (d == R + 1) return 4;
  return -1;
}

// pass 1: scatter the upper-triangle entries (and the diagonal) of every row into rows[]; flag entries off the lattice
template <class T>
__global__ __launch_bounds__(256) void dia_fill_kernel(int n, int R, const int* __restrict__ rp,
                                                       const int* __restrict__ ci, const T* __restrict__ va,
                                                       T* __restrict__ rows, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int64_t d = (int64_t)ci[k] - i;
      const int s = dia_slot(d < 0 ? -d : d, R);
      if (s < 0) {
        atomicOr(bad, 1);
        continue;
      }
      if (d >= 0) rows[(size_t)i * 5 + s] = va[k];
    }
  }
}

// pass 2: the lower-triangle entries must equal their mirror images bit for bit
template <class T>
__global__ __launch_bounds__(256) void dia_check_kernel(int n, int R, const int* __restrict__ rp,
                                                        const int* __restrict__ ci, const T* __restrict__ va,
                                                        const T* __restrict__ rows, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int64_t d = (int64_t)ci[k] - i;
      if (d >= 0) continue;
      const int s = dia_slot(-d, R);
      if (s < 0 || !(rows[(size_t)ci[k] * 5 + s] == va[k])) atomicOr(bad, 1);
    }
  }
}

// Build the lattice form of A for period R; returns false (out untouched) when A is not such a matrix.
// trusted: the matrix was built here from a raster (bit-symmetric by construction): the mirror-image pass is skipped
template <class T>
inline bool dia_from_csr(const Csr<T>& A, int R, Dia<T>& out, hipStream_t st, bool trusted = false) {
  const int n = A.nrows;
  if (A.nrows != A.ncols || R < 4 || n < 4 * R || (n % R) != 0) return false;
  DBuf rows((size_t)n * 5 * sizeof(T));
  DBuf bad = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(rows.p, 0, rows.bytes, st));
  CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  const int g = grid_for(n);
  hipLaunchKernelGGL((dia_fill_kernel<T>), dim3(g), dim3(256), 0, st, n, R, A.rp(), A.ci(), A.va(), dptr<T>(rows),
                     dptr<int>(bad));
  if (!trusted)
    hipLaunchKernelGGL((dia_check_kernel<T>), dim3(g), dim3(256), 0, st, n, R, A.rp(), A.ci(), A.va(),
                       (const T*)dptr<T>(rows), dptr<int>(bad));
  check_launch("lattice form");
  if (read_int(dptr<int>(bad), st) != 0) return false;
  out.n = n;
  out.R = R;
  out.rows = std::move(rows);
  return true;
}

// what the marching kernel computes per tile column
enum DiaMode {
  DIA_PLAIN = 0,  // y = A x, partials of x'y                                  (x = pin)
  DIA_CG = 1,     // x = z + beta pin -> pout ; y = A x ; partials of x'y      (Krylov.cg's p-update fused in)
  DIA_SQP = 4,    // DIA_SQ with x = the right-hand sides of a batch of pair solves, never stored: column c of x is -1 at
                  // node psrc[c] and +1 at pdst[c] (pcg.h: the first V-cycle of a batch whose r0 is not materialised)
  DIA_SQ = 2,     // y = S x + Q xc, partials of x'y                            (second product of the two-product V(1,1)
                  //                                                             level, amg_setup.h: A = S in lattice form,
                  //                                                             Q in its index-free 3x3-tile form
                  //                                                             (LatticeQ), xc the coarse solution,
                  //                                                             staged tile by tile in LDS)
  DIA_RUPD = 3    // r -= alpha (A x) -> r, rp ; optionally xsol += alpha x     (Krylov.cg's residual update with the
                  //                                                             product A p RECOMPUTED from the lattice
                  //                                                             form instead of stored by DIA_CG and
                  //                                                             re-read: x = pin = the new p; partials
                  //                                                             of r'r when `partials` is set)
};

template <class T, class XT>
struct DiaArgs {
  int64_t n;
  int R, C;
  int nstrips, nseg, seg;  // tiles: nstrips (along a raster column) x nseg (across columns), seg columns each
  const T* rows;
  const CgScalars* S;      // beta[c], all_done (may be null: beta = `beta0` for every column, never skipped)
  const XT* z;             // FUSE: p = z + beta * pin; otherwise pin is the input vector and z is unused
  const XT* pin;
  XT* pout;                // FUSE: new search direction (must not alias pin)
  T* y;                    // A p
  double* partials;        // [gridDim.x][K] partials of p'(A p)
  const double* beta_dev;  // test hook: per-column beta (device pointer) when S is null; null = 0
  const int* skip;         // optional device flag: non-zero turns the launch into a no-op
  const T* qell;           // DIA_SQ: index-free Q, [n][9] (LatticeQ)
  int Rc, Cc;              // DIA_SQ: coarse lattice
  const XT* xc;            // DIA_SQ: coarse vector, interleaved [Rc*Cc][K]
  T* r;                    // DIA_RUPD: residual (in/out), interleaved [n][K]
  XT* rp;                  // DIA_RUPD: copy of the new residual in the preconditioner's precision (null when XT == T)
  T* xsol;                 // DIA_RUPD: optional whole solution vector, xsol += alpha x
  const int* restart = nullptr;  // DIA_RUPD, streaming pair solves (pcg_stream_pairs): columns with restart[c] != 0 take a
                                 // new pair -- their residual is ZEROED here (stream_restart_kernel then writes the +-1)
  const T* bsub = nullptr; // DIA_PLAIN: y = bsub - A x instead of A x (residual of a lattice level, pcg.h)
  const T* xadd = nullptr; // DIA_SQ: y = xadd + S x + Q xc (second half of a lattice V(2,2) level, pcg.h)
  const int* psrc = nullptr;  // DIA_SQP: node ids of the pairs of the batch (device, one per column), columns >= pcols and
  const int* pdst = nullptr;  // pairs with psrc == pdst have a zero right-hand side
  int pcols = 0;
};

template <class T, class XT, int K>
struct DiaShape {
  static constexpr int VEC = 16 / (int)sizeof(XT);
  static constexpr int CPL = K < VEC ? K : VEC;  // columns per lane
  static constexpr int LPR = K / CPL;            // lanes per node
  static constexpr int TI = 256 / LPR;           // raster rows per tile
  static constexpr int MELEMS = 5 * (TI + 2);    // matrix values staged per raster column (halo rows included)
  static constexpr int MU = (MELEMS + 255) / 256;
  static constexpr int QELEMS = 9 * TI;          // DIA_SQ: values of Q staged per raster column
  static constexpr int QU = (QELEMS + 255) / 256;
  static constexpr int CR = TI / 3 + 4;          // DIA_SQ: coarse rows staged per coarse column (one tile above / below)
};

// Second launch bound = waves per SIMD the register allocation must leave room for. The marching kernels are bound by
// the bytes in flight per CU (a step waits for the loads of the step before), so for the CG product one more resident
// workgroup per CU is worth two spilled registers: 128 VGPRs / 4 waves instead of 130 / 3 -> 4.27 -> 3.44 ms (measured,
// profiles/r2_launch_bounds_ab.json). The second product (142 VGPRs) spills 13 registers under the same bound and gets
// 1.9x slower; the residual update gains nothing from 3 waves instead of 2: both stay unconstrained.
// (narrow batches, K <= 4, hold CPL = K columns per lane on fewer lanes per node and cannot reach 4 waves: no bound there)
#ifndef CSGPU_RUPD_WAVES
#define CSGPU_RUPD_WAVES 1   // A/B knob (build-time): waves per SIMD asked of the residual update at K = 32
#endif
template <class T, class XT, int K, int MODE>
__global__ __launch_bounds__(256, ((MODE == DIA_CG && K >= 8) ? 4 : ((MODE == DIA_RUPD && K >= 32) ? CSGPU_RUPD_WAVES : 1))) void dia_cg_kernel(
    DiaArgs<T, XT> a) {
  constexpr bool FUSE = MODE == DIA_CG;
  typedef DiaShape<T, XT, K> SH;
  constexpr int CPL = SH::CPL, LPR = SH::LPR, TI = SH::TI, MELEMS = SH::MELEMS, MU = SH::MU;
  typedef SpmvVec<XT, CPL> XV;
  typedef SpmvVec<T, CPL> YV;
  constexpr int QELEMS = SH::QELEMS, QU = SH::QU, CR = SH::CR;
  constexpr bool SQ = MODE == DIA_SQ || MODE == DIA_SQP;
  constexpr bool SYNTH = MODE == DIA_SQP;
  static_assert(CR * LPR <= 256, "one lane per staged coarse entry");
  __shared__ XV s_x[4][(TI + 2) * LPR];
  __shared__ T s_m[4][MELEMS];
  __shared__ T s_q[SQ ? 4 : 1][SQ ? QELEMS : 1];    // DIA_SQ: Q values of raster columns j-1 .. j+2 (ring)
  __shared__ XV s_xc[SQ ? 4 : 1][SQ ? CR * LPR : 1];  // DIA_SQ: coarse columns J-1 .. J+2 (ring), rows Ilo .. Ilo+CR-1
  __shared__ double s_red[4 * K];
  if (a.S && a.S->all_done) return;
  if (a.skip && *a.skip) return;
  const int tid = threadIdx.x;
  const int t = tid / LPR;        // row of the tile owned by this lane
  const int lq = tid % LPR;       // which CPL-wide slice of the K columns
  const int c0 = lq * CPL;
  T beta[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q)
    beta[q] = FUSE ? (a.S ? (T)a.S->beta[c0 + q] : (a.beta_dev ? (T)a.beta_dev[c0 + q] : T(0))) : T(0);
  T alpha[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) alpha[q] = (MODE == DIA_RUPD) ? (T)a.S->alpha[c0 + q] : T(0);
  bool fresh[CPL];  // DIA_RUPD: the column takes a new pair (streaming solves)
#pragma unroll
  for (int q = 0; q < CPL; ++q) fresh[q] = (MODE == DIA_RUPD) && a.restart && a.restart[c0 + q] != 0;
  double dot_acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) dot_acc[q] = 0.0;
  int64_t ps[SYNTH ? CPL : 1], pd[SYNTH ? CPL : 1];  // DIA_SQP: the pair of each of this lane's columns (-1: none)
  if (SYNTH) {
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const bool on = c0 + q < a.pcols && a.psrc[c0 + q] != a.pdst[c0 + q];
      ps[q] = on ? a.psrc[c0 + q] : -1;
      pd[q] = on ? a.pdst[c0 + q] : -1;
    }
  }
  auto synth = [&](int64_t id) {
    XV v;
#pragma unroll
    for (int q = 0; q < CPL; ++q) v.e[q] = id == ps[SYNTH ? q : 0] ? XT(-1) : (id == pd[SYNTH ? q : 0] ? XT(1) : XT(0));
    return v;
  };

  const int ntiles = a.nstrips * a.nseg;
  // XCD-aware tile walk: workgroup b runs on XCD b % 8; give every XCD a contiguous range of tiles (strip index
  // fastest), so the one-cell halos shared by neighbouring tiles are re-read out of that XCD's L2
  int t_first = blockIdx.x, t_last = ntiles, t_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, chunk = (ntiles + 7) >> 3;
    t_first = xcd * chunk + (blockIdx.x >> 3);
    t_last = min(ntiles, (xcd + 1) * chunk);
    t_step = gridDim.x >> 3;
  }
  for (int tile = t_first; tile < t_last; tile += t_step) {
    const int si = tile % a.nstrips, sj = tile / a.nstrips;
    const int i0 = si * TI;
    const int j0 = sj * a.seg, j1 = min(a.C, j0 + a.seg);
    const bool row_on = i0 + t < a.R;  // rows past the raster's last row belong to no tile
    // DIA_SQ: coarse rows staged = tiles of the strip's rows plus one above and one below
    const int Ilo = SQ ? min(i0 / 3, a.Rc - 1) - 1 : 0;
    const int crow = SQ ? min((i0 + t) / 3, a.Rc - 1) - Ilo : 0;  // this lane's tile row inside the staged coarse column
    // Column in flight: the values of a raster column are loaded into registers one step before the column's turn and
    // written to the LDS ring right before it, so its loads overlap the product of the column before. In DIA_CG the two
    // loaded vectors stay raw until then: combining them at load time would make every step wait for its own loads.
    // (Two columns in flight -- a second register set -- was tried: 150-196 VGPRs, 2-3 waves/SIMD, scratch; dropped.)
    XV xr, xh;              // vector entries of the tile's rows / of the two halo rows (DIA_CG: the old p)
    XV zr, zh;              // DIA_CG: z
    T mr[MU];
    T qr[SQ ? QU : 1];
    XV xcr;
    YV r_pre;               // DIA_RUPD: the residual entries of the NEXT column, in flight one step ahead like the vector loads
    int pend_c = -1, next_c = 0;
    auto load_r = [&](int jc) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) r_pre.e[q] = T(0);
      const int64_t id = (int64_t)jc * a.R + i0 + t;
      if (MODE == DIA_RUPD && jc < j1 && row_on && id < a.n) r_pre = dia_load(reinterpret_cast<const YV*>(a.r + (size_t)id * K + c0));
    };
    auto load_coarse = [&](int Jc) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) xcr.e[q] = XT(0);
      if (tid < CR * LPR) {
        const int Ic = Ilo + tid / LPR;
        if (Jc >= 0 && Jc < a.Cc && Ic >= 0 && Ic < a.Rc)
          xcr = *reinterpret_cast<const XV*>(a.xc + ((size_t)Jc * a.Rc + Ic) * K + (tid % LPR) * CPL);
      }
    };
    auto store_coarse = [&](int Jc) {
      if (tid < CR * LPR) s_xc[Jc & 3][tid] = xcr;
    };
    auto load_column = [&](int jc) {
      // node id of tile row -1 (the halo row above) in raster column jc; ids outside [0, n) read as zero
      const int64_t base = (int64_t)jc * a.R + i0 - 1;
      {
        const int64_t id = base + 1 + t;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          xr.e[q] = XT(0);
          zr.e[q] = XT(0);
        }
        if (id >= 0 && id < a.n) {
          const size_t e = (size_t)id * K + c0;
          if (SYNTH) xr = synth(id);
          else xr = *reinterpret_cast<const XV*>(a.pin + e);
          if (FUSE) zr = dia_load(reinterpret_cast<const XV*>(a.z + e));
        }
      }
      if (tid < 2 * LPR) {  // halo rows: tile row -1 (lanes 0..LPR-1) and tile row TI (lanes LPR..2LPR-1)
        const int64_t id = base + (tid < LPR ? 0 : TI + 1);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          xh.e[q] = XT(0);
          zh.e[q] = XT(0);
        }
        if (id >= 0 && id < a.n) {
          const size_t e = (size_t)id * K + c0;  // (tid < 2 LPR: tid % LPR is the lane's own column slice)
          if (SYNTH) xh = synth(id);
          else xh = *reinterpret_cast<const XV*>(a.pin + e);
          if (FUSE) zh = *reinterpret_cast<const XV*>(a.z + e);
        }
      }
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        const int64_t g = base * 5 + e;
        mr[u] = (e < MELEMS && g >= 0 && g < a.n * 5) ? a.rows[g] : T(0);
      }
      if (SQ) {  // Q values of the PREVIOUS raster column (the one whose product is taken right after this store)
        const int64_t qb = ((int64_t)(jc - 1) * a.R + i0) * 9;
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          const int e = tid + u * 256;
          const int64_t g = qb + e;
          qr[u] = (jc - 1 >= j0 && e < QELEMS && g < a.n * 9) ? a.qell[g] : T(0);
        }
      }
    };
    auto store_column = [&](int jc) {
      const int slot = jc & 3;
      if (FUSE) {  // p = z + beta p (halo included); the tile's own entries are the new search direction
        XV v;
#pragma unroll
        // (beta == 0: the first step of a column -- p = z exactly, whatever the old p holds; same bits as the fma otherwise)
        for (int q = 0; q < CPL; ++q) v.e[q] = beta[q] == T(0) ? zr.e[q] : (XT)fma(beta[q], (T)xr.e[q], (T)zr.e[q]);
        const int64_t id = (int64_t)jc * a.R + i0 + t;
        if (jc >= j0 && jc < j1 && row_on && id < a.n) dia_store(reinterpret_cast<XV*>(a.pout + (size_t)id * K + c0), v);
        xr = v;
#pragma unroll
        for (int q = 0; q < CPL; ++q) xh.e[q] = beta[q] == T(0) ? zh.e[q] : (XT)fma(beta[q], (T)xh.e[q], (T)zh.e[q]);
      }
      s_x[slot][(t + 1) * LPR + lq] = xr;
      if (tid < 2 * LPR) s_x[slot][(tid < LPR ? 0 : TI + 1) * LPR + lq] = xh;
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        if (e < MELEMS) s_m[slot][e] = mr[u];
      }
      if (SQ) {
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          const int e = tid + u * 256;
          if (e < QELEMS) s_q[(jc - 1) & 3][e] = qr[u];
        }
      }
    };
    __syncthreads();  // previous tile finished with the ring
    // prologue: columns j0-1 and j0 into the ring, column j0+1 in flight
    load_column(j0 - 1);
    store_column(j0 - 1);
    load_column(j0);
    store_column(j0);
    load_column(j0 + 1);
    if (MODE == DIA_RUPD) load_r(j0);
    if (SQ) {  // coarse columns J(j0)-1 .. J(j0)+2 up front; later ones one per step, two steps ahead of their use
      const int Jb = min(j0 / 3, a.Cc - 1);
      for (int d = -1; d <= 2; ++d) {
        load_coarse(Jb + d);
        store_coarse(Jb + d);
      }
      next_c = Jb + 3;
    }
    __syncthreads();
    // sliding 3x3 window of x: xw[dj][di] = x(row t-1+di of the tile, raster column j-1+dj)
    XV xw[3][3];
#pragma unroll
    for (int di = 0; di < 3; ++di) {
      xw[1][di] = s_x[(j0 - 1) & 3][(t + di) * LPR + lq];
      xw[2][di] = s_x[j0 & 3][(t + di) * LPR + lq];
    }
    for (int j = j0; j < j1; ++j) {
      store_column(j + 1);                 // the column loaded one step ago
      if (SQ && pend_c >= 0) {
        store_coarse(pend_c);
        pend_c = -1;
      }
      if (j + 2 <= j1) load_column(j + 2); // next one in flight while this column is computed
      YV rv_cur;
      if (MODE == DIA_RUPD) {              // this column's residual arrived during the previous step; fetch the next one
        rv_cur = r_pre;
        load_r(j + 1);
      }
      if (SQ) {
        const int need = min((j + 2) / 3, a.Cc - 1) + 1;  // last coarse column read two steps from now
        if (need >= next_c) {
          load_coarse(next_c);
          pend_c = next_c++;
        }
      }
      __syncthreads();
#pragma unroll
      for (int di = 0; di < 3; ++di) {
        xw[0][di] = xw[1][di];
        xw[1][di] = xw[2][di];
        xw[2][di] = s_x[(j + 1) & 3][(t + di) * LPR + lq];
      }
      if (row_on) {
        const T* mc = s_m[j & 3];        // matrix rows of raster column j   (tile rows -1 .. TI at 5*(row+1))
        const T* mp = s_m[(j - 1) & 3];  // matrix rows of raster column j-1
        const int me = 5 * (t + 1);
        // ascending column order, like the CSR row: i-R-1, i-R, i-R+1, i-1, i, i+1, i+R-1, i+R, i+R+1
        const T w_mm = mp[me - 5 + 4];   // rows[i-R-1].a_{R+1}
        const T w_m0 = mp[me + 3];       // rows[i-R].a_R
        const T w_mp = mp[me + 5 + 2];   // rows[i-R+1].a_{R-1}
        const T w_0m = mc[me - 5 + 1];   // rows[i-1].a_1
        const T w_00 = mc[me + 0];
        const T w_0p = mc[me + 1];
        const T w_pm = mc[me + 2];
        const T w_p0 = mc[me + 3];
        const T w_pp = mc[me + 4];
        YV out;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          T s = w_mm * (T)xw[0][0].e[q];
          s = fma(w_m0, (T)xw[0][1].e[q], s);
          s = fma(w_mp, (T)xw[0][2].e[q], s);
          s = fma(w_0m, (T)xw[1][0].e[q], s);
          s = fma(w_00, (T)xw[1][1].e[q], s);
          s = fma(w_0p, (T)xw[1][2].e[q], s);
          s = fma(w_pm, (T)xw[2][0].e[q], s);
          s = fma(w_p0, (T)xw[2][1].e[q], s);
          s = fma(w_pp, (T)xw[2][2].e[q], s);
          out.e[q] = s;
        }
        const int64_t id = (int64_t)j * a.R + i0 + t;
        if (SQ) {
          // + Q xc: the 3 x 3 block of tiles around this cell's tile, coarse values out of the staged columns
          const T* qrow = s_q[j & 3] + 9 * t;
          const int Jj = min(j / 3, a.Cc - 1);
#pragma unroll 1
          for (int dj = 0; dj < 3; ++dj) {  // (not unrolled: three coarse values live at a time instead of nine)
            const XV* xcol = s_xc[(Jj + dj - 1) & 3];
#pragma unroll
            for (int di = 0; di < 3; ++di) {
              const T w = qrow[dj * 3 + di];
              const XV xv = xcol[(crow + di - 1) * LPR + lq];
#pragma unroll
              for (int q = 0; q < CPL; ++q) out.e[q] = fma(w, (T)xv.e[q], out.e[q]);
            }
          }
        }
        if (MODE == DIA_RUPD) {
          // r -= alpha * (A p): same arithmetic as cg_update_r_kernel on a stored A p
          const size_t e = (size_t)id * K + c0;
          const YV rv = rv_cur;
          YV rn;
          XV rq;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            rn.e[q] = fresh[q] ? T(0) : rv.e[q] - alpha[q] * out.e[q];
            rq.e[q] = (XT)rn.e[q];
            if (a.partials) dot_acc[q] += (double)rn.e[q] * (double)rn.e[q];
          }
          dia_store(reinterpret_cast<YV*>(a.r + e), rn);
          if (a.rp) dia_store(reinterpret_cast<XV*>(a.rp + e), rq);
          if (a.xsol) {
            const YV xv = *reinterpret_cast<const YV*>(a.xsol + e);
            YV xn;
#pragma unroll
            for (int q = 0; q < CPL; ++q) xn.e[q] = fma(alpha[q], (T)xw[1][1].e[q], xv.e[q]);
            *reinterpret_cast<YV*>(a.xsol + e) = xn;
          }
        } else {
          if (SQ && a.xadd) {
            const YV av = *reinterpret_cast<const YV*>(a.xadd + (size_t)id * K + c0);
#pragma unroll
            for (int q = 0; q < CPL; ++q) out.e[q] += av.e[q];
          }
          if (MODE == DIA_PLAIN && a.bsub) {
            const YV bv = *reinterpret_cast<const YV*>(a.bsub + (size_t)id * K + c0);
#pragma unroll
            for (int q = 0; q < CPL; ++q) out.e[q] = bv.e[q] - out.e[q];
          }
#pragma unroll
          for (int q = 0; q < CPL; ++q) dot_acc[q] += (double)(T)xw[1][1].e[q] * (double)out.e[q];
          if (a.y) dia_store(reinterpret_cast<YV*>(a.y + (size_t)id * K + c0), out);
        }
      }
    }
  }
  if (!a.partials) return;  // (block-uniform)
  // lanes owning the same columns sit LPR apart: reduce over lane bits >= log2(LPR), then across the 4 waves via LDS
  const int lane = tid & 63, w = tid >> 6;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    double v = dot_acc[q];
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane < LPR) s_red[w * K + lane * CPL + q] = v;
  }
  __syncthreads();
  if (tid < K) a.partials[(size_t)blockIdx.x * K + tid] = s_red[tid] + s_red[K + tid] + s_red[2 * K + tid] + s_red[3 * K + tid];
}

// raster columns per tile (tuning knob Knobs::dia_seg). Default 32; 64 for batches of 32 columns, whose tiles hold half
// as many rows (fp64: 16): measured at 10000^2, K = 32 (profiles/r4_dia_seg_k32.txt): 346.3 / 341.3 / 339.5 / 340.2 ms per
// 16 pairs at 32 / 48 / 64 / 96 (fp64), 211.7 / 205.6 / 205.3 / 204.5 (mixed). At K = 16 the knob is inside the noise
// (profiles/r3_tile_shape_knobs.txt).
inline int dia_seg(int K = 16) {
  const int v = knobs().dia_seg;
  const int seg = v <= 0 ? 0 : (v < 4 ? 4 : v);
  return seg > 0 ? seg : (K >= 32 ? 64 : 32);
}

template <class T, class XT, int K>
inline void dia_tiling(const Dia<T>& D, int& nstrips, int& nseg, int& seg, int& grid) {
  const int C = (int)(D.n / D.R);
  seg = std::min(dia_seg(K), std::max(C, 1));
  nstrips = ceil_div(D.R, DiaShape<T, XT, K>::TI);
  nseg = ceil_div(C, seg);
  int64_t g = (int64_t)nstrips * nseg;
  // never more workgroups than rows in the dot-partial arrays (PcgWork::ensure)
  const int64_t cap = std::min<int64_t>(std::max(1024, spmv_grid_cap()), std::max<size_t>(16384, spmv_grid_upper(D.n)));
  if (g > cap) g = cap;
  if (g >= 64) g &= ~(int64_t)7;
  grid = (int)std::max<int64_t>(g, 1);
}

// number of workgroups (= rows of dot partials) of the fused CG product
template <class T, class XT, int K>
inline int dia_grid(const Dia<T>& D) {
  int a, b, c, g;
  dia_tiling<T, XT, K>(D, a, b, c, g);
  return g;
}

// p_out = z + beta p_in ; y = A p_out ; partials of p_out' y   (beta and the skip flag from the CG scalars S)
template <class T, class XT, int K>
inline void dia_cg_product(const Dia<T>& D, const CgScalars* S, const XT* z, const XT* pin, XT* pout, T* y,
                           double* partials, hipStream_t st, const double* beta_dev = nullptr) {
  DiaArgs<T, XT> a;
  a.n = D.n;
  a.R = D.R;
  a.C = (int)(D.n / D.R);
  int grid;
  dia_tiling<T, XT, K>(D, a.nstrips, a.nseg, a.seg, grid);
  a.rows = D.data();
  a.S = S;
  a.z = z;
  a.pin = pin;
  a.pout = pout;
  a.y = y;
  a.partials = partials;
  a.beta_dev = beta_dev;
  a.skip = nullptr;
  a.qell = nullptr;
  a.Rc = a.Cc = 0;
  a.xc = nullptr;
  a.r = nullptr;
  a.rp = nullptr;
  a.xsol = nullptr;
  hipLaunchKernelGGL((dia_cg_kernel<T, XT, K, DIA_CG>), dim3(grid), dim3(256), 0, st, a);
}

// r -= alpha (A p) with A p recomputed from the lattice form (p = the search direction dia_cg_product just wrote; it
// need not have stored A p: y = nullptr there); rp = XT copy of the new r (may be null); xsol += alpha p (may be null);
// partials of r'r (may be null). alpha and the skip flag from the CG scalars S.
template <class T, class XT, int K>
inline void dia_residual_update(const Dia<T>& D, const CgScalars* S, const XT* p, T* r, XT* rp, T* xsol,
                                double* partials_rr, hipStream_t st, const int* restart = nullptr) {
  DiaArgs<T, XT> a;
  a.n = D.n;
  a.R = D.R;
  a.C = (int)(D.n / D.R);
  int grid;
  dia_tiling<T, XT, K>(D, a.nstrips, a.nseg, a.seg, grid);
  a.rows = D.data();
  a.S = S;
  a.z = nullptr;
  a.pin = p;
  a.pout = nullptr;
  a.y = nullptr;
  a.partials = partials_rr;
  a.beta_dev = nullptr;
  a.skip = nullptr;
  a.qell = nullptr;
  a.Rc = a.Cc = 0;
  a.xc = nullptr;
  a.r = r;
  a.rp = rp;
  a.xsol = xsol;
  a.restart = restart;
  hipLaunchKernelGGL((dia_cg_kernel<T, XT, K, DIA_RUPD>), dim3(grid), dim3(256), 0, st, a);
}

// out = S b + Q xc with the partials of b'out: second product of the two-product V(1,1) level (S in lattice form, Q in
// its index-free tile form)
template <class T, int K>
inline void dia_sq_product(const Dia<T>& Sd, const LatticeQ<T>& Q, const T* b, const T* xc, T* out, double* partials,
                           const int* skip, hipStream_t st, const T* xadd = nullptr, const int* psrc = nullptr,
                           const int* pdst = nullptr, int pcols = 0) {
  DiaArgs<T, T> a;
  a.n = Sd.n;
  a.R = Sd.R;
  a.C = (int)(Sd.n / Sd.R);
  int grid;
  dia_tiling<T, T, K>(Sd, a.nstrips, a.nseg, a.seg, grid);
  a.rows = Sd.data();
  a.S = nullptr;
  a.z = nullptr;
  a.pin = b;
  a.pout = nullptr;
  a.y = out;
  a.partials = partials;
  a.beta_dev = nullptr;
  a.skip = skip;
  a.qell = Q.data();
  a.Rc = Q.Rc;
  a.Cc = Q.Cc;
  a.xc = xc;
  a.r = nullptr;
  a.rp = nullptr;
  a.xsol = nullptr;
  a.xadd = xadd;
  if (psrc) {  // b is the batch's pair right-hand sides, synthesised in the kernel (b itself is not read)
    if constexpr (K >= 16) {
      a.psrc = psrc;
      a.pdst = pdst;
      a.pcols = pcols;
      hipLaunchKernelGGL((dia_cg_kernel<T, T, K, DIA_SQP>), dim3(grid), dim3(256), 0, st, a);
      return;
    } else {
      CS_REQUIRE(false, CSGPU_INTERNAL, "synthesised pair right-hand sides: batches of 16 / 32 columns only");
    }
  }
  hipLaunchKernelGGL((dia_cg_kernel<T, T, K, DIA_SQ>), dim3(grid), dim3(256), 0, st, a);
}

// y = D x, or y = bsub - D x when bsub is given (D in lattice form; no dot partials)
template <class T, int K>
inline void dia_apply(const Dia<T>& D, const T* x, T* y, const T* bsub, const int* skip, hipStream_t st) {
  DiaArgs<T, T> a;
  a.n = D.n;
  a.R = D.R;
  a.C = (int)(D.n / D.R);
  int grid;
  dia_tiling<T, T, K>(D, a.nstrips, a.nseg, a.seg, grid);
  a.rows = D.data();
  a.S = nullptr;
  a.z = nullptr;
  a.pin = x;
  a.pout = nullptr;
  a.y = y;
  a.partials = nullptr;
  a.beta_dev = nullptr;
  a.skip = skip;
  a.qell = nullptr;
  a.Rc = a.Cc = 0;
  a.xc = nullptr;
  a.r = nullptr;
  a.rp = nullptr;
  a.xsol = nullptr;
  a.bsub = bsub;
  hipLaunchKernelGGL((dia_cg_kernel<T, T, K, DIA_PLAIN>), dim3(grid), dim3(256), 0, st, a);
}

// Lattice form of S = 2 w D^-1 - w D^-1 A w D^-1 (w = damped-Jacobi weight) from the lattice form of A (precision U,
// rounded to T first -- the hierarchy's level-0 matrix is the element-wise rounded CG matrix) and dinv = 1/diag(A):
// the same values build_sq_kernel puts into the CSR form of [S Q].
// General form: two Jacobi sweeps with weights w0, w1 from a zero initial guess are x = S b with
// S = (w0 + w1) D^-1 - w0 w1 D^-1 A D^-1 (the two-product level: w0 = w1 = w).
template <class U, class T>
__global__ __launch_bounds__(256) void dia_build_s_kernel(int64_t n, int R, const U* __restrict__ arows,
                                                          const T* __restrict__ dinv, double w0, double w1,
                                                          T* __restrict__ srows) {
  const int off[5] = {0, 1, R - 1, R, R + 1};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double wi = w0 * (double)dinv[i];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int64_t j = i + off[s];
      const T av = (T)arows[i * 5 + s];
      double v = 0.0;
      if (j < n) {
        v = -wi * (double)av * w1 * (double)dinv[j];
        if (s == 0) v += (w0 + w1) * (double)dinv[i];
      }
      srows[i * 5 + s] = (T)v;
    }
  }
}

template <class U, class T>
inline void dia_build_s(const Dia<U>& A, const T* dinv, double w0, Dia<T>& S, hipStream_t st, double w1 = -1.0) {
  if (w1 < 0.0) w1 = w0;
  S.n = A.n;
  S.R = A.R;
  S.rows.alloc((size_t)A.n * 5 * sizeof(T));
  hipLaunchKernelGGL((dia_build_s_kernel<U, T>), dim3(grid_for(A.n)), dim3(256), 0, st, A.n, A.R, A.data(), dinv, w0, w1,
                     dptr<T>(S.rows));
  check_launch("lattice form of S");
}

}  // namespace csgpu
