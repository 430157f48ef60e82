// amg_setup.h -- K3/K4/K5: smoothed-aggregation AMG setup entirely on the device.
//
// GPU counterpart of AlgebraicMultigrid.jl `smoothed_aggregation(matrix; ...)` as the reference calls it
// (src/core.jl:164-167, src/raster/advanced.jl:308; algorithm restated in SURVEY.md section 2.3):
//
//   reference (sequential)                          here (parallel, deterministic, atomic-free floating point)
//   ----------------------------------------------  -------------------------------------------------------------
//   SymmetricStrength(theta)                        same predicate, evaluated on the fly (no S matrix stored)
//   StandardAggregation: greedy 3-pass sweep        distance-2 maximal independent set with fixed priorities:
//     (seeds a node whose neighbours are all free,    raster coordinates -> 3x3 tile centres first (the pattern the
//     i.e. a lexicographic MIS(2))                    greedy sweep yields on rasters), hashed node id otherwise;
//                                                     neighbours join their root, distance-2 nodes join their
//                                                     most strongly coupled aggregated neighbour
//   fit_candidates with B = 1                       T_i = sqrt(size_i / size_agg(i)), sizes tracked as exact integers
//   JacobiProlongation(4/3), local weighting        P = T - (4/3) Dl^-1 A T, Dl_i = sum_j |a_ij|      (same formula)
//   R = P'                                          explicit transpose (count / scan / fill / per-row rank sort)
//   A_c = R*A*P                                     two row-wise multiway-merge SpGEMMs (sorted, deterministic)
//   improve_candidates (4 Gauss-Seidel sweeps)      omitted: B = 1 already satisfies A*B ~ 0 for a Laplacian
//   GaussSeidel pre/post smoother                   damped Jacobi, omega = omega_s / rho_Gershgorin(D^-1 A)
//   Pinv coarse solver                              dense symmetric pseudo-inverse of the coarsest operator
//
// Parity with the reference is at the level of converged solutions (the reference pins nothing else).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <limits>

#include "prims.h"
#include "spmv.h"
#include "dia25.h"

namespace csgpu {

// ------------------------------------------------------------------------------------------------ row statistics
// diag[i] = a_ii, labs[i] = sum_j |a_ij|, block partial max of labs/|diag| (Gershgorin bound on rho(D^-1 A)).
template <class T>
__global__ __launch_bounds__(256) void row_stats_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        const T* __restrict__ va, T* __restrict__ diag,
                                                        T* __restrict__ labs, double* __restrict__ part_max,
                                                        double* __restrict__ part_dmax) {
  __shared__ double sm[4], smd[4];
  double mx = 0.0, dmx = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    T d = T(0), l = T(0);
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const T v = va[k];
      if (ci[k] == i) d += v;
      l += v < T(0) ? -v : v;
    }
    diag[i] = d;
    labs[i] = l;
    const double ad = d < T(0) ? -(double)d : (double)d;
    if (ad > 0.0) {
      const double q = (double)l / ad;
      mx = q > mx ? q : mx;
    }
    dmx = ad > dmx ? ad : dmx;
  }
  mx = wave_max(mx);
  dmx = wave_max(dmx);
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6] = mx;
    smd[threadIdx.x >> 6] = dmx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = sm[0], md = smd[0];
    for (int w = 1; w < 4; ++w) {
      m = sm[w] > m ? sm[w] : m;
      md = smd[w] > md ? smd[w] : md;
    }
    part_max[blockIdx.x] = m;
    part_dmax[blockIdx.x] = md;
  }
}

// ------------------------------------------------------------------------------------------------ aggregation
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

template <class T>
__device__ __forceinline__ bool is_strong(int i, int j, T a, const T* diag, double theta2) {
  if (j == i || a == T(0)) return false;
  if (theta2 == 0.0) return true;
  const double di = fabs((double)diag[i]), dj = fabs((double)diag[j]);
  return (double)a * (double)a >= theta2 * di * dj;
}

static const unsigned long long kKeyIn = ~0ull;
static const unsigned long long kKeyOut = 0ull;

// Undecided nodes carry their priority: [class:8][hash:24][node id:32]; class 2 = raster 3x3 tile centre.
__global__ __launch_bounds__(256) void mis_init_kernel(int n, unsigned long long* __restrict__ key,
                                                       const int* __restrict__ nrow, const int* __restrict__ ncol,
                                                       const long long* __restrict__ size_f) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (size_f && size_f[i] == 0) {  // a weightless row (empty tile of a cell-space raster) seeds no aggregate
      key[i] = kKeyOut;
      continue;
    }
    unsigned long long cls = 1;
    if (nrow && (nrow[i] % 3 == 1) && (ncol[i] % 3 == 1)) cls = 2;
    key[i] = (cls << 56) | ((unsigned long long)(hash32((unsigned)i) & 0xffffffu) << 32) | (unsigned)i;
  }
}

// out[i] = max(in[i], max over strong neighbours in[j])
template <class T>
__global__ __launch_bounds__(256) void mis_prop_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                       const T* __restrict__ va, const T* __restrict__ diag,
                                                       double theta2, const unsigned long long* __restrict__ in,
                                                       unsigned long long* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    unsigned long long m = in[i];
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int j = ci[k];
      if (is_strong(i, j, va[k], diag, theta2)) {
        const unsigned long long v = in[j];
        m = v > m ? v : m;
      }
    }
    out[i] = m;
  }
}

__global__ __launch_bounds__(256) void mis_decide_kernel(int n, unsigned long long* __restrict__ key,
                                                         const unsigned long long* __restrict__ k2,
                                                         int* __restrict__ undecided) {
  int cnt = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const unsigned long long me = key[i];
    if (me == kKeyIn || me == kKeyOut) continue;
    const unsigned long long m = k2[i];
    if (m == me)
      key[i] = kKeyIn;
    else if (m == kKeyIn)
      key[i] = kKeyOut;
    else
      ++cnt;
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(undecided, cnt);
}

__global__ __launch_bounds__(256) void mis_roots_kernel(int n, const unsigned long long* __restrict__ key,
                                                        int* __restrict__ flag) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) flag[i] = key[i] == kKeyIn ? 1 : 0;
}

// pass 1: roots take their scanned id, direct strong neighbours of a root join it
template <class T>
__global__ __launch_bounds__(256) void agg_pass1_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        const T* __restrict__ va, const T* __restrict__ diag,
                                                        double theta2, const unsigned long long* __restrict__ key,
                                                        const int* __restrict__ root_id, int* __restrict__ agg1) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int a = -1;
    if (key[i] == kKeyIn) {
      a = root_id[i];
    } else {
      for (int k = rp[i]; k < rp[i + 1]; ++k) {
        const int j = ci[k];
        if (is_strong(i, j, va[k], diag, theta2) && key[j] == kKeyIn) {
          a = root_id[j];
          break;
        }
      }
    }
    agg1[i] = a;
  }
}

// pass 2: the rest joins its most strongly coupled neighbour that was aggregated in pass 1. On a raster whose extent
// is known (gridR x gridC > 0) a left-over cell -- last row / column of a raster whose size is 1 mod 3, tile corners of
// a 4-neighbour raster -- prefers a neighbour of its OWN 3x3 tile (tile = (min(row/3, Rc-1), min(col/3, Cc-1))), so the
// aggregates stay the regular tiles and the transfer operators keep their index-free form (lattice.h).
template <class T>
__global__ __launch_bounds__(256) void agg_pass2_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                        const T* __restrict__ va, const T* __restrict__ diag,
                                                        double theta2, const int* __restrict__ agg1,
                                                        int* __restrict__ agg, int* __restrict__ orphan_flag,
                                                        const int* __restrict__ nrow, const int* __restrict__ ncol,
                                                        int gridR, int gridC, const long long* __restrict__ size_f) {
  const int Rc = (gridR + 1) / 3, Cc = (gridC + 1) / 3;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int a = agg1[i];
    if (a < 0 && size_f && size_f[i] == 0) {  // weightless and uncoupled: belongs nowhere (column 0 with weight 0)
      agg[i] = 0;
      orphan_flag[i] = 0;
      continue;
    }
    if (a < 0) {
      double best = -1.0;
      bool same_tile = false;
      int ti = 0, tj = 0;
      if (gridR > 0 && nrow) {
        ti = min(nrow[i] / 3, Rc - 1);
        tj = min(ncol[i] / 3, Cc - 1);
      }
      for (int k = rp[i]; k < rp[i + 1]; ++k) {
        const int j = ci[k];
        if (!is_strong(i, j, va[k], diag, theta2)) continue;
        const int aj = agg1[j];
        if (aj < 0) continue;
        const double w = fabs((double)va[k]);
        const bool st = gridR > 0 && nrow && min(nrow[j] / 3, Rc - 1) == ti && min(ncol[j] / 3, Cc - 1) == tj;
        if ((st && !same_tile) || (st == same_tile && w > best)) {
          best = w;
          a = aj;
          same_tile = st;
        }
      }
    }
    agg[i] = a;
    orphan_flag[i] = a < 0 ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void agg_orphans_kernel(int n, int* __restrict__ agg,
                                                          const int* __restrict__ orphan_scan, int nagg) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (agg[i] < 0) agg[i] = nagg + orphan_scan[i];
}

// coarse sizes (exact integers) and coarse raster coordinates (root's tile index)
__global__ __launch_bounds__(256) void agg_sizes_kernel(int n, const int* __restrict__ agg,
                                                        const long long* __restrict__ size_f,
                                                        unsigned long long* __restrict__ size_c) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    atomicAdd(&size_c[agg[i]], (unsigned long long)(size_f ? size_f[i] : 1));
}

__global__ __launch_bounds__(256) void agg_coords_kernel(int n, const int* __restrict__ agg,
                                                         const unsigned long long* __restrict__ key,
                                                         const int* __restrict__ orphan_flag,
                                                         const int* __restrict__ nrow, const int* __restrict__ ncol,
                                                         int* __restrict__ crow, int* __restrict__ ccol) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (key[i] == kKeyIn || orphan_flag[i]) {
      crow[agg[i]] = nrow[i] / 3;
      ccol[agg[i]] = ncol[i] / 3;
    }
}

template <class T>
__global__ __launch_bounds__(256) void candidate_kernel(int n, const long long* __restrict__ size, T* __restrict__ v) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v[i] = (T)sqrt((double)size[i]);
}

// tentative prolongator as CSR with exactly one entry per row
template <class T>
__global__ __launch_bounds__(256) void tentative_kernel(int n, const int* __restrict__ agg,
                                                        const long long* __restrict__ size_f,
                                                        const unsigned long long* __restrict__ size_c,
                                                        int* __restrict__ trp, int* __restrict__ tci,
                                                        T* __restrict__ tva) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) {
    trp[i] = i;
    if (i < n) {
      const int a = agg[i];
      tci[i] = a;
      const double sf = size_f ? (double)size_f[i] : 1.0;
      const double sc = (double)size_c[a];
      tva[i] = sc > 0.0 ? (T)sqrt(sf / sc) : T(0);  // (an aggregate of weightless nodes -- NODATA cells of a cell-space raster)
    }
  }
}

// ------------------------------------------------------------------------------------------------ SpGEMM
// Row-wise multiway merge C = A*B with a group of G lanes per row (G = 2..64, power of two).
// B's rows have sorted columns; every A entry keeps a cursor into its B row (global workspace `cursor`,
// one int per nonzero of A). Each step takes the minimum column under the cursors (group min through
// width-G shuffles), sums the matching products in a fixed order and advances those cursors, so C's rows
// come out sorted and the floating-point result is deterministic. NUMERIC=false only counts.
template <class T, int G, bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_merge_kernel(int nrows, const int* __restrict__ Arp,
                                                           const int* __restrict__ Aci, const T* __restrict__ Ava,
                                                           const int* __restrict__ Brp, const int* __restrict__ Bci,
                                                           const T* __restrict__ Bva, int* __restrict__ cursor,
                                                           int* __restrict__ Ccount, const int* __restrict__ Crp,
                                                           int* __restrict__ Cci, T* __restrict__ Cva) {
  const int lg = threadIdx.x % G;
  const int groups_per_block = 256 / G;
  for (int row = blockIdx.x * groups_per_block + threadIdx.x / G; row < nrows; row += gridDim.x * groups_per_block) {
    const int ab = Arp[row], ae = Arp[row + 1];
    for (int ka = ab + lg; ka < ae; ka += G) cursor[ka] = Brp[Aci[ka]];
    int count = 0;
    const int out0 = NUMERIC ? Crp[row] : 0;
    for (;;) {
      int cmin = 0x7fffffff;
      for (int ka = ab + lg; ka < ae; ka += G) {
        const int k = Aci[ka];
        const int cu = cursor[ka];
        if (cu < Brp[k + 1]) {
          const int cc = Bci[cu];
          cmin = cc < cmin ? cc : cmin;
        }
      }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        const int t = __shfl_xor(cmin, o, G);
        cmin = t < cmin ? t : cmin;
      }
      if (cmin == 0x7fffffff) break;
      T s = T(0);
      for (int ka = ab + lg; ka < ae; ka += G) {
        const int k = Aci[ka];
        const int cu = cursor[ka];
        if (cu < Brp[k + 1] && Bci[cu] == cmin) {
          if (NUMERIC) s += Ava[ka] * Bva[cu];
          cursor[ka] = cu + 1;
        }
      }
      if (NUMERIC) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
        if (lg == 0) {
          Cci[out0 + count] = cmin;
          Cva[out0 + count] = s;
        }
      }
      ++count;
    }
    if (!NUMERIC && lg == 0) Ccount[row] = count;
  }
}

template <class T, int G>
inline void spgemm_group(const Csr<T>& A, const Csr<T>& B, Csr<T>& C, hipStream_t st) {
  const int n = A.nrows;
  C.nrows = n;
  C.ncols = B.ncols;
  C.rowptr.alloc((size_t)(n + 1) * sizeof(int));
  DBuf cursor = dalloc<int>((size_t)std::max<int64_t>(A.nnz, 1));
  const int groups_per_block = 256 / G;
  int grid = ceil_div(n, groups_per_block);
  if (grid > 65536) grid = 65536;
  if (grid < 1) grid = 1;
  CS_HIP(hipMemsetAsync(C.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
  hipLaunchKernelGGL((spgemm_merge_kernel<T, G, false>), dim3(grid), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(),
                     B.rp(), B.ci(), B.va(), dptr<int>(cursor), C.rp(), (const int*)nullptr, (int*)nullptr,
                     (T*)nullptr);
  check_launch("spgemm symbolic");
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(C.rp(), (int64_t)n + 1, st, dptr<int>(total));
  C.nnz = read_int(dptr<int>(total), st);
  CS_REQUIRE(C.nnz >= 0, CSGPU_BAD_ARGS, "SpGEMM result exceeds 2^31 nonzeros");
  C.col.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(int));
  C.val.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(T));
  hipLaunchKernelGGL((spgemm_merge_kernel<T, G, true>), dim3(grid), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(),
                     B.rp(), B.ci(), B.va(), dptr<int>(cursor), (int*)nullptr, C.rp(), C.ci(), C.va());
  check_launch("spgemm numeric");
  CS_HIP(hipStreamSynchronize(st));  // cursor freed on return
}

// ---- fast path: short rows on both sides (every product of the raster hierarchy) -------------------------------
// Same multiway merge, but the B rows a lane needs are staged ONCE into LDS and the cursors live in registers, so the
// merge loop touches no global memory: lane lg of the row's G-lane group owns A entries ab + lg + G*e (e < EPL) and
// keeps the (at most MAXB) entries of B row A.col[...] in its LDS slot. Output order and summation order are identical
// to spgemm_merge_kernel (ascending A entry within a lane, fixed shuffle tree across lanes).
template <class T, int G, int EPL, int MAXB, bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_lds_kernel(int nrows, const int* __restrict__ Arp,
                                                         const int* __restrict__ Aci, const T* __restrict__ Ava,
                                                         const int* __restrict__ Brp, const int* __restrict__ Bci,
                                                         const T* __restrict__ Bva, int* __restrict__ Ccount,
                                                         const int* __restrict__ Crp, int* __restrict__ Cci,
                                                         T* __restrict__ Cva) {
  __shared__ int s_col[256 * EPL * MAXB];
  __shared__ T s_val[NUMERIC ? 256 * EPL * MAXB : 1];
  const int lg = threadIdx.x % G;
  const int groups_per_block = 256 / G;
  int* mycol = s_col + (size_t)threadIdx.x * EPL * MAXB;
  T* myval = s_val + (NUMERIC ? (size_t)threadIdx.x * EPL * MAXB : 0);
  for (int row = blockIdx.x * groups_per_block + threadIdx.x / G; row < nrows; row += gridDim.x * groups_per_block) {
    const int ab = Arp[row], ae = Arp[row + 1];
    int blen[EPL], bpos[EPL];
    T aval[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int ka = ab + lg + G * e;
      blen[e] = 0;
      bpos[e] = 0;
      aval[e] = T(0);
      if (ka < ae) {
        const int k = Aci[ka];
        const int b0 = Brp[k];
        blen[e] = Brp[k + 1] - b0;
        if (NUMERIC) aval[e] = Ava[ka];
        for (int j = 0; j < blen[e]; ++j) {
          mycol[e * MAXB + j] = Bci[b0 + j];
          if (NUMERIC) myval[e * MAXB + j] = Bva[b0 + j];
        }
      }
    }
    int count = 0;
    const int out0 = NUMERIC ? Crp[row] : 0;
    for (;;) {
      int cmin = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (bpos[e] < blen[e]) {
          const int cc = mycol[e * MAXB + bpos[e]];
          cmin = cc < cmin ? cc : cmin;
        }
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) {
        const int t = __shfl_xor(cmin, o, G);
        cmin = t < cmin ? t : cmin;
      }
      if (cmin == 0x7fffffff) break;
      T sacc = T(0);
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (bpos[e] < blen[e] && mycol[e * MAXB + bpos[e]] == cmin) {
          if (NUMERIC) sacc += aval[e] * myval[e * MAXB + bpos[e]];
          ++bpos[e];
        }
      if (NUMERIC) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, G);
        if (lg == 0) {
          Cci[out0 + count] = cmin;
          Cva[out0 + count] = sacc;
        }
      }
      ++count;
    }
    if (!NUMERIC && lg == 0) Ccount[row] = count;
  }
}

// max row length of a CSR matrix (block partial maxima; finished on the host)
__global__ __launch_bounds__(256) void max_row_len_kernel(int nrows, const int* __restrict__ rp, int* __restrict__ out) {
  int m = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nrows; i += gridDim.x * 256) m = max(m, rp[i + 1] - rp[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

template <class T>
inline int max_row_len(const Csr<T>& A, hipStream_t st) {
  DBuf d = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(d.p, 0, sizeof(int), st));
  if (A.nrows > 0) hipLaunchKernelGGL(max_row_len_kernel, dim3(grid_for(A.nrows)), dim3(256), 0, st, A.nrows, A.rp(), dptr<int>(d));
  return read_int(dptr<int>(d), st);
}

template <class T, int G, int EPL, int MAXB>
inline void spgemm_lds(const Csr<T>& A, const Csr<T>& B, Csr<T>& C, hipStream_t st) {
  const int n = A.nrows;
  C.nrows = n;
  C.ncols = B.ncols;
  C.rowptr.alloc((size_t)(n + 1) * sizeof(int));
  const int groups_per_block = 256 / G;
  int grid = ceil_div(n, groups_per_block);
  if (grid > 65536) grid = 65536;
  if (grid < 1) grid = 1;
  CS_HIP(hipMemsetAsync(C.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
  hipLaunchKernelGGL((spgemm_lds_kernel<T, G, EPL, MAXB, false>), dim3(grid), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(),
                     B.rp(), B.ci(), B.va(), C.rp(), (const int*)nullptr, (int*)nullptr, (T*)nullptr);
  check_launch("spgemm (lds) symbolic");
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(C.rp(), (int64_t)n + 1, st, dptr<int>(total));
  C.nnz = read_int(dptr<int>(total), st);
  CS_REQUIRE(C.nnz >= 0, CSGPU_BAD_ARGS, "SpGEMM result exceeds 2^31 nonzeros");
  C.col.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(int));
  C.val.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(T));
  hipLaunchKernelGGL((spgemm_lds_kernel<T, G, EPL, MAXB, true>), dim3(grid), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(),
                     B.rp(), B.ci(), B.va(), (int*)nullptr, C.rp(), C.ci(), C.va());
  check_launch("spgemm (lds) numeric");
  CS_HIP(hipStreamSynchronize(st));
}

// ---- A * T for a tentative prolongator T with exactly one entry per row (column agg[j], value tva[j]) -------------
// Row i of A*T is row i of A with every column j replaced by agg[j] and equal columns merged: no second matrix to walk,
// so one thread per row with a sorted list of at most MAXL aggregates in registers does it (rows of a raster level touch
// <= 9 columns, i.e. <= 4 aggregates). Entries are added in A's column order (deterministic). FILL = false: counts only.
template <class T, int MAXL, bool FILL>
__global__ __launch_bounds__(256) void spgemm_tentative_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                               const T* __restrict__ va, const int* __restrict__ agg,
                                                               const T* __restrict__ tva, int* __restrict__ count,
                                                               const int* __restrict__ crp, int* __restrict__ cci,
                                                               T* __restrict__ cva) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int cols[MAXL];
    T vals[MAXL];
    int m = 0;
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int j = ci[k];
      if (tva[j] == T(0)) continue;  // a weightless row (cell-space hierarchies) belongs to no aggregate: no entry
      const int a = agg[j];
      const T v = FILL ? va[k] * tva[j] : T(0);
      int q = 0;
      while (q < m && cols[q] < a) ++q;
      if (q < m && cols[q] == a) {
        vals[q] += v;
      } else {
        for (int s2 = m; s2 > q; --s2) {
          cols[s2] = cols[s2 - 1];
          vals[s2] = vals[s2 - 1];
        }
        cols[q] = a;
        vals[q] = v;
        ++m;
      }
    }
    if (!FILL) {
      count[i] = m;
    } else {
      const int o = crp[i];
      for (int q = 0; q < m; ++q) {
        cci[o + q] = cols[q];
        cva[o + q] = vals[q];
      }
    }
  }
}

template <class T>
inline void spgemm(const Csr<T>& A, const Csr<T>& B, Csr<T>& C, hipStream_t st);

// C = A * T (see above); falls back to the general SpGEMM when a row of A is longer than the register list
template <class T>
inline void spgemm_tentative(const Csr<T>& A, const Csr<T>& Tm, const int* agg, Csr<T>& C, hipStream_t st) {
  constexpr int MAXL = 32;  // (rows of a cell-space level hold up to 25 entries)
  if (!knobs().direct_at || max_row_len(A, st) > MAXL) return spgemm(A, Tm, C, st);
  const int n = A.nrows;
  C.nrows = n;
  C.ncols = Tm.ncols;
  C.rowptr.alloc((size_t)(n + 1) * sizeof(int));
  CS_HIP(hipMemsetAsync(C.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
  const int g = grid_for(n);
  hipLaunchKernelGGL((spgemm_tentative_kernel<T, MAXL, false>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), agg,
                     Tm.va(), C.rp(), (const int*)nullptr, (int*)nullptr, (T*)nullptr);
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(C.rp(), (int64_t)n + 1, st, dptr<int>(total));
  C.nnz = read_int(dptr<int>(total), st);
  C.col.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(int));
  C.val.alloc((size_t)std::max<int64_t>(C.nnz, 1) * sizeof(T));
  hipLaunchKernelGGL((spgemm_tentative_kernel<T, MAXL, true>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), agg,
                     Tm.va(), (int*)nullptr, (const int*)C.rp(), C.ci(), C.va());
  check_launch("A * T");
}

template <class T>
inline void spgemm(const Csr<T>& A, const Csr<T>& B, Csr<T>& C, hipStream_t st) {
  // fast path when every row of A fits a lane group and every row of B fits its LDS slot
  const int ma = max_row_len(A, st), mb = max_row_len(B, st);
  if (mb <= 4 && ma <= 16) return spgemm_lds<T, 8, 2, 4>(A, B, C, st);
  if (mb <= 12 && ma <= 16) return spgemm_lds<T, 8, 2, 12>(A, B, C, st);
  if (mb <= 12 && ma <= 64) return spgemm_lds<T, 32, 2, 12>(A, B, C, st);
  const double avg = A.nrows > 0 ? (double)A.nnz / (double)A.nrows : 1.0;
  if (avg <= 3.0)
    spgemm_group<T, 2>(A, B, C, st);
  else if (avg <= 6.0)
    spgemm_group<T, 4>(A, B, C, st);
  else if (avg <= 12.0)
    spgemm_group<T, 8>(A, B, C, st);
  else if (avg <= 24.0)
    spgemm_group<T, 16>(A, B, C, st);
  else if (avg <= 48.0)
    spgemm_group<T, 32>(A, B, C, st);
  else
    spgemm_group<T, 64>(A, B, C, st);
}

// ------------------------------------------------------------------------------------------------ transpose
__global__ __launch_bounds__(256) void col_count_kernel(int64_t nnz, const int* __restrict__ ci,
                                                        int* __restrict__ counts) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256)
    atomicAdd(&counts[ci[k]], 1);
}

template <class T>
__global__ __launch_bounds__(256) void transpose_fill_kernel(int nrows, const int* __restrict__ rp,
                                                             const int* __restrict__ ci, const T* __restrict__ va,
                                                             int* __restrict__ fillpos, int* __restrict__ tci,
                                                             T* __restrict__ tva) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nrows; i += gridDim.x * 256)
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int pos = atomicAdd(&fillpos[ci[k]], 1);
      tci[pos] = i;
      tva[pos] = va[k];
    }
}

// Per-row rank sort (keys unique within a row): entry e goes to rowstart + #{keys in the row smaller than key_e}.
template <class T, int G>
__global__ __launch_bounds__(256) void row_rank_sort_kernel(int nrows, const int* __restrict__ rp,
                                                            const int* __restrict__ ci_in, const T* __restrict__ va_in,
                                                            int* __restrict__ ci_out, T* __restrict__ va_out) {
  const int lg = threadIdx.x % G;
  const int groups_per_block = 256 / G;
  for (int row = blockIdx.x * groups_per_block + threadIdx.x / G; row < nrows; row += gridDim.x * groups_per_block) {
    const int b = rp[row], e = rp[row + 1];
    for (int k = b + lg; k < e; k += G) {
      const int key = ci_in[k];
      int rank = 0;
      for (int m = b; m < e; ++m) rank += ci_in[m] < key ? 1 : 0;
      ci_out[b + rank] = key;
      va_out[b + rank] = va_in[k];
    }
  }
}

template <class T>
inline void transpose(const Csr<T>& A, Csr<T>& At, hipStream_t st) {
  At.nrows = A.ncols;
  At.ncols = A.nrows;
  At.nnz = A.nnz;
  const int m = At.nrows;
  At.rowptr.alloc((size_t)(m + 1) * sizeof(int));
  At.col.alloc((size_t)std::max<int64_t>(A.nnz, 1) * sizeof(int));
  At.val.alloc((size_t)std::max<int64_t>(A.nnz, 1) * sizeof(T));
  CS_HIP(hipMemsetAsync(At.rp(), 0, (size_t)(m + 1) * sizeof(int), st));
  if (A.nnz > 0)
    hipLaunchKernelGGL(col_count_kernel, dim3(grid_for(A.nnz)), dim3(256), 0, st, A.nnz, A.ci(), At.rp());
  exclusive_scan_i32(At.rp(), (int64_t)m + 1, st);
  DBuf fillpos = dalloc<int>((size_t)m + 1);
  CS_HIP(hipMemcpyAsync(fillpos.p, At.rp(), (size_t)(m + 1) * sizeof(int), hipMemcpyDeviceToDevice, st));
  DBuf tci = dalloc<int>((size_t)std::max<int64_t>(A.nnz, 1));
  DBuf tva = dalloc<T>((size_t)std::max<int64_t>(A.nnz, 1));
  hipLaunchKernelGGL((transpose_fill_kernel<T>), dim3(grid_for(A.nrows)), dim3(256), 0, st, A.nrows, A.rp(), A.ci(),
                     A.va(), dptr<int>(fillpos), dptr<int>(tci), dptr<T>(tva));
  int grid = ceil_div(m, 256 / 16);
  if (grid > 65536) grid = 65536;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((row_rank_sort_kernel<T, 16>), dim3(grid), dim3(256), 0, st, m, At.rp(), dptr<int>(tci),
                     dptr<T>(tva), At.ci(), At.va());
  check_launch("transpose");
  CS_HIP(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------------------------------------ prolongator smoothing
// In place on C = A*T:  P_ij = [j == agg(i)] t_i - (omega_p / labs_i) C_ij
template <class T>
__global__ __launch_bounds__(256) void smooth_prolongator_kernel(int n, const int* __restrict__ rp,
                                                                 const int* __restrict__ ci, T* __restrict__ va,
                                                                 const int* __restrict__ agg, const T* __restrict__ tva,
                                                                 const T* __restrict__ labs, double omega_p,
                                                                 int* __restrict__ missing,
                                                                 const long long* __restrict__ size_f = nullptr) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double l = (double)labs[i];
    const double w = l != 0.0 ? omega_p / l : 0.0;
    const int a = agg[i];
    bool found = false;
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      double v = -w * (double)va[k];
      if (ci[k] == a) {
        v += (double)tva[i];
        found = true;
      }
      va[k] = (T)v;
    }
    if (!found && !(size_f && size_f[i] == 0)) atomicAdd(missing, 1);  // (a weightless row has no entry at all)
  }
}

// In place on AP (sorted rows, pattern contains P's):  Q_ij = P_ij - omega * dinv_i * (AP)_ij.
// With r = b - A x the first damped-Jacobi sweep after the coarse-grid correction,
//   (x + P e) + omega D^-1 (b - A (x + P e)) = x + omega D^-1 r + Q e,
// becomes ONE product with Q instead of a product with P followed by a product with A.
template <class T>
__global__ __launch_bounds__(256) void build_q_kernel(int n, const int* __restrict__ qrp, const int* __restrict__ qci,
                                                      T* __restrict__ qva, const int* __restrict__ prp,
                                                      const int* __restrict__ pci, const T* __restrict__ pva,
                                                      const T* __restrict__ dinv, T omega, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const T w = omega * dinv[i];
    int kp = prp[i];
    const int kpe = prp[i + 1];
    for (int k = qrp[i]; k < qrp[i + 1]; ++k) {
      T v = -w * qva[k];
      if (kp < kpe && pci[kp] == qci[k]) {
        v += pva[kp];
        ++kp;
      }
      qva[k] = v;
    }
    if (kp != kpe) atomicAdd(bad, 1);
  }
}

// Two-product form of a V(1,1) level (zero initial guess, damped Jacobi with weight omega):
//   x1 = omega D^-1 b                                   pre-smoothing
//   b_c = R (b - A x1) = (P - omega D^-1 A P)^T b = Q^T b          (A, D symmetric)
//   x_c = coarse solve(b_c)
//   out = (x1 + P x_c) + omega D^-1 (b - A (x1 + P x_c)) = S b + Q x_c,   S = 2 omega D^-1 - omega D^-1 A omega D^-1
// i.e. the whole level is  b_c = Q^T b  and  out = [S Q] [b; x_c]: two products instead of residual + restriction +
// fused prolongation, and no x1 / residual vectors at all. M = [S Q] is stored as one CSR with n + n_c columns
// (Q's columns shifted by n) acting on the vector [b; x_c] (x_c is written right behind b).
template <class T>
__global__ __launch_bounds__(256) void sq_rowptr_kernel(int n, const int* __restrict__ arp, const int* __restrict__ qrp,
                                                        int* __restrict__ mrp) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) mrp[i] = arp[i] + qrp[i];
}

template <class T>
__global__ __launch_bounds__(256) void build_sq_kernel(int n, const int* __restrict__ arp, const int* __restrict__ aci,
                                                       const T* __restrict__ ava, const int* __restrict__ qrp,
                                                       const int* __restrict__ qci, const T* __restrict__ qva,
                                                       const T* __restrict__ dinv, double omega,
                                                       const int* __restrict__ mrp, int* __restrict__ mci,
                                                       T* __restrict__ mva) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    int o = mrp[i];
    const double wi = omega * (double)dinv[i];
    for (int k = arp[i]; k < arp[i + 1]; ++k, ++o) {
      const int j = aci[k];
      double v = -wi * (double)ava[k] * omega * (double)dinv[j];
      if (j == i) v += 2.0 * wi;
      mci[o] = j;
      mva[o] = (T)v;
    }
    for (int k = qrp[i]; k < qrp[i + 1]; ++k, ++o) {
      mci[o] = n + qci[k];
      mva[o] = qva[k];
    }
  }
}

// 1 / diagonal for the Jacobi sweeps, 0 (the sweeps leave the unknown alone) where the diagonal is no pivot:
//  - not positive;
//  - an ISOLATED row (no off-diagonal entry: labs == |diag|) whose diagonal is below `floor_` = 64 eps(T) max|diag|. Such
//    a row is a connected component that has shrunk to ONE coarse unknown: its operator is the regularisation shift
//    (1e-13 of the couplings) in exact arithmetic and rounding noise OF EITHER SIGN in practice (measured on MI355X, fp32
//    hierarchy of a 4-neighbour raster with islands: -1.5e-17 where the CPU emulator, without fused multiply-adds, has an
//    exact 0), while the restricted right-hand side of a consistent system is rounding noise as well: 1 / diagonal
//    would blow that noise up by 1e17. The constant of a floating component is its null space: 0 is the pseudo-inverse.
template <class T>
__global__ __launch_bounds__(256) void dinv_kernel(int n, const T* __restrict__ diag, const T* __restrict__ labs,
                                                   double floor_, T* __restrict__ dinv) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const T d = diag[i];
    const bool isolated = labs[i] <= (d < T(0) ? -d : d);
    dinv[i] = (d > T(0) && !(isolated && (double)d <= floor_)) ? T(1) / d : T(0);
  }
}

// ------------------------------------------------------------------------------------------------ host side
// Near-kernel of the coarsest operator of an fp32 hierarchy. The Galerkin chain keeps the candidate vector
// v = sqrt(number of fine nodes under a coarse node) (tentative_kernel) in the near-kernel of every level: A_c v is the
// regularisation shift only. In fp32 the chain's rounding errors replace that eigenvalue by noise of either sign which
// GROWS relative to lambda_max from level to level (the stiffness shrinks ~10x per level, the errors do not): measured
// 2e-6 lambda_max after four coarsenings, 7e-4 after seven -- sometimes above, sometimes below the pseudo-inverse's
// cutoff, and when the mode is kept its huge gain feeds the rounding noise of the residual back into z (300 x 300
// raster: 14.4 instead of 10.0 iterations, 3e-10 instead of 5e-13 agreement with the fp64 hierarchy; 2000 x 2000:
// 13.8 instead of 10.6 iterations). The candidate is known, so the pseudo-inverse treats as null space every eigenpair
// below `thr` * lambda_max whose eigenvector lies (>= 80 % of its norm) in the span of the candidate restricted to the
// connected components of the coarsest graph. Dropping a direction of the preconditioner costs CG at most one
// iteration; keeping a noisy one costs many. Returns the orthonormal per-component candidates, one per row.
inline std::vector<std::vector<double>> component_candidates(const std::vector<double>& M, int n,
                                                             const std::vector<double>& cand) {
  std::vector<std::vector<double>> out;
  std::vector<int> comp(n, -1), stack;
  for (int s0 = 0; s0 < n; ++s0) {
    if (comp[s0] >= 0) continue;
    std::vector<double> v((size_t)n, 0.0);
    double nrm = 0;
    comp[s0] = s0;
    stack.push_back(s0);
    while (!stack.empty()) {
      const int i = stack.back();
      stack.pop_back();
      v[i] = cand[i];
      nrm += cand[i] * cand[i];
      for (int j = 0; j < n; ++j)
        if (comp[j] < 0 && (M[(size_t)i * n + j] != 0 || M[(size_t)j * n + i] != 0)) {
          comp[j] = s0;
          stack.push_back(j);
        }
    }
    if (!(nrm > 0)) continue;
    const double inv = 1.0 / std::sqrt(nrm);
    for (double& x : v) x *= inv;
    out.push_back(std::move(v));
  }
  return out;
}

// Dense symmetric pseudo-inverse (cyclic Jacobi eigen-solver); n is at most a few hundred.
// `defl_cand` / `defl_out`: a second pseudo-inverse from the same decomposition that leaves out every eigenpair below
// kernel_thr * lambda_max lying along one of `defl_cand` (Dirichlet-masked solves, pcg.h).
inline std::vector<double> dense_sym_pinv(std::vector<double> M, int n, double eps,
                                          const std::vector<std::vector<double>>* kernel_cand = nullptr,
                                          double kernel_thr = 0.0, int* kernel_dropped = nullptr,
                                          const std::vector<std::vector<double>>* defl_cand = nullptr,
                                          std::vector<double>* defl_out = nullptr) {
  std::vector<double> V((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const double s = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
      M[(size_t)i * n + j] = M[(size_t)j * n + i] = s;
    }
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < n; ++i) {
      dg += M[(size_t)i * n + i] * M[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += M[(size_t)i * n + j] * M[(size_t)i * n + j];
    }
    if (off <= 1e-30 * dg || off == 0) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = M[(size_t)p * n + q];
        if (apq == 0) continue;
        const double app = M[(size_t)p * n + p], aqq = M[(size_t)q * n + q];
        const double tau = (aqq - app) / (2 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double mkp = M[(size_t)k * n + p], mkq = M[(size_t)k * n + q];
          M[(size_t)k * n + p] = c * mkp - s * mkq;
          M[(size_t)k * n + q] = s * mkp + c * mkq;
        }
        for (int k = 0; k < n; ++k) {
          const double mpk = M[(size_t)p * n + k], mqk = M[(size_t)q * n + k];
          M[(size_t)p * n + k] = c * mpk - s * mqk;
          M[(size_t)q * n + k] = s * mpk + c * mqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  double smax = 0;
  for (int i = 0; i < n; ++i) smax = std::max(smax, std::fabs(M[(size_t)i * n + i]));
  // eigenvalues below n*eps(T)*lambda_max are treated as the null space (Julia pinv's default rtol)
  double cut = eps * (double)n * smax;
  if (knobs().pinv_cut > 0.0) cut = knobs().pinv_cut * smax;  // experiment knob
  if (knobs().verbose) {
    std::vector<double> ev(n);
    for (int i = 0; i < n; ++i) ev[i] = M[(size_t)i * n + i];
    std::sort(ev.begin(), ev.end());
    fprintf(stderr, "csgpu: coarsest level n=%d eigenvalues/lambda_max: %.3e %.3e %.3e ... cut %.3e\n", n, ev[0] / smax,
            n > 1 ? ev[1] / smax : 0.0, n > 2 ? ev[2] / smax : 0.0, cut / smax);
  }
  std::vector<double> Pinv((size_t)n * n, 0.0);
  std::vector<int> kind((size_t)n, 0);  // 0: null space, 1: near-kernel eigenpair, 2: kept, 3: kept with the gain 1 / cutoff
  double lam_ref = 0;                   // smallest kept eigenvalue
  for (int e = 0; e < n; ++e) {
    const double lam = M[(size_t)e * n + e];
    const bool below = !(std::fabs(lam) > cut);
    if (below && !(kernel_cand && lam > 0)) continue;  // null space (as Julia's pinv treats it)
    kind[e] = 2;
    if (kernel_cand && std::fabs(lam) < kernel_thr * smax) {  // near-kernel eigenpair of an fp32 hierarchy (see above)
      double overlap = 0;
      for (const auto& v : *kernel_cand) {
        double d = 0;
        for (int i = 0; i < n; ++i) d += v[i] * V[(size_t)i * n + e];
        overlap += d * d;
      }
      if (overlap >= 0.8) {
        kind[e] = below ? 0 : 1;
        if (!below && kernel_dropped) ++*kernel_dropped;
      }
    }
    // fp32 hierarchies: a POSITIVE eigenvalue below the cutoff (n eps(fp32) lambda_max ~ 1e-5 lambda_max) that is not the
    // candidate's is a genuine mode of a badly conditioned operator more often than it is noise -- a single-level handle
    // of a component with conductances spread over ten orders of magnitude has several. Dropping it makes the
    // preconditioner singular on a mode the right-hand side excites, and CG then cannot converge (found by fuzzing: 69
    // nodes, sigma = 3.5, relative residual stuck at 2.5). It keeps a bounded gain instead: 1 / cutoff.
    if (kind[e] == 2 && below) kind[e] = 3;
    if (kind[e] == 2 && lam > 0 && (lam_ref == 0 || lam < lam_ref)) lam_ref = lam;
  }
  // A near-kernel eigenpair gets NO gain: the coarsest problem is solved in the orthogonal complement of the candidate.
  // (Giving it the gain of the smoothest kept mode instead -- a non-singular preconditioner -- was measured too,
  // CSGPU_KERNEL_GAIN_REF=1: fine up to 5000^2, but at 10000^2 the fp32 restriction chain of eight levels has put so much
  // spurious weight on the candidate that any gain feeds it back: 13.3 iterations instead of 11.1.)
  const bool kernel_ref = knobs().kernel_gain_ref;  // A/B knob
  if (defl_out) defl_out->assign((size_t)n * n, 0.0);
  for (int e = 0; e < n; ++e) {
    if (kind[e] == 0) continue;
    double inv = kind[e] == 3 ? 1.0 / cut : 1.0 / M[(size_t)e * n + e];
    bool in_defl = defl_out != nullptr && kind[e] != 1;
    if (in_defl && defl_cand && std::fabs(M[(size_t)e * n + e]) < kernel_thr * smax) {
      double overlap = 0;
      for (const auto& v : *defl_cand) {
        double d = 0;
        for (int i = 0; i < n; ++i) d += v[i] * V[(size_t)i * n + e];
        overlap += d * d;
      }
      if (overlap >= 0.8) in_defl = false;
    }
    if (in_defl)
      for (int i = 0; i < n; ++i) {
        const double vi = V[(size_t)i * n + e] * inv;
        for (int j = 0; j < n; ++j) (*defl_out)[(size_t)i * n + j] += vi * V[(size_t)j * n + e];
      }
    if (kind[e] == 1) {
      if (!kernel_ref || !(lam_ref > 0)) continue;
      inv = 1.0 / lam_ref;
    }
    for (int i = 0; i < n; ++i) {
      const double vi = V[(size_t)i * n + e] * inv;
      for (int j = 0; j < n; ++j) Pinv[(size_t)i * n + j] += vi * V[(size_t)j * n + e];
    }
  }
  return Pinv;
}

// ---- strength of connection inside the regular tiles (heterogeneous rasters) ----------------------------------------
// The reference's aggregation treats every coupling as strong (SymmetricStrength, theta = 0) and relies on symmetric
// Gauss-Seidel to cope with conductances that differ by orders of magnitude (src/core.jl:164-167). With Jacobi sweeps
// the 3 x 3 tiles themselves must respect the weak couplings: a cell tied to its tile-mates only through couplings below
// theta * sqrt(a_ii a_jj) (the symmetric measure) does not follow the tile's coarse unknown. The piece analysis written
// for NODATA rasters (tile_pieces_kernel) does exactly the bookkeeping needed -- pieces of a tile, the main piece keeps
// the aggregate, the other cells join the neighbouring tile they are strongly coupled to -- so weak couplings are
// simply not followed there. Measured (log-normal conductances exp(sigma N(0,1)), PCG iterations per pair, oracle =
// the reference's algorithm with its Gauss-Seidel smoother):
//                     sigma = 1        sigma = 2          sigma = 3
//   300^2   before    10.0             22.5               50.3
//           after     (not triggered)  16.0   oracle 15   34.0   oracle 36
//   1000^2  before    10.0             29.8               114
//           after     (not triggered)  22.5   oracle 19   68.3   oracle 61
// theta between 0.03 and 0.1 gives the same counts within one iteration; 0.15 starts to cost on sigma = 1. The filter is
// only used when it matters -- when more than tile_split_min of the cells would leave their tile -- because regular
// tiles are what keeps level 1 a nine-point lattice (lattice_level1_setup). CSGPU_TILE_THETA / CSGPU_TILE_SPLIT_MIN
// override the defaults (theta 0 switches the filter off).
inline double default_tile_theta() {
  return knobs().tile_theta;
}
inline int tile_sample_stride(int64_t ntiles) {  // the heterogeneity test looks at ~65 k tiles
  const int64_t s = ntiles / 65536;
  return (int)std::max<int64_t>(1, s);
}
inline double default_tile_split_min() {
  return knobs().tile_split_min;
}

template <class T>
struct Level {
  Csr<T> A, P, R;       // P, R empty on the coarsest level
  Csr<T> Q;             // Q = P - omega D^-1 A P: prolongation fused with the first post-smoothing sweep
  Csr<T> QT, M;         // level 0 with V(1,1) smoothing only: Q^T and [S Q] of the two-product form (see build_sq_kernel)
  Dia<T> Sdia;          // ... or, when the fine matrix is a raster lattice with regular 3x3 aggregates: S in lattice form
  LatticeQ<T> Ql;       // (stencil.h) and Q in its index-free tile form (lattice.h); Q^T and [S Q] are not built then
  DBuf agg0;            // level 0 aggregate of every node (kept only until Ql has been built)
  Dia<T> Adia;          // level 1 of a raster hierarchy in lattice form (lattice_level1_setup, lattice_setup.h): with it
                        // Sdia = the two-sweep smoother polynomial and Ql = (I - S A) P, and the level runs as four
                        // marching products (vcycle in pcg.h) instead of seven CSR ones
  Dia25<T> A25;         // levels >= 1 of a raster hierarchy with REFINED tiles (cell-space NODATA rasters, strength-aware
                        // tiles): A in its index-free 25-point lattice form (dia25.h); the level's Jacobi sweeps and its
                        // residual march over it instead of going through the CSR SpMM (P, R, Q stay CSR)
  bool lattice_v22() const { return Adia.n > 0 && Sdia.n > 0 && Ql.n > 0; }
  bool lattice_two_product() const { return Adia.n == 0 && Sdia.n > 0 && Ql.n > 0; }
  bool two_product() const { return M.nnz > 0 || lattice_two_product(); }
  DBuf dinv;            // 1/a_ii
  DBuf orderA;          // band-aware row-block traversal order for products with A (may be empty)
  DBuf orderQT;         // traversal order of the long-row kernel on Q^T (two-product level; may be empty)
  long long periodA = 0;  // band period detected on A (0: none)
  double omega = 0;     // damped-Jacobi weight (Chebyshev levels: the weight of the first sweep = the weight Q is built with)
  double lam_max = 0;   // Chebyshev levels (l >= 1): upper end of the targeted eigenvalue interval of D^-1 A
  DBuf cand;            // levels >= 1 with at most 4096 rows: the candidate vector sqrt(fine nodes under the node) (tail.h)
  std::vector<double> weights;  // Chebyshev levels: one Jacobi weight per sweep (empty: damped Jacobi with `omega`)
  double rho = 0;       // Gershgorin bound on rho(D^-1 A)
  int n = 0;
  // solve-phase work vectors (allocated for a batch width K on demand)
  DBuf xa, rb, b, qs;
};

// Spectral enrichment of the level-0 coarse space (enrich.h): a second coarse function on the aggregates of a perforated /
// refined-tile raster lattice whose local Fiedler value is small, handled multiplicatively around the V-cycle.
struct Enrich {
  int nvec = 0, nmem = 0, nhalo = 0, R = 0, ntiles = 0;
  int64_t n = 0;
  int phi_bytes = 0;        // precision of vphi (= the hierarchy's)
  DBuf vptr, vcell, vphi;   // the vectors: members of every vector (cell ids in window order) and their values
  DBuf vhalo;               // position of every member's cell in hcell
  DBuf binv;                // [nvec] double: 1 / (G_vv + sum_w |G_vw|)
  DBuf aptr, acell, acoef;  // A E by columns: (cell, coefficient) per vector
  DBuf hcell;               // the non-empty rows of A E (members and their coupled neighbours), ascending cell ids
  DBuf hptr, hvec, hcoef;   // A E by rows: (vector, coefficient) per halo cell
  DBuf t, c, c2, sbuf, save, ppart;  // work: [nvec][K] doubles x 3, [nhalo][K] (A E c) and saved residual entries, block partials
  int work_k = 0, work_bytes = 0;
  // W = Q^T A E by coarse rows (enrich_coarse_setup): the correction of a restriction that ran on the residual BEFORE the
  // pre-pass changed it (fused residual update + restriction, lattice.h): b_c -= W c
  int ntouch = 0;           // coarse nodes with a halo cell in their 3 x 3 block of tiles
  DBuf tcell, tptr;         // [ntouch] coarse node, [ntouch + 1] first entry
  DBuf th, tw;              // per entry: vector, W[coarse node, vector] (double)
  size_t device_bytes() const {
    return vptr.bytes + vcell.bytes + vphi.bytes + vhalo.bytes + binv.bytes + aptr.bytes + acell.bytes + acoef.bytes +
           hcell.bytes + hptr.bytes + hvec.bytes + hcoef.bytes + t.bytes + c.bytes + c2.bytes + sbuf.bytes + save.bytes + ppart.bytes +
           tcell.bytes + tptr.bytes + th.bytes + tw.bytes;
  }
};

static const int kMaxDirComp = 256;  // components of the coarsest graph the Dirichlet correction handles (pcg.h)

template <class T>
struct Hierarchy {
  std::vector<Level<T>> levels;
  DBuf coarse_inv;      // dense pseudo-inverse of the last level's A (n_c x n_c), empty if too large
  int coarse_n = 0;
  bool coarse_dense = false;
  double setup_ms = 0;
  int virtual_rhs_solves = 0;       // ... of which the batch's right-hand side was never stored (Knobs::sparse_init)
  int fused_restrict_solves = 0;    // pcg_solve calls that ran the fused residual update + restriction (csgpu_info)
  bool expander_probe_hit = false;  // the set-up skipped the aggregation of level 0: the expansion probe predicted the bail-out
  int work_k = 0;       // batch width the work vectors are laid out for
  int work_kcap = 0;    // batch width they were allocated for (>= work_k)
  // coarse tail (tail.h): first level run inside the single-launch tail kernel (-1: none, -2: not decided yet), the
  // per-column scratch area and the batch width it is allocated for
  bool near_singular = false;  // the coarsest operator's near-kernel eigenpair was dropped (fp32 hierarchy of a Laplacian)
  // heterogeneity of the raster as the strength test of level 0 measured it (TileStrength / lattice_level0_setup):
  // fraction of the cells that leave their 3x3 tile because of the strength filter (-1: the test did not run). The C
  // API uses it to give strongly heterogeneous rasters an fp64 hierarchy (csgpu.hip, hetero_wants_fp64)
  double hetero_frac = -1.0;
  // Dirichlet-masked solves on this hierarchy (pcg.h, DirichletCoarse): the pseudo-inverse WITHOUT the near-kernel
  // eigenpairs, the coarsest level's candidate, the connected component of the coarsest graph every coarse node lies in
  // (-1: a weightless row of a cell-space hierarchy) and their number (0: the correction is not available)
  DBuf coarse_inv_defl;
  DBuf coarse_cand;
  DBuf coarse_comp;
  int dir_ncomp = 0;
  // set for the duration of a Dirichlet-masked solve: coef[k * kMaxK + c] = 1 / G of component k for column c (device);
  // mode 1: the coarsest solve applies the correction, 2: it WRITES coef from the restricted probe vector instead
  double* dir_coef = nullptr;
  int dir_mode = 0;
  double cand_norm2 = 0;       // |candidate|^2 (= number of fine nodes: the same on every level)
  Enrich enr;                  // level 0 of a cell-space / refined-tile raster lattice (enrich.h); nvec == 0: none
  int tail_first = -2;
  DBuf tail_ws;
  int64_t tail_stride = 0;
  int tail_k = 0;
};

struct SetupParams {
  int max_levels = 16;
  int max_coarse = 100;
  int aggregation = CSGPU_AGG_AUTO;
  double theta = 0.0;
  double omega_p = 1.6;
  double omega_s = 1.5;
  bool two_product = false;  // build Q^T and [S Q] on level 0 (the solve phase runs V(1,1) there)
  int grid_rows = 0, grid_cols = 0;  // extent of the raster the node coordinates refer to (0 = unknown)
  bool coarse_chebyshev = true;  // levels >= 1 smooth with a Chebyshev polynomial in D^-1 A (degree = sweeps) instead of
                                 // damped Jacobi; level 0 keeps Jacobi (its V(1,1) form collapses into two products)
  int nu_l1 = 2, nu_deep = 3;    // sweeps (= polynomial degree) on level 1 / on the levels below it
  bool lattice_s = false;    // the caller builds the lattice forms of the two-product level (Level::Sdia, Level::Ql) from
                             // the aggregates kept in Level::agg0; Q^T and [S Q] are not built here
  // Cell-space rasters (csgpu.hip): weight of every level-0 row in its aggregate (device, long long, 1 = a real node,
  // 0 = the isolated row of a NODATA cell) and the number of real nodes; null / 0 = every row is a node
  const long long* size0 = nullptr;
  int64_t n_real = 0;
  // Heterogeneous rasters (tile_strength in amg_setup.h): couplings weaker than tile_theta * sqrt(a_ii a_jj) do not hold
  // a 3 x 3 tile together; used when more than tile_split_min of the cells leave their tile because of it
  double tile_theta = default_tile_theta();
  double tile_split_min = default_tile_split_min();
};

// Q^T (with its traversal order) of the two-product form in CSR
template <class T>
inline void build_qt_matrix(Level<T>& L, hipStream_t st) {
  transpose(L.Q, L.QT, st);
  spmv_block_order_rect(L.QT, L.periodA, L.orderQT, st);
}

// [S Q] of the two-product form as one CSR matrix with n + n_c columns (build_sq_kernel)
template <class T>
inline void build_sq_matrix(Level<T>& L, hipStream_t st) {
  const int n = L.A.nrows, nagg = L.Q.ncols;
  Csr<T>& M = L.M;
  M.nrows = n;
  M.ncols = n + nagg;
  M.nnz = L.A.nnz + L.Q.nnz;
  M.rowptr.alloc((size_t)(n + 1) * sizeof(int));
  M.col.alloc((size_t)M.nnz * sizeof(int));
  M.val.alloc((size_t)M.nnz * sizeof(T));
  hipLaunchKernelGGL((sq_rowptr_kernel<T>), dim3(grid_for((int64_t)n + 1)), dim3(256), 0, st, n, L.A.rp(), L.Q.rp(),
                     M.rp());
  hipLaunchKernelGGL((build_sq_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, n, L.A.rp(), L.A.ci(), L.A.va(), L.Q.rp(),
                     L.Q.ci(), L.Q.va(), dptr<T>(L.dinv), L.omega, M.rp(), M.ci(), M.va());
  check_launch("build [S Q]");
}

// Regular 3x3 tiles of a full raster (every cell of a gridR x gridC raster is a node, coordinates known): exactly the
// aggregates the MIS(2) rounds + the two attachment passes below produce on such a level (tile centres win the first
// round, their neighbours join them, left-over cells join their own tile) -- written down directly, in one pass.
__global__ __launch_bounds__(256) void tile_aggregate_kernel(int n, const int* __restrict__ nrow,
                                                             const int* __restrict__ ncol, int Rc, int Cc,
                                                             int* __restrict__ agg, int* __restrict__ crow,
                                                             int* __restrict__ ccol) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int I = min(nrow[i] / 3, Rc - 1), J = min(ncol[i] / 3, Cc - 1);
    const int a = J * Rc + I;
    agg[i] = a;
    if (nrow[i] == 3 * I && ncol[i] == 3 * J) {  // one writer per tile
      crow[a] = I;
      ccol[a] = J;
    }
  }
}

// ---- regular tiles on a cell-space raster (rows of NODATA cells carry weight 0) -------------------------------------
// A 3x3 tile may hold cells that are NOT connected inside the tile (two banks of a NODATA line, the rims of two windows
// of a stacked raster, ...). One aggregate over disconnected pieces gives a coarse basis function that cannot tell the
// pieces apart -- and couples connected components the graph does not couple. So per tile (one thread, <= 4 x 4 cells):
//   pass 1: label the pieces the tile's cells form under the couplings stored in A; the largest piece (ties: the one
//           holding the smallest cell id) is the tile's MAIN piece and keeps the tile's aggregate;
//   pass 2: every cell of another piece joins the aggregate of the neighbouring tile whose main piece it is most strongly
//           coupled to DIRECTLY;
//   pass 3: what is left over joins the aggregate of an attached neighbour (tile_orphans_kernel below).
// An aggregate thus reaches exactly one cell into the next tile: a row of Q = P - w D^-1 A P sees the aggregates of the
// cells within distance 2, each of which holds a cell within distance 3 of the row's cell -- inside the 3 x 3 block of
// tiles around the row's own tile, so the index-free form (lattice.h) survives. (Moving a piece as a whole does not:
// its far end can be three cells from the tile it joins.)
constexpr int kPieceAttached = 100;  // piece[] value of a cell attached to a neighbouring tile's aggregate (+ round of pass 3)
constexpr int kOrphanRounds = 3;

__device__ __forceinline__ void tile_extent(int t, int nt, int len, int& lo, int& hi) {
  lo = 3 * t;
  hi = t == nt - 1 ? len : 3 * t + 3;
}

// full_connected: a tile whose rows all carry weight is connected without looking (true on the raster's own level, where
// adjacent valid cells are always coupled; NOT on coarser levels, where two non-empty tiles next to each other may lie on
// the two sides of a NODATA line)
template <class T, int PASS>
__global__ __launch_bounds__(256) void tile_pieces_kernel(int R, int C, int Rc, int Cc, const int* __restrict__ rp,
                                                          const int* __restrict__ ci, const T* __restrict__ va,
                                                          long long* __restrict__ size_f, signed char* __restrict__ piece,
                                                          signed char* __restrict__ mainlab, int* __restrict__ agg,
                                                          int full_connected, const T* __restrict__ diag, double theta2,
                                                          int stride) {
  const int ntiles = Rc * Cc;
  // (stride > 1: every stride-th tile only -- the sampled heterogeneity test of TileStrength)
  for (int64_t tl = ((int64_t)blockIdx.x * 256 + threadIdx.x) * stride; tl < ntiles; tl += (int64_t)gridDim.x * 256 * stride) {
    const int tile = (int)tl;
    const int I = tile % Rc, J = tile / Rc;
    int r0, r1, c0, c1;
    tile_extent(I, Rc, R, r0, r1);
    tile_extent(J, Cc, C, c0, c1);
    const int h = r1 - r0, w = c1 - c0;  // <= 4 each
    if (PASS == 1) {
      int lab[16];
      int nvalid = 0;
      for (int kc = 0; kc < w; ++kc)
        for (int kr = 0; kr < h; ++kr) {
          const int64_t cell = (int64_t)(c0 + kc) * R + r0 + kr;
          const bool valid = size_f[cell] != 0;
          lab[kc * h + kr] = valid ? kc * h + kr : -1;
          nvalid += valid ? 1 : 0;
        }
      if (nvalid > 0 && (nvalid < h * w || !full_connected || theta2 > 0.0)) {
        for (int sweep = 0; sweep < 16; ++sweep) {
          bool changed = false;
          for (int kc = 0; kc < w; ++kc)
            for (int kr = 0; kr < h; ++kr) {
              int& me = lab[kc * h + kr];
              if (me < 0) continue;
              const int64_t cell = (int64_t)(c0 + kc) * R + r0 + kr;
              for (int e = rp[cell]; e < rp[cell + 1]; ++e) {
                const int nb = ci[e];
                if (nb == cell || va[e] == T(0)) continue;
                if (theta2 > 0.0 && (double)va[e] * (double)va[e] < theta2 * (double)diag[cell] * (double)diag[nb]) continue;
                const int ni = nb % R - r0, nj = nb / R - c0;
                if (ni < 0 || ni >= h || nj < 0 || nj >= w) continue;
                const int l2 = lab[nj * h + ni];
                if (l2 >= 0 && l2 < me) {
                  me = l2;
                  changed = true;
                }
              }
            }
          if (!changed) break;
        }
      } else if (nvalid == h * w) {
        for (int k = 0; k < h * w; ++k) lab[k] = 0;
      }
      // main piece: most cells, ties to the smaller label
      int best = -1, bestcnt = 0;
      for (int q = 0; q < h * w; ++q) {
        int cnt = 0;
        for (int k = 0; k < h * w; ++k) cnt += lab[k] == q ? 1 : 0;
        if (cnt > bestcnt) {
          bestcnt = cnt;
          best = q;
        }
      }
      mainlab[tile] = (signed char)best;
      for (int kc = 0; kc < w; ++kc)
        for (int kr = 0; kr < h; ++kr) piece[(int64_t)(c0 + kc) * R + r0 + kr] = (signed char)lab[kc * h + kr];
    } else {
      // every cell outside the tile's main piece joins the tile of the main-piece cell it is most strongly coupled to
      // (ties: the first in CSR order); without such a coupling it weighs 0
      const int mainq = mainlab[tile];
      for (int kc = 0; kc < w; ++kc)
        for (int kr = 0; kr < h; ++kr) {
          const int64_t cell = (int64_t)(c0 + kc) * R + r0 + kr;
          if (piece[cell] < 0 || piece[cell] == mainq) continue;
          int target = -1;
          double best = 0.0;
          for (int e = rp[cell]; e < rp[cell + 1]; ++e) {
            const int nb = ci[e];
            if (nb == cell || va[e] == T(0)) continue;
            if (theta2 > 0.0 && (double)va[e] * (double)va[e] < theta2 * (double)diag[cell] * (double)diag[nb]) continue;
            const int ni = nb % R, nj = nb / R;
            if (ni >= r0 && ni < r1 && nj >= c0 && nj < c1) continue;  // inside this tile
            const int nt = min(nj / 3, Cc - 1) * Rc + min(ni / 3, Rc - 1);
            if (piece[nb] < 0 || piece[nb] != mainlab[nt]) continue;
            const double a = fabs((double)va[e]);
            if (a > best) {
              best = a;
              target = nt;
            }
          }
          if (target >= 0) {
            agg[cell] = target;
            piece[cell] = kPieceAttached;  // (never equal to a main label: other tiles' threads still read it as "not main")
          }
        }
    }
  }
}

// pass 3, rounds 1 .. kOrphanRounds: a cell pass 2 left over (an ORPHAN: outside its tile's main piece, not coupled to
// the main piece of any neighbouring tile) adopts the aggregate of the attached cell it is most strongly coupled to --
// attached by pass 2 or by an earlier round, so the result does not depend on the order the threads run in. After the
// last round an orphan without any coupling (an island of one cell) weighs 0; a COUPLED one keeps its tile's aggregate:
// a coupled row without weight is a row on which the coarse space cannot represent the constant, and every such row
// costs the hierarchy its near-kernel (measured: ONE of them among 76 697 nodes lifts the smallest eigenvalue of the
// level-1 operator from 6e-13 to 2e-7 of the largest, and a 1000 x 1000 raster with 15 % NODATA then needs 27 iterations
// for distant pairs instead of 13).
template <class T>
__global__ __launch_bounds__(256) void tile_orphans_kernel(int n, int R, int Rc, int Cc, const int* __restrict__ rp,
                                                           const int* __restrict__ ci, const T* __restrict__ va,
                                                           long long* __restrict__ size_f, signed char* __restrict__ piece,
                                                           const signed char* __restrict__ mainlab, int* __restrict__ agg,
                                                           int round, int last, int C, int constrain,
                                                           const T* __restrict__ diag, double theta2) {
  for (int cell = blockIdx.x * 256 + threadIdx.x; cell < n; cell += gridDim.x * 256) {
    const int pc = piece[cell];
    if (pc < 0 || pc >= kPieceAttached) continue;
    const int i = cell % R, j = cell / R;
    const int I = min(i / 3, Rc - 1), J = min(j / 3, Cc - 1);
    if (pc == mainlab[J * Rc + I]) continue;
    // constrain (the raster's own level, whose Q is kept index-free): the adopted tile must lie within one tile of every
    // row within two cells of this one (dia_orphans_kernel in lattice_setup.h is the same rule on the lattice form)
    int Ilo = 0, Ihi = Rc - 1, Jlo = 0, Jhi = Cc - 1;
    if (constrain) {
      int r0, r1, c0, c1;
      tile_extent(I, Rc, R, r0, r1);
      tile_extent(J, Cc, C, c0, c1);
      const bool up = i - r0 <= 1 && I > 0, dn = r1 - 1 - i <= 1 && I < Rc - 1;
      const bool lf = j - c0 <= 1 && J > 0, rt = c1 - 1 - j <= 1 && J < Cc - 1;
      Ilo = dn ? I : I - 1;
      Ihi = up ? I : I + 1;
      Jlo = rt ? J : J - 1;
      Jhi = lf ? J : J + 1;
    }
    int target = -1;
    double best = 0.0;
    bool coupled = false;
    for (int e = rp[cell]; e < rp[cell + 1]; ++e) {
      const int nb = ci[e];
      if (nb == cell || va[e] == T(0)) continue;
      coupled = true;
      if (theta2 > 0.0 && (double)va[e] * (double)va[e] < theta2 * (double)diag[cell] * (double)diag[nb]) continue;
      const int pn = piece[nb];
      if (pn < kPieceAttached || pn >= kPieceAttached + round) continue;
      const int ta = agg[nb], tI = ta % Rc, tJ = ta / Rc;
      if (tI < Ilo || tI > Ihi || tJ < Jlo || tJ > Jhi) continue;
      const double a = fabs((double)va[e]);
      if (a > best) {
        best = a;
        target = ta;
      }
    }
    if (target >= 0) {
      agg[cell] = target;
      piece[cell] = (signed char)(kPieceAttached + round);
    } else if (last && !coupled) {
      size_f[cell] = 0;
    }
  }
}

// cnt[0] += cells with weight, cnt[1] += those outside their tile's main piece (after pass 1 of the piece analysis)
__global__ __launch_bounds__(256) void piece_count_kernel(int64_t n, int R, int Rc, int Cc, const signed char* __restrict__ piece,
                                                          const signed char* __restrict__ mainlab, int* __restrict__ cnt,
                                                          int stride) {
  int valid = 0, out = 0;
  for (int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x; cell < n; cell += (int64_t)gridDim.x * 256) {
    const int I = min((int)(cell % R) / 3, Rc - 1), J = min((int)(cell / R) / 3, Cc - 1);
    const int64_t tile = (int64_t)J * Rc + I;
    if (tile % stride != 0) continue;  // (only the sampled tiles hold labels of this pass)
    const int pc = piece[cell];
    if (pc < 0) continue;
    ++valid;
    out += pc != mainlab[tile] ? 1 : 0;
  }
  if (valid) atomicAdd(&cnt[0], valid);
  if (out) atomicAdd(&cnt[1], out);
}

inline void piece_counts(int64_t n, int R, int Rc, int Cc, const DBuf& piece, const DBuf& mainlab, int64_t& valid, int64_t& out,
                         hipStream_t st, int stride = 1) {
  DBuf cnt = dalloc<int>(2);
  CS_HIP(hipMemsetAsync(cnt.p, 0, 2 * sizeof(int), st));
  hipLaunchKernelGGL(piece_count_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, R, Rc, Cc, (const signed char*)piece.p,
                     (const signed char*)mainlab.p, dptr<int>(cnt), stride);
  int h[2];
  CS_HIP(hipMemcpyAsync(h, cnt.p, sizeof(h), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  valid = h[0];
  out = h[1];
}

// How the strength filter is decided for a hierarchy (level 0) and handed down the levels
struct TileStrength {
  double hetero_frac = -1.0;  // out (decide): fraction of the cells the filter moves out of their tile (-1: not measured)
  double theta = 0.0;       // in: the threshold to try (0 = none). out (decide): the threshold in effect
  double split_min = 0.0;   // decide: use theta only if more than this fraction of the weighted cells leaves its tile
  bool decide = false;      // level 0: run the test; deeper levels just apply theta
  bool theta_tried = false;   // (internal) the sampled pass ran with the filter
  bool unit_weights = false;  // size_f was allocated as all ones for this test only (an all-valid raster): without the
                              // filter the piece analysis has nothing to do
};

// Aggregate the nodes of A. Returns nagg; fills agg (n ints) and, when coordinates are tracked, the coarse ones.
// size_f (may be null): weight of every row (cell-space rasters: 0 for the rows of NODATA cells / empty tiles). With
// weights the regular tiles are refined by the piece analysis above (which may zero further weights) -- on EVERY level:
// the tiles of level l are the rows of level l + 1 in column-major order again, so a cell-space hierarchy has the level
// sizes of the all-valid raster's, empty tiles riding along as weightless rows (a diagonal entry, nothing else).
// (Measured on MI355X, 15 % random NODATA: MIS(2) on the levels below the raster's own made the iteration count grow
// with the raster, 17 at 3000^2 and 32 at 10000^2; profiles/r3_nodata_*.) In the MIS(2) path weightless rows seed and
// join nothing.
// ---- expander probe ----------------------------------------------------------------------------------------------------
// The coarsening loop stops when the smoothed prolongator is as dense as the matrix (nnz(P) > 0.75 nnz(A): a graph without
// locality, whose Galerkin operator would be dense while the graph is well conditioned to begin with) -- but finds out only
// AFTER the MIS(2) aggregation, which on exactly those graphs is all random gathers: 44 propagation launches of 3.9 ms =
// 172 of the 200 ms of device set-up on BASELINE configs[4]'s network (profiles/r6_network_kernel_stats_a.csv). The outcome is
// predictable from the growth of the 2-hop balls: an MIS(2) aggregate holds about deg + 1 nodes, so the neighbours of a node
// fall into about |B2| / (deg + 1) aggregates (B2 = the node, its neighbours and theirs) and nnz(P) / nnz(A) is about
// mean|B2| / mean((deg + 1)^2): 0.31 on an 8-neighbour raster (measured nnz(P) / nnz(A): 0.3), 0.34 on a geometric network of
// mean degree 10, 0.52 / 0.63 on a 4-neighbour lattice / a binary tree, 0.92 on an Erdos-Renyi graph of mean degree 21. One
// wavefront per sampled row counts |B2| with an LDS hash set (rows and neighbour lists capped at 64 entries).
static const int kProbeSamples = 2048, kProbeTable = 2048, kProbeCap = 1400;
__global__ __launch_bounds__(64) void expansion_probe_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                             unsigned long long* __restrict__ out) {
  __shared__ int tab[kProbeTable];
  __shared__ int cnt;
  const int lane = threadIdx.x;
  const int i = (int)(((unsigned long long)blockIdx.x * 2654435761ull + 12345ull) % (unsigned long long)n);
  for (int e = lane; e < kProbeTable; e += 64) tab[e] = -1;
  if (lane == 0) cnt = 0;
  __syncthreads();
  auto insert = [&](int v) {
    unsigned h = ((unsigned)v * 2654435761u) & (kProbeTable - 1);
    for (int probe = 0; probe < kProbeTable; ++probe) {
      const int old = atomicCAS(&tab[h], -1, v);
      if (old == -1) {
        atomicAdd(&cnt, 1);
        return;
      }
      if (old == v) return;
      h = (h + 1) & (kProbeTable - 1);
    }
  };
  const int b0 = rp[i], deg = min(rp[i + 1] - b0, 64);
  if (lane == 0) insert(i);
  if (lane < deg) insert(ci[b0 + lane]);
  __syncthreads();
  for (int q = 0; q < deg; ++q) {
    if (cnt >= kProbeCap) break;   // (block-uniform: read after the barrier below)
    const int j = ci[b0 + q];
    const int c0 = rp[j], dj = min(rp[j + 1] - c0, 64);
    if (lane < dj) insert(ci[c0 + lane]);
    __syncthreads();
  }
  if (lane == 0) {
    const int own = rp[i + 1] - b0;   // stored entries of the row (the diagonal included: deg + 1)
    atomicAdd(&out[0], (unsigned long long)min(cnt, kProbeCap));
    atomicAdd(&out[1], (unsigned long long)own * (unsigned long long)own);
    atomicAdd(&out[2], (unsigned long long)own);
  }
}

// estimate of nnz(P) / nnz(A) after an MIS(2) aggregation of A (see above); `mean_row` receives the mean row length
template <class T>
inline double expansion_estimate(const Csr<T>& A, hipStream_t st, double* mean_row) {
  DBuf out = dalloc<unsigned long long>(3);
  CS_HIP(hipMemsetAsync(out.p, 0, out.bytes, st));
  hipLaunchKernelGGL(expansion_probe_kernel, dim3(kProbeSamples), dim3(64), 0, st, A.nrows, A.rp(), A.ci(),
                     dptr<unsigned long long>(out));
  unsigned long long h[3];
  CS_HIP(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  check_launch("expansion probe");
  if (mean_row) *mean_row = (double)h[2] / kProbeSamples;
  return h[1] > 0 ? (double)h[0] / (double)h[1] : 0.0;
}

template <class T>
inline int aggregate(const Csr<T>& A, const T* diag, double theta, const int* nrow, const int* ncol, DBuf& agg,
                     DBuf& crow, DBuf& ccol, hipStream_t st, int gridR = 0, int gridC = 0, long long* size_f = nullptr,
                     bool cell_level = false, TileStrength* ts = nullptr) {
  const int n = A.nrows;
  const double theta2 = theta * theta;
  const bool no_direct_tiles = !knobs().direct_tiles;  // A/B knob
  if (!no_direct_tiles && theta == 0.0 && nrow && gridR >= 6 && gridC >= 6 && (int64_t)gridR * gridC == n) {
    const int Rc = (gridR + 1) / 3, Cc = (gridC + 1) / 3;
    agg.alloc((size_t)n * sizeof(int));
    crow.alloc((size_t)Rc * Cc * sizeof(int));
    ccol.alloc((size_t)Rc * Cc * sizeof(int));
    hipLaunchKernelGGL(tile_aggregate_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, nrow, ncol, Rc, Cc, dptr<int>(agg),
                       dptr<int>(crow), dptr<int>(ccol));
    const bool no_pieces = !knobs().tile_pieces;  // A/B knob
    if (size_f && !no_pieces) {
      DBuf piece((size_t)n), mainlab((size_t)Rc * Cc);
      const int gt = grid_for((int64_t)Rc * Cc);
      auto pass1 = [&](double t2, int stride) {
        hipLaunchKernelGGL((tile_pieces_kernel<T, 1>), dim3(grid_for(ceil_div((int64_t)Rc * Cc, stride))), dim3(256), 0, st,
                           gridR, gridC, Rc, Cc, A.rp(), A.ci(), A.va(), size_f, (signed char*)piece.p, (signed char*)mainlab.p,
                           dptr<int>(agg), cell_level ? 1 : 0, diag, t2, stride);
      };
      double th2 = ts ? ts->theta * ts->theta : 0.0;
      if (ts && ts->decide) {
        // heterogeneity test (TileStrength): cells that leave their tile's main piece BECAUSE of the filter, counted on
        // a sample of the tiles (every stride-th; ~65 k tiles: the full analysis costs 50 ms at 1e8 cells)
        const int stride = tile_sample_stride((int64_t)Rc * Cc);
        int64_t valid = 0, out0 = 0, out1 = 0;
        bool ran_plain = false;
        if (th2 > 0.0) {
          if (!ts->unit_weights) {
            pass1(0.0, 1);
            ran_plain = true;
            piece_counts(n, gridR, Rc, Cc, piece, mainlab, valid, out0, st, stride);
          }
          pass1(th2, stride);
          ts->theta_tried = true;
          piece_counts(n, gridR, Rc, Cc, piece, mainlab, valid, out1, st, stride);
        }
        const bool hetero = th2 > 0.0 && (double)(out1 - out0) > ts->split_min * (double)std::max<int64_t>(valid, 1);
        if (th2 > 0.0) ts->hetero_frac = (double)(out1 - out0) / (double)std::max<int64_t>(valid, 1);
        if (knobs().verbose)
          fprintf(stderr, "csgpu: tile strength test: %lld of %lld cells leave their tile at theta %.3g (%lld without): %s\n",
                  (long long)out1, (long long)valid, ts->theta, (long long)out0, hetero ? "filter ON" : "filter off");
        if (!hetero) {
          ts->theta = 0.0;
          th2 = 0.0;
          if (ts->unit_weights) {  // regular tiles, nothing to analyse
            check_launch("tile aggregation");
            return Rc * Cc;
          }
          if (!ran_plain)
            pass1(0.0, 1);
          else if (ts->theta_tried)
            pass1(0.0, stride);  // (the sampled tiles hold the labels of the filtered pass)
        } else {
          pass1(th2, 1);
        }
      } else {
        pass1(th2, 1);
      }
      hipLaunchKernelGGL((tile_pieces_kernel<T, 2>), dim3(gt), dim3(256), 0, st, gridR, gridC, Rc, Cc, A.rp(), A.ci(), A.va(),
                         size_f, (signed char*)piece.p, (signed char*)mainlab.p, dptr<int>(agg), cell_level ? 1 : 0, diag, th2, 1);
      for (int round = 1; round <= kOrphanRounds; ++round)
        hipLaunchKernelGGL((tile_orphans_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, n, gridR, Rc, Cc, A.rp(), A.ci(),
                           A.va(), size_f, (signed char*)piece.p, (const signed char*)mainlab.p, dptr<int>(agg), round,
                           round == kOrphanRounds ? 1 : 0, gridC, cell_level ? 1 : 0, diag, th2);
      check_launch("tile pieces");
      CS_HIP(hipStreamSynchronize(st));  // piece / mainlab are released on return
    }
    check_launch("tile aggregation");
    return Rc * Cc;
  }
  if (ts && ts->decide) ts->theta = 0.0;  // (no regular tiles here: nothing for the strength filter to refine)
  DBuf key = dalloc<unsigned long long>(n), k1 = dalloc<unsigned long long>(n), k2 = dalloc<unsigned long long>(n);
  DBuf counter = dalloc<int>(1);
  const int g = grid_for(n);
  hipLaunchKernelGGL(mis_init_kernel, dim3(g), dim3(256), 0, st, n, dptr<unsigned long long>(key), nrow, ncol,
                     (const long long*)size_f);
  for (int round = 0; round < 1000; ++round) {
    CS_HIP(hipMemsetAsync(counter.p, 0, sizeof(int), st));
    hipLaunchKernelGGL((mis_prop_kernel<T>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), diag, theta2,
                       dptr<unsigned long long>(key), dptr<unsigned long long>(k1));
    hipLaunchKernelGGL((mis_prop_kernel<T>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), diag, theta2,
                       dptr<unsigned long long>(k1), dptr<unsigned long long>(k2));
    hipLaunchKernelGGL(mis_decide_kernel, dim3(g), dim3(256), 0, st, n, dptr<unsigned long long>(key),
                       dptr<unsigned long long>(k2), dptr<int>(counter));
    check_launch("mis round");
    if (read_int(dptr<int>(counter), st) == 0) break;
    CS_REQUIRE(round < 999, CSGPU_INTERNAL, "MIS(2) aggregation did not terminate");
  }
  DBuf root_id = dalloc<int>((size_t)n + 1);
  CS_HIP(hipMemsetAsync(root_id.p, 0, ((size_t)n + 1) * sizeof(int), st));
  hipLaunchKernelGGL(mis_roots_kernel, dim3(g), dim3(256), 0, st, n, dptr<unsigned long long>(key), dptr<int>(root_id));
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(dptr<int>(root_id), (int64_t)n + 1, st, dptr<int>(total));
  int nagg = read_int(dptr<int>(total), st);
  DBuf agg1 = dalloc<int>(n);
  agg.alloc((size_t)n * sizeof(int));
  DBuf orphan = dalloc<int>((size_t)n + 1);
  CS_HIP(hipMemsetAsync(orphan.p, 0, ((size_t)n + 1) * sizeof(int), st));
  hipLaunchKernelGGL((agg_pass1_kernel<T>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), diag, theta2,
                     dptr<unsigned long long>(key), dptr<int>(root_id), dptr<int>(agg1));
  hipLaunchKernelGGL((agg_pass2_kernel<T>), dim3(g), dim3(256), 0, st, n, A.rp(), A.ci(), A.va(), diag, theta2,
                     dptr<int>(agg1), dptr<int>(agg), dptr<int>(orphan), nrow, ncol, gridR, gridC, (const long long*)size_f);
  // nodes that could not be attached (cannot happen for a symmetric strength graph; kept as a safety net)
  DBuf orphan_flag = dalloc<int>((size_t)n + 1);
  CS_HIP(hipMemcpyAsync(orphan_flag.p, orphan.p, ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToDevice, st));
  exclusive_scan_i32(dptr<int>(orphan), (int64_t)n + 1, st, dptr<int>(total));
  const int norph = read_int(dptr<int>(total), st);
  if (norph > 0) {
    hipLaunchKernelGGL(agg_orphans_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(agg), dptr<int>(orphan), nagg);
    nagg += norph;
  }
  if (nrow) {
    crow.alloc((size_t)nagg * sizeof(int));
    ccol.alloc((size_t)nagg * sizeof(int));
    hipLaunchKernelGGL(agg_coords_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(agg), dptr<unsigned long long>(key),
                       dptr<int>(orphan_flag), nrow, ncol, dptr<int>(crow), dptr<int>(ccol));
  }
  check_launch("aggregate");
  CS_HIP(hipStreamSynchronize(st));
  return nagg;
}

template <class T>
inline void level_stats(Level<T>& L, DBuf& diag, DBuf& labs, double omega_s, hipStream_t st) {
  const int n = L.A.nrows;
  diag.alloc((size_t)n * sizeof(T));
  labs.alloc((size_t)n * sizeof(T));
  const int g = grid_for(n);
  DBuf part = dalloc<double>(2 * (size_t)g);
  hipLaunchKernelGGL((row_stats_kernel<T>), dim3(g), dim3(256), 0, st, n, L.A.rp(), L.A.ci(), L.A.va(), dptr<T>(diag),
                     dptr<T>(labs), dptr<double>(part), dptr<double>(part) + g);
  L.dinv.alloc((size_t)n * sizeof(T));
  std::vector<double> hp(2 * (size_t)g);
  CS_HIP(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  double rho = 0, dmax = 0;
  for (int b = 0; b < g; ++b) {
    rho = std::max(rho, hp[b]);
    dmax = std::max(dmax, hp[(size_t)g + b]);
  }
  hipLaunchKernelGGL((dinv_kernel<T>), dim3(g), dim3(256), 0, st, n, dptr<T>(diag), dptr<T>(labs),
                     64.0 * (double)std::numeric_limits<T>::epsilon() * dmax, dptr<T>(L.dinv));
  if (!(rho > 0)) rho = 1.0;
  L.rho = rho;
  L.omega = omega_s / rho;
  L.n = n;
  spmv_block_order(L.A, L.orderA, st, &L.periodA);
}

// State the level loop carries from one level to the next (also the hand-over from lattice_setup.h, which builds level 0
// of a raster without a CSR matrix and enters the loop at level 1).
struct SetupCarry {
  DBuf crow, ccol;   // raster coordinates of the current level's nodes (may be empty)
  DBuf size;         // long long: fine nodes under every node of the current level (empty: all ones)
  int gridR = 0, gridC = 0;  // raster extent of the current level while the aggregates are the regular tiles (0: unknown)
  bool weighted = false;     // rows carry weights although the caller gave none (strength filter on an all-valid raster)
  double tile_theta = -1.0;  // strength filter of this hierarchy's piece analyses (< 0: level 0 decides, TileStrength)
};

template <class T>
inline void amg_setup_levels(Hierarchy<T>& H, const SetupParams& sp, const int* cur_row, const int* cur_col,
                             SetupCarry& carry, hipStream_t st);

// lattice_setup.h: lattice forms of level 1 (A, the two-sweep smoother, the transfer operator) when that level is an
// R x C nine-point lattice with regular 3 x 3 aggregates `agg`; leaves the level untouched otherwise
template <class T>
inline void lattice_level1_setup(Level<T>& L, const int* agg, int R, int C, int nagg, hipStream_t st);

// Build the hierarchy. A0 is moved into level 0. node_row/node_col (device, may be null) are raster coordinates.
template <class T>
inline void amg_setup(Hierarchy<T>& H, Csr<T>&& A0, const SetupParams& sp, const int* node_row, const int* node_col,
                      hipStream_t st) {
  hipEvent_t e0, e1;
  CS_HIP(hipEventCreate(&e0));
  CS_HIP(hipEventCreate(&e1));
  CS_HIP(hipEventRecord(e0, st));
  H.levels.clear();
  H.levels.emplace_back();
  H.levels.back().A = std::move(A0);
  SetupCarry carry;
  const int* cur_row = (sp.aggregation == CSGPU_AGG_MIS2) ? nullptr : node_row;
  const int* cur_col = (sp.aggregation == CSGPU_AGG_MIS2) ? nullptr : node_col;
  if (sp.size0) {
    carry.size.alloc((size_t)H.levels[0].A.nrows * sizeof(long long));
    CS_HIP(hipMemcpyAsync(carry.size.p, sp.size0, carry.size.bytes, hipMemcpyDeviceToDevice, st));
  }
  carry.gridR = cur_row ? sp.grid_rows : 0;
  carry.gridC = cur_row ? sp.grid_cols : 0;
  amg_setup_levels(H, sp, cur_row, cur_col, carry, st);
  CS_HIP(hipEventRecord(e1, st));
  CS_HIP(hipEventSynchronize(e1));
  float ms = 0;
  CS_HIP(hipEventElapsedTime(&ms, e0, e1));
  H.setup_ms = ms;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

// The level loop: coarsens H.levels.back() until the coarsest level is reached, then builds the coarse solver.
// cur_row / cur_col: raster coordinates of the nodes of H.levels.back() (device; null = none; carry.crow / carry.ccol
// take over from the next level on).
template <class T>
inline void amg_setup_levels(Hierarchy<T>& H, const SetupParams& sp, const int* cur_row, const int* cur_col,
                             SetupCarry& carry, hipStream_t st) {
  DBuf crow_prev = std::move(carry.crow), ccol_prev = std::move(carry.ccol);  // coarse coordinates of the current level
  if (!cur_row && crow_prev.p && sp.aggregation != CSGPU_AGG_MIS2) {
    cur_row = dptr<int>(crow_prev);
    cur_col = dptr<int>(ccol_prev);
  }
  DBuf size_prev = std::move(carry.size);  // long long fine sizes of the current level (null => all ones)
  int gridR = cur_row ? carry.gridR : 0, gridC = cur_row ? carry.gridC : 0;  // raster extent of the current level
  for (;;) {
    Level<T>& L = H.levels.back();
    DBuf diag, labs;
    level_stats(L, diag, labs, sp.omega_s, st);
    const int n = L.A.nrows;
    if (n <= sp.max_coarse || (int)H.levels.size() >= sp.max_levels) break;
    if (sp.coarse_chebyshev && H.levels.size() > 1) {
      // Chebyshev smoothing as a sequence of damped-Jacobi sweeps whose weights are the reciprocals of the roots of the
      // degree-m Chebyshev polynomial on [lam_max / 10, lam_max]: prod_j (I - w_j D^-1 A) IS that polynomial, the factors
      // commute (so pre- and post-smoother are each other's adjoint whatever the order), every sweep is the existing
      // fused Jacobi epilogue and the first post-sweep still folds into Q = P - w_1 D^-1 A P.
      // lam_max is the Gershgorin bound: never below rho(D^-1 A), so no eigenvalue is ever amplified. (A 12-step power
      // iteration x 1.1 was tried first: sharper on the Galerkin operators, whose bound is loose -- 6.4 against a true
      // 3.2 on level 1 of a log-normal sigma = 3 raster -- and 5 % fewer iterations at 300^2, but it underestimates rho on
      // the 1e7-row level of a 10000^2 raster and the polynomial then amplifies the top of the spectrum: 1792 iterations.)
      const int m = H.levels.size() == 2 ? sp.nu_l1 : sp.nu_deep;
      if (m >= 1) {
        L.lam_max = L.rho;
        const double lo = L.lam_max / 10.0, theta = 0.5 * (L.lam_max + lo), delta = 0.5 * (L.lam_max - lo);
        L.weights.resize(m);
        for (int j = 0; j < m; ++j) L.weights[j] = 1.0 / (theta + delta * std::cos(M_PI * (2 * j + 1) / (2.0 * m)));
        L.omega = L.weights[0];  // (the smallest weight: the root next to lam_max)
      }
    }
    DBuf agg, crow, ccol;
    long long* wts = ((sp.size0 || carry.weighted) && size_prev.p) ? dptr<long long>(size_prev) : (long long*)nullptr;
    TileStrength ts;
    ts.theta = carry.tile_theta < 0.0 ? sp.tile_theta : carry.tile_theta;
    ts.split_min = sp.tile_split_min;
    ts.decide = carry.tile_theta < 0.0;
    if (ts.decide && ts.theta > 0.0 && !wts && sp.theta == 0.0 && cur_row && gridR >= 6 && gridC >= 6 &&
        (int64_t)gridR * gridC == n && sp.aggregation != CSGPU_AGG_MIS2) {
      // an all-valid raster: unit weights so that the piece analysis can run the heterogeneity test
      size_prev.alloc((size_t)n * sizeof(long long));
      hipLaunchKernelGGL(fill_ll_kernel, dim3(grid_for(n)), dim3(256), 0, st, dptr<long long>(size_prev), (int64_t)n, 1LL);
      wts = dptr<long long>(size_prev);
      ts.unit_weights = true;
    }
    const bool cell_level = (sp.size0 || ts.unit_weights) && H.levels.size() == 1;
    if (knobs().expander_probe && H.levels.size() == 1 && !cur_row && !wts && sp.theta == 0.0 && n >= 100000) {
      // a large graph without coordinates: would the aggregation be thrown away? (expansion_probe_kernel)
      double mean_row = 0.0;
      const double est = expansion_estimate(L.A, st, &mean_row);
      if (knobs().verbose)
        fprintf(stderr, "csgpu: expansion probe: mean row %.1f entries, estimated nnz(P) / nnz(A) after MIS(2) = %.2f\n", mean_row, est);
      if (est > 0.8 && mean_row >= 7.0) {
        H.expander_probe_hit = true;
        L.weights.clear();
        L.omega = sp.omega_s / L.rho;
        break;
      }
    }
    int nagg = aggregate(L.A, dptr<T>(diag), sp.theta, cur_row, cur_col, agg, crow, ccol, st, gridR, gridC, wts, cell_level, &ts);
    if (ts.decide) {
      H.hetero_frac = ts.hetero_frac;
      carry.tile_theta = ts.theta;  // (0 when the test declined, or when this level has no regular tiles)
      if (ts.unit_weights) {
        if (ts.theta > 0.0) {
          carry.weighted = true;
        } else {
          size_prev.release();
          wts = nullptr;
        }
      }
    }
    if (sp.theta > 0.0 && (double)nagg > 0.5 * (double)n) {
      // the strength filter left too few strong couplings to coarsen this level: aggregate on the full pattern
      nagg = aggregate(L.A, dptr<T>(diag), 0.0, cur_row, cur_col, agg, crow, ccol, st, gridR, gridC, wts, cell_level, &ts);
    }
    const int lvlR = gridR, lvlC = gridC;  // raster extent of THIS level
    // coarse raster extent (tile counts), valid while the aggregates are the regular tiles
    gridR = gridR > 0 ? (gridR + 1) / 3 : 0;
    gridC = gridC > 0 ? (gridC + 1) / 3 : 0;
    if ((int64_t)gridR * gridC != nagg) gridR = gridC = 0;
    if (nagg >= n || nagg < 1 || (double)nagg > 0.8 * (double)n) {  // coarsening stagnated: this is the last level
      L.weights.clear();
      L.omega = sp.omega_s / L.rho;
      break;
    }
    // sizes
    DBuf size_c = dalloc<unsigned long long>(nagg);
    CS_HIP(hipMemsetAsync(size_c.p, 0, (size_t)nagg * sizeof(unsigned long long), st));
    const int g = grid_for(n);
    hipLaunchKernelGGL(agg_sizes_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(agg),
                       (const long long*)size_prev.p, dptr<unsigned long long>(size_c));
    // tentative prolongator
    Csr<T> Tm;
    Tm.nrows = n;
    Tm.ncols = nagg;
    Tm.nnz = n;
    Tm.rowptr.alloc((size_t)(n + 1) * sizeof(int));
    Tm.col.alloc((size_t)n * sizeof(int));
    Tm.val.alloc((size_t)n * sizeof(T));
    hipLaunchKernelGGL((tentative_kernel<T>), dim3(g), dim3(256), 0, st, n, dptr<int>(agg),
                       (const long long*)size_prev.p, dptr<unsigned long long>(size_c), Tm.rp(), Tm.ci(), Tm.va());
    check_launch("tentative");
    // P = T - omega_p Dl^-1 A T
    spgemm_tentative(L.A, Tm, (const int*)dptr<int>(agg), L.P, st);
    if ((double)L.P.nnz > 0.75 * (double)L.A.nnz) {
      // (nearly) every neighbour of every node lies in a different aggregate: the graph has no locality for
      // aggregation to exploit (expander-like networks, e.g. random graphs; rasters sit at 0.3). The Galerkin
      // operator would be dense (measured on a 1e6-node random graph: 2.6e8 nonzeros on 16081 coarse nodes,
      // operator complexity 13, 52 s of setup) while such graphs are well conditioned to begin with: stop here,
      // this level becomes the coarsest one (damped-Jacobi sweeps, or the dense inverse if it is small).
      L.P = Csr<T>();
      L.weights.clear();
      L.omega = sp.omega_s / L.rho;
      break;
    }
    DBuf missing = dalloc<int>(1);
    CS_HIP(hipMemsetAsync(missing.p, 0, sizeof(int), st));
    hipLaunchKernelGGL((smooth_prolongator_kernel<T>), dim3(g), dim3(256), 0, st, n, L.P.rp(), L.P.ci(), L.P.va(),
                       dptr<int>(agg), Tm.va(), dptr<T>(labs), sp.omega_p, dptr<int>(missing), (const long long*)wts);
    check_launch("smooth prolongator");
    CS_REQUIRE(read_int(dptr<int>(missing), st) == 0, CSGPU_BAD_ARGS,
               "matrix has rows without a stored diagonal entry (not a graph Laplacian)");
    transpose(L.P, L.R, st);
    Csr<T> AP, Ac;
    spgemm(L.A, L.P, AP, st);
    spgemm(L.R, AP, Ac, st);
    {
      // Q = P - omega D^-1 A P, written over AP (no longer needed)
      CS_HIP(hipMemsetAsync(missing.p, 0, sizeof(int), st));
      hipLaunchKernelGGL((build_q_kernel<T>), dim3(g), dim3(256), 0, st, n, AP.rp(), AP.ci(), AP.va(), L.P.rp(), L.P.ci(),
                         L.P.va(), dptr<T>(L.dinv), (T)L.omega, dptr<int>(missing));
      check_launch("build Q");
      CS_REQUIRE(read_int(dptr<int>(missing), st) == 0, CSGPU_INTERNAL, "pattern(P) is not contained in pattern(A*P)");
      L.Q = std::move(AP);
    }
    if (H.levels.size() == 2 && sp.lattice_s && gridR > 0 && (int64_t)lvlR * lvlC == n && sp.nu_l1 == 2 &&
        (L.weights.empty() || L.weights.size() == 2))
      lattice_level1_setup(L, (const int*)dptr<int>(agg), lvlR, lvlC, nagg, st);
    if (H.levels.size() >= 2 && !L.lattice_v22() && lvlR >= 6 && (int64_t)lvlR * lvlC == n && n >= dia25_min_rows()) {
      // refined tiles: the level's operator reaches two lattice steps; index-free 25-point form when it fits (dia25.h)
      if (dia25_from_csr(L.A, lvlR, L.A25, st) && knobs().verbose)
        fprintf(stderr, "csgpu: level %d (%d x %d) in 25-point lattice form\n", (int)H.levels.size() - 1, lvlR, lvlC);
    }
    if (sp.two_product && H.levels.size() == 1 && L.A.nnz + L.Q.nnz < 0x7fffffffLL &&
        (int64_t)n + nagg < 0x7fffffffLL) {
      if (sp.lattice_s) {
        L.agg0 = std::move(agg);  // the caller turns Q into its index-free form (or falls back to the CSR forms)
      } else {
        build_qt_matrix(L, st);
        build_sq_matrix(L, st);
      }
      if (knobs().verbose)
        fprintf(stderr, "csgpu: two-product level: nnz(Q^T)=%lld nnz([S Q])=%lld periodA=%lld orderA=%s orderQT=%s\n",
                (long long)L.QT.nnz, (long long)L.M.nnz, L.periodA, L.orderA.p ? "yes" : "no", L.orderQT.p ? "yes" : "no");
    }
    // next level
    size_prev = std::move(size_c);  // unsigned long long and long long share the representation for these counts
    crow_prev = std::move(crow);
    ccol_prev = std::move(ccol);
    cur_row = crow_prev.p ? dptr<int>(crow_prev) : nullptr;
    cur_col = ccol_prev.p ? dptr<int>(ccol_prev) : nullptr;
    H.levels.emplace_back();
    H.levels.back().A = std::move(Ac);
    if (nagg <= 4096) {  // (levels the coarse tail may run; see tail.h)
      Level<T>& Ln = H.levels.back();
      Ln.cand.alloc((size_t)nagg * sizeof(T));
      hipLaunchKernelGGL((candidate_kernel<T>), dim3(grid_for(nagg)), dim3(256), 0, st, nagg, (const long long*)size_prev.p,
                         dptr<T>(Ln.cand));
    }
  }
  // coarsest level: dense pseudo-inverse when small enough
  {
    Level<T>& L = H.levels.back();
    const int n = L.A.nrows;
    H.coarse_n = n;
    // dense pseudo-inverse by cyclic Jacobi on the host: O(n^3) per sweep, so only for small levels (max_coarse, or a
    // level left by a stagnation / expander bail-out that happens to be small); larger ones get damped-Jacobi sweeps
    H.coarse_dense = n <= 400;
    if (H.coarse_dense) {
      std::vector<int> rp(n + 1), ci((size_t)L.A.nnz);
      std::vector<T> va((size_t)L.A.nnz);
      CS_HIP(hipMemcpyAsync(rp.data(), L.A.rp(), (size_t)(n + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
      if (L.A.nnz > 0) {
        CS_HIP(hipMemcpyAsync(ci.data(), L.A.ci(), (size_t)L.A.nnz * sizeof(int), hipMemcpyDeviceToHost, st));
        CS_HIP(hipMemcpyAsync(va.data(), L.A.va(), (size_t)L.A.nnz * sizeof(T), hipMemcpyDeviceToHost, st));
      }
      CS_HIP(hipStreamSynchronize(st));
      std::vector<double> M((size_t)n * n, 0.0);
      for (int i = 0; i < n; ++i)
        for (int k = rp[i]; k < rp[i + 1]; ++k) M[(size_t)i * n + ci[k]] += (double)va[k];
      std::vector<std::vector<double>> kc;
      const bool deflate = sizeof(T) == 4 && knobs().deflation;
      std::vector<double> cand((size_t)n, 1.0);
      if (size_prev.p) {
        std::vector<long long> sz((size_t)n);
        CS_HIP(hipMemcpyAsync(sz.data(), size_prev.p, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost, st));
        CS_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < n; ++i) cand[i] = std::sqrt((double)sz[i]);
      }
      kc = component_candidates(M, n, cand);
      // Dirichlet-masked solves may use the correction along the per-component candidates (pcg.h, DirichletCoarse);
      // weightless rows of a cell-space hierarchy have a zero candidate entry and belong to no component
      const bool dir_ok = !kc.empty() && (int)kc.size() <= kMaxDirComp && knobs().dirichlet_coarse;
      H.dir_ncomp = dir_ok ? (int)kc.size() : 0;
      int dropped = 0;
      std::vector<double> Pdefl;
      std::vector<double> Pi = dense_sym_pinv(std::move(M), n, (double)std::numeric_limits<T>::epsilon(),
                                              deflate ? &kc : nullptr, 1e-2, &dropped, &kc,
                                              dir_ok ? &Pdefl : nullptr);
      if (dir_ok) {
        std::vector<T> Dt((size_t)n * n), ct((size_t)n);
        std::vector<int> cc((size_t)n, -1);
        for (size_t i = 0; i < Dt.size(); ++i) Dt[i] = (T)Pdefl[i];
        for (int i = 0; i < n; ++i) ct[i] = (T)cand[i];
        for (size_t k = 0; k < kc.size(); ++k)
          for (int i = 0; i < n; ++i)
            if (kc[k][i] != 0) cc[i] = (int)k;
        H.coarse_inv_defl.alloc(std::max<size_t>(Dt.size(), 1) * sizeof(T));
        H.coarse_cand.alloc((size_t)n * sizeof(T));
        H.coarse_comp.alloc((size_t)n * sizeof(int));
        CS_HIP(hipMemcpyAsync(H.coarse_inv_defl.p, Dt.data(), Dt.size() * sizeof(T), hipMemcpyHostToDevice, st));
        CS_HIP(hipMemcpyAsync(H.coarse_cand.p, ct.data(), ct.size() * sizeof(T), hipMemcpyHostToDevice, st));
        CS_HIP(hipMemcpyAsync(H.coarse_comp.p, cc.data(), cc.size() * sizeof(int), hipMemcpyHostToDevice, st));
        CS_HIP(hipStreamSynchronize(st));
      }
      H.near_singular = deflate && dropped > 0;
      H.cand_norm2 = sp.n_real > 0 ? (double)sp.n_real : (double)H.levels[0].A.nrows;
      if (deflate && knobs().verbose)
        fprintf(stderr, "csgpu: coarsest level: %d near-kernel eigenpair(s) of %zu candidate(s) dropped\n", dropped, kc.size());
      std::vector<T> Pt((size_t)n * n);
      for (size_t i = 0; i < Pt.size(); ++i) Pt[i] = (T)Pi[i];
      H.coarse_inv.alloc(std::max<size_t>(Pt.size(), 1) * sizeof(T));
      CS_HIP(hipMemcpyAsync(H.coarse_inv.p, Pt.data(), Pt.size() * sizeof(T), hipMemcpyHostToDevice, st));
      CS_HIP(hipStreamSynchronize(st));
    }
  }
}

}  // namespace csgpu
