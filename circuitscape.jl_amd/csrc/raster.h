// raster.h -- device-side construction of the raster graph Laplacian (scope row N4, used by bench.py and tests).
//
// GPU counterpart, for a raster without polygons (NODATA = conductance <= 0 allowed), of
//   construct_node_map   src/raster/pairwise.jl:271-301  (column-major numbering of cells with conductance > 0)
//   construct_graph      src/raster/pairwise.jl:316-362  (E, S, SE, NE neighbours; cond_avg / res_avg, diagonals / sqrt 2)
//   laplacian!           src/core.jl:608-634
//   connected_components src/raster/pairwise.jl:233 (Graphs.jl), as min-label hooking + pointer jumping
// producing the CSR Laplacian directly in HBM (no COO, no host transient): one thread per cell writes its own
// sorted row. Also emits each node's (row, col) for the tile-seeded aggregation.
#pragma once
#include "prims.h"

namespace csgpu {

__device__ __forceinline__ double raster_edge(double x, double y, bool diag, bool avg_res) {
  double v = avg_res ? 1.0 / ((1.0 / x + 1.0 / y) * 0.5) : (x + y) * 0.5;
  return diag ? v / 1.4142135623730951 : v;
}

// Cells with conductance <= 0 (or NaN) are NODATA and get no node (construct_node_map numbers the cells with
// gmap > 0 in column-major order, pairwise.jl:273-275). flag: one int per cell in COLUMN-major order (id = j*R + i).
// colmajor != 0: cond is the transposed copy (raster_transpose_kernel, prims.h), cell (i, j) at j*R + i = id.
template <class T>
__global__ __launch_bounds__(256) void raster_valid_kernel(int R, int C, const T* __restrict__ cond,
                                                           int* __restrict__ flag, int colmajor = 0) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    if (colmajor) {
      flag[id] = cond[id] > T(0) ? 1 : 0;
      continue;
    }
    int i, j;
    cell_rc(id, R, i, j);
    flag[id] = cond[(size_t)i * C + j] > T(0) ? 1 : 0;
  }
}

// node[id] = exclusive scan of the valid flags (the node index of a valid cell). counts[node] = stored entries of
// the node's Laplacian row (itself + its valid neighbours); nodemap (row-major, 1-based, 0 = no node) for the host.
// cellspace != 0 ("cell space", see csgpu.hip): the matrix keeps one row per CELL of the raster (row id = j*R + i); a
// NODATA cell gets a row holding its diagonal entry only (an isolated unknown whose right-hand side is always zero), so
// the matrix of a raster with holes is still a lattice. cellmap (may be null): row-major, 1-based ROW id of every cell
// with a node, 0 elsewhere (= nodemap when cellspace == 0).
template <class T>
__global__ __launch_bounds__(256) void raster_count_kernel(int R, int C, int four, const T* __restrict__ cond,
                                                           const int* __restrict__ node, int* __restrict__ counts,
                                                           int* __restrict__ nodemap, int cellspace = 0,
                                                           int* __restrict__ cellmap = nullptr,
                                                           int* __restrict__ node2cell = nullptr,
                                                           int* __restrict__ cell2node = nullptr) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const bool valid = cond[(size_t)i * C + j] > T(0);
    nodemap[(size_t)i * C + j] = valid ? node[id] + 1 : 0;
    if (cellmap) cellmap[(size_t)i * C + j] = valid ? (int)id + 1 : 0;
    if (node2cell && valid) node2cell[node[id]] = (int)id;
    if (cell2node) cell2node[id] = valid ? node[id] + 1 : 0;  // (column-major cell id -> 1-based node id, 0 = no node)
    if (!valid) {
      if (cellspace) counts[id] = 1;
      continue;
    }
    int cnt = 1;
    for (int dj = -1; dj <= 1; ++dj) {
      const int jj = j + dj;
      if (jj < 0 || jj >= C) continue;
      for (int di = -1; di <= 1; ++di) {
        const int ii = i + di;
        if (ii < 0 || ii >= R || (di == 0 && dj == 0)) continue;
        if (four && di != 0 && dj != 0) continue;
        cnt += cond[(size_t)ii * C + jj] > T(0) ? 1 : 0;
      }
    }
    counts[cellspace ? (int)id : node[id]] = cnt;
  }
}

// cond: device array, row-major [R][C] (cell (i,j) at i*C + j). One thread per valid cell writes its sorted row
// (neighbours visited in column-major order, and the node numbering is monotone in that order).
template <class T>
__global__ __launch_bounds__(256) void raster_fill_kernel(int R, int C, int four, int avg_res,
                                                          const T* __restrict__ cond, const int* __restrict__ node,
                                                          const int* __restrict__ rp, int* __restrict__ ci,
                                                          T* __restrict__ va, int* __restrict__ nrow,
                                                          int* __restrict__ ncol, const T* __restrict__ ground,
                                                          T* __restrict__ ground_node, int cellspace = 0) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const double g0 = (double)cond[(size_t)i * C + j];
    if (!(g0 > 0.0)) {
      if (cellspace) {  // NODATA cell: a row of its own holding the diagonal only (value set by raster_identity_kernel)
        ci[rp[id]] = (int)id;
        va[rp[id]] = T(0);
        nrow[id] = i;
        ncol[id] = j;
        if (ground_node) ground_node[id] = T(0);
      }
      continue;
    }
    const int me = cellspace ? (int)id : node[id];
    nrow[me] = i;
    ncol[me] = j;
    int k = rp[me];
    int kdiag = -1;
    double deg = 0.0;
    for (int dj = -1; dj <= 1; ++dj) {
      const int jj = j + dj;
      if (jj < 0 || jj >= C) continue;
      for (int di = -1; di <= 1; ++di) {
        const int ii = i + di;
        if (ii < 0 || ii >= R) continue;
        const bool self = (di == 0 && dj == 0);
        const bool diag = (di != 0 && dj != 0);
        if (diag && four) continue;
        if (self) {
          ci[k] = me;
          kdiag = k;
        } else {
          const double g1 = (double)cond[(size_t)ii * C + jj];
          if (!(g1 > 0.0)) continue;
          ci[k] = cellspace ? (int)((int64_t)jj * R + ii) : node[(int64_t)jj * R + ii];
          const double w = raster_edge(g0, g1, diag, avg_res != 0);
          va[k] = (T)(-w);
          deg += w;
        }
        ++k;
      }
    }
    // finite ground conductance of the cell (advanced-mode `asolve = a + spdiagm(finitegrounds)`, advanced.jl:277-280)
    const double gnd = ground ? (double)ground[(size_t)i * C + j] : 0.0;
    if (ground_node) ground_node[me] = (T)gnd;
    va[kdiag] = (T)(deg + gnd);
  }
}

// Right-hand side of an advanced-mode solve from a source raster (row-major): b[node] = source at the node's cell.
// The source / ground conflict policy (remove_src_or_gnd) is applied by the caller to the rasters it hands over.
// has[2*comp + 0/1] flags the components that hold a source / a ground (advanced_kernel solves only those that
// hold both, advanced.jl:186-191).
template <class T>
__global__ __launch_bounds__(256) void raster_rhs_kernel(int64_t ncells, const int* __restrict__ nodemap,
                                                         const T* __restrict__ source, const T* __restrict__ ground_node,
                                                         const int* __restrict__ comp, T* __restrict__ b,
                                                         int* __restrict__ has) {
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * 256) {
    const int nd = nodemap[c] - 1;
    if (nd < 0) continue;
    const T g = ground_node ? ground_node[nd] : T(0);
    const T s = source[c];
    b[nd] = s;
    if (s != T(0)) atomicOr(&has[2 * comp[nd]], 1);
    if (g != T(0)) atomicOr(&has[2 * comp[nd] + 1], 1);
  }
}

template <class T>
__global__ __launch_bounds__(256) void raster_rhs_mask_kernel(int n, const int* __restrict__ comp,
                                                              const int* __restrict__ has, T* __restrict__ b) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256)
    if (!(has[2 * comp[u]] && has[2 * comp[u] + 1])) b[u] = T(0);
}

// ---- per-component right-hand-side normalisation of a block-diagonal solve -----------------------------------------
// The reference solves every component of an advanced-mode problem separately, each to its own relative tolerance
// (src/raster/advanced.jl:186-312). One PCG over the block-diagonal system has ONE stopping rule, so a component whose
// sources are orders of magnitude weaker than the others' would be under-resolved. Scaling every component's
// right-hand side by an exact power of two that brings its largest entry into [1, 2) makes all components weigh alike
// in the stopping rule; the solution is scaled back by the inverse power of two -- both scalings are exact in floating
// point, and the per-component maximum is order-independent (atomicMax on the bit pattern), so results stay
// bit-reproducible. absmax[c] holds the bit pattern of max |b| over component c.
template <class T>
__global__ __launch_bounds__(256) void comp_absmax_kernel(int n, const int* __restrict__ comp, const T* __restrict__ b,
                                                          unsigned long long* __restrict__ absmax) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const double a = fabs((double)b[u]);
    if (a > 0.0) atomicMax(&absmax[comp[u]], (unsigned long long)__double_as_longlong(a));
  }
}

// v[u] *= 2^(sign * (1 - exponent(absmax[comp[u]])))   (sign = +1: normalise, -1: undo)
template <class T>
__global__ __launch_bounds__(256) void comp_scale_kernel(int n, const int* __restrict__ comp,
                                                         const unsigned long long* __restrict__ absmax, int sign,
                                                         T* __restrict__ v) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const double m = __longlong_as_double((long long)absmax[comp[u]]);
    if (!(m > 0.0)) continue;
    int e = 0;
    frexp(m, &e);  // m = f * 2^e, f in [0.5, 1)
    v[u] = (T)ldexp((double)v[u], sign * (1 - e));
  }
}

// per-component residual check of a block-diagonal solve: rr[c] += res[u]^2, bb[c] += b[u]^2 (check only: the order of
// the atomic additions does not feed back into any result)
template <class T>
__global__ __launch_bounds__(256) void comp_norms_kernel(int n, const int* __restrict__ comp, const T* __restrict__ res,
                                                         const T* __restrict__ b, double* __restrict__ rr,
                                                         double* __restrict__ bb) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const double r = (double)res[u], v = (double)b[u];
    if (r != 0.0) atomicAdd(&rr[comp[u]], r * r);
    if (v != 0.0) atomicAdd(&bb[comp[u]], v * v);
  }
}

// worst[0] = max over components with a right-hand side of sqrt(rr / bb)
__global__ __launch_bounds__(256) void comp_relres_kernel(int ncomp, const double* __restrict__ rr,
                                                          const double* __restrict__ bb, double* __restrict__ worst) {
  __shared__ double sm[4];
  double m = 0.0;
  for (int c = threadIdx.x; c < ncomp; c += 256)
    if (bb[c] > 0.0) {
      const double q = sqrt(rr[c] / bb[c]);
      m = q > m ? q : m;
    }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double w = sm[0];
    for (int i = 1; i < 4; ++i) w = sm[i] > w ? sm[i] : w;
    worst[0] = w;
  }
}

// out[cell] = vec[node of the cell], 0 where the cell has no node (_create_current_maps / _create_voltage_map)
template <class T>
__global__ __launch_bounds__(256) void raster_scatter_kernel(int64_t ncells, const int* __restrict__ nodemap,
                                                             const T* __restrict__ vec, T* __restrict__ out) {
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * 256) {
    const int nd = nodemap[c] - 1;
    out[c] = nd >= 0 ? vec[nd] : T(0);
  }
}

// ---- connected components of a symmetric CSR graph (connected_components(SimpleGraph(G)), pairwise.jl:233,
// advanced.jl:59): min-label hooking + pointer jumping. Invariant: f[x] <= x and f[x] lies in x's component; at the
// fixed point every node of a component carries the component's smallest node id.
// An off-diagonal entry that is not NEGATIVE carries no conductance and is no edge: construct_graph stores the zero weights
// res_avg yields next to a zero-conductance cell that still has a node (a NODATA cell inside a short-circuit polygon,
// src/raster/pairwise.jl:316-362), the regularisation then lifts them to +eps ||nzval|| (src/core.jl:161), and the reference's
// connected_components(SimpleGraph(A)) never sees them as edges (A[i, j] != 0 decides there). Found by tools/fuzz_polygons.py
// (round 6, seed 61: two parts of a raster hanging together through such entries only were one "component" here, and
// pairs across them were solved on a numerically disconnected system instead of being refused).
template <class T>
__global__ __launch_bounds__(256) void cc_hook_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const T* __restrict__ va, int* __restrict__ f,
                                                      int* __restrict__ changed) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const int fu = f[u];
    int m = fu;
    for (int k = rp[u]; k < rp[u + 1]; ++k) {
      if (va && ci[k] != u && !(va[k] < T(0))) continue;
      m = min(m, f[ci[k]]);
    }
    if (m < fu) {
      atomicMin(&f[fu], m);  // hook u's current representative under the smaller label seen next door
      atomicMin(&f[u], m);
      *changed = 1;
    }
  }
}

__global__ __launch_bounds__(256) void cc_jump_kernel(int n, int* __restrict__ f, int* __restrict__ changed) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const int p = f[u];
    const int gp = f[p];
    if (gp != p) {
      f[u] = gp;
      *changed = 1;
    }
  }
}

__global__ __launch_bounds__(256) void cc_root_flag_kernel(int n, const int* __restrict__ f, int* __restrict__ flag) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) flag[u] = f[u] == u ? 1 : 0;
}

__global__ __launch_bounds__(256) void cc_relabel_kernel(int n, const int* __restrict__ f, const int* __restrict__ idx,
                                                         int* __restrict__ label) {
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) label[u] = idx[f[u]];
}

// label[u] = dense component index (components ordered by their smallest node id); returns the number of components.
template <class T>
inline int connected_components(int n, const int* rp, const int* ci, const T* va, int* label, hipStream_t st) {
  if (n <= 0) return 0;
  DBuf f = dalloc<int>(n), flag = dalloc<int>((size_t)n + 1), changed = dalloc<int>(1);
  const int g = grid_for(n);
  hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(256), 0, st, dptr<int>(f), (int64_t)n);
  for (int round = 0; round < 256; ++round) {
    CS_HIP(hipMemsetAsync(changed.p, 0, sizeof(int), st));
    hipLaunchKernelGGL((cc_hook_kernel<T>), dim3(g), dim3(256), 0, st, n, rp, ci, va, dptr<int>(f), dptr<int>(changed));
    if (read_int(dptr<int>(changed), st) == 0) break;
    for (int pass = 0; pass < 64; ++pass) {
      CS_HIP(hipMemsetAsync(changed.p, 0, sizeof(int), st));
      hipLaunchKernelGGL(cc_jump_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(f), dptr<int>(changed));
      if (read_int(dptr<int>(changed), st) == 0) break;
    }
    CS_REQUIRE(round < 255, CSGPU_INTERNAL, "connected components did not converge");
  }
  CS_HIP(hipMemsetAsync(flag.p, 0, ((size_t)n + 1) * sizeof(int), st));
  hipLaunchKernelGGL(cc_root_flag_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(f), dptr<int>(flag));
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(dptr<int>(flag), (int64_t)n + 1, st, dptr<int>(total));
  hipLaunchKernelGGL(cc_relabel_kernel, dim3(g), dim3(256), 0, st, n, dptr<int>(f), dptr<int>(flag), label);
  check_launch("connected components");
  return read_int(dptr<int>(total), st);
}

// nzval .+= eps(T) * norm(nzval)   (src/core.jl:161); norm2 partials come from dot_kernel<T,1,false>
template <class T>
__global__ __launch_bounds__(256) void add_scalar_kernel(int64_t nnz, T* __restrict__ va, const double* __restrict__ partials,
                                                         int nparts, double eps) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
  s = block_sum_256(s, sm);
  const T shift = (T)(eps * sqrt(s));
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * 256) va[k] += shift;
}

// cell space: diagonal of the NODATA rows = 1 (after the regularisation shift, which must see 0 there: the reference's
// norm runs over the entries of the real graph only); size0[cell] = 1 for a cell with a node, 0 otherwise -- the
// weight of the cell in the aggregates (amg_setup.h: a NODATA cell carries no part of the near-null-space candidate)
template <class T>
__global__ __launch_bounds__(256) void raster_identity_kernel(int R, int C, const T* __restrict__ cond,
                                                              const int* __restrict__ rp, T* __restrict__ va,
                                                              long long* __restrict__ size0) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const bool valid = cond[(size_t)i * C + j] > T(0);
    size0[id] = valid ? 1 : 0;
    if (!valid) va[rp[id]] = T(1);
  }
}

// cell space <-> node numbering of the reference, column-major n x ncols arrays
template <class T>
__global__ __launch_bounds__(256) void cells_from_nodes_kernel(int64_t ncell, int64_t nnode, const int* __restrict__ cell2node,
                                                               const T* __restrict__ in, int ncols, T* __restrict__ out) {
  const int64_t total = ncell * ncols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t c = t / ncell, cell = t % ncell;
    const int nd = cell2node[cell] - 1;
    out[t] = nd >= 0 ? in[c * nnode + nd] : T(0);
  }
}
template <class T>
__global__ __launch_bounds__(256) void nodes_from_cells_kernel(int64_t ncell, int64_t nnode, const int* __restrict__ node2cell,
                                                               const T* __restrict__ in, int ncols, T* __restrict__ out) {
  const int64_t total = nnode * ncols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t c = t / nnode, nd = t % nnode;
    out[t] = in[c * ncell + node2cell[nd]];
  }
}
// out[k] = map[ids[k]] (64-bit ids in, 64-bit out)
__global__ __launch_bounds__(256) void map_ids_kernel(int64_t cnt, const int64_t* __restrict__ ids,
                                                      const int* __restrict__ map, int64_t* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * 256) out[k] = map[ids[k]];
}
// out[k] = v[idx[k]]
__global__ __launch_bounds__(256) void gather_int_kernel(int64_t cnt, const int64_t* __restrict__ idx,
                                                         const int* __restrict__ v, int* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * 256) out[k] = v[idx[k]];
}
// components of a cell-space graph in the reference's terms: keep[label] = 1 for the labels that own a real node
__global__ __launch_bounds__(256) void comp_keep_kernel(int64_t nnode, const int* __restrict__ node2cell,
                                                        const int* __restrict__ label, int* __restrict__ keep) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnode; k += (int64_t)gridDim.x * 256)
    keep[label[node2cell[k]]] = 1;
}
__global__ __launch_bounds__(256) void comp_compact_kernel(int64_t nnode, const int* __restrict__ node2cell,
                                                           const int* __restrict__ label, const int* __restrict__ dense,
                                                           int* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnode; k += (int64_t)gridDim.x * 256)
    out[k] = dense[label[node2cell[k]]];
}

// =====================================================================================================================
// Rasters WITH short-circuit polygons (construct_node_map with a polymap, src/raster/pairwise.jl:276-301): every cell
// of a polygon -- NODATA cells included, as in the reference -- shares the node of the polygon's first valid cell
// (column-major order); parallel edges created by the merge are summed (sparse(I, J, V) + a + a', pairwise.jl:357-361)
// and edges inside a polygon vanish (laplacian! zeroes the stored diagonal, core.jl:608-634).
//
// Device scheme. k = column-major cell index j*R + i.
//   1. rep[p]   = min k over the valid cells of polygon p (atomicMin)            -> the polygon's representative cell
//   2. label[k] = rep[poly] for polygon cells, k for other valid cells, -1 otherwise; flag[k] = 1 where label[k] == k
//      node ids = exclusive scan of the flags (= relabel! of the reference: order of the surviving original ids)
//   3. rows of ordinary cells: <= 8 neighbours, merged by a tiny insertion sort (several neighbours may be cells of the
//      same polygon), written by one thread per cell like raster_fill_kernel
//   4. rows of polygon nodes: every (member cell, outside neighbour) pair becomes a candidate keyed by
//      (column, lower cell index, direction); a segmented bitonic sort (one workgroup per polygon) orders them, runs of
//      equal columns are summed in key order. Both rows of a coupling sum the same cell pairs in the same canonical
//      order, so the matrix is bit-symmetric.
// Only positive polygon ids are polygons; ids must be < 2^26.
static const int kMaxPolyId = 1 << 26;

__global__ __launch_bounds__(256) void poly_max_kernel(int64_t ncells, const int* __restrict__ poly, int* __restrict__ out) {
  int m = 0;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * 256) m = max(m, poly[c]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

template <class T>
__global__ __launch_bounds__(256) void poly_rep_kernel(int R, int C, const T* __restrict__ cond,
                                                       const int* __restrict__ poly, int* __restrict__ rep) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int i = (int)(k % R), j = (int)(k / R);
    const int p = poly[(size_t)i * C + j];
    if (p > 0 && cond[(size_t)i * C + j] > T(0)) atomicMin(&rep[p], (int)k);
  }
}

// label / flag per cell (column-major), polygon flags for the dense polygon numbering
template <class T>
__global__ __launch_bounds__(256) void poly_label_kernel(int R, int C, const T* __restrict__ cond,
                                                         const int* __restrict__ poly, const int* __restrict__ rep,
                                                         int* __restrict__ label, int* __restrict__ flag) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int i = (int)(k % R), j = (int)(k / R);
    const int p = poly[(size_t)i * C + j];
    const bool valid = cond[(size_t)i * C + j] > T(0);
    const bool merged = p > 0 && rep[p] != 0x7fffffff;
    const int lab = merged ? rep[p] : (valid ? (int)k : -1);
    label[k] = lab;
    flag[k] = lab == (int)k ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void poly_present_kernel(int npolyids, const int* __restrict__ rep, int* __restrict__ present) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p <= npolyids; p += gridDim.x * 256)
    present[p] = (p > 0 && rep[p] != 0x7fffffff) ? 1 : 0;
}

// node of every cell (column-major, -1 = none), node map for the host (row-major, 1-based), node coordinates, and the
// polygon (dense index, -1 = ordinary) of every NODE
__global__ __launch_bounds__(256) void poly_node_kernel(int R, int C, const int* __restrict__ poly,
                                                        const int* __restrict__ rep, const int* __restrict__ label,
                                                        const int* __restrict__ scan, const int* __restrict__ pdense,
                                                        int* __restrict__ node, int* __restrict__ nodemap,
                                                        int* __restrict__ nrow, int* __restrict__ ncol,
                                                        int* __restrict__ node_poly, int* __restrict__ poly_node) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int i = (int)(k % R), j = (int)(k / R);
    const int lab = label[k];
    const int nd = lab >= 0 ? scan[lab] : -1;
    node[k] = nd;
    nodemap[(size_t)i * C + j] = nd + 1;
    if (lab == (int)k) {  // this cell carries the node: its coordinates seed the aggregation
      nrow[nd] = i;
      ncol[nd] = j;
      const int p = poly[(size_t)i * C + j];
      const bool merged = p > 0 && rep[p] != 0x7fffffff;
      node_poly[nd] = merged ? pdense[p] : -1;
      if (merged) poly_node[pdense[p]] = nd;
    }
  }
}

// the (up to 8) neighbours of cell (i, j) in ascending column-major order; calls f(ii, jj, diagonal)
template <class F>
__device__ __forceinline__ void for_each_neighbour(int i, int j, int R, int C, int four, F f) {
  for (int dj = -1; dj <= 1; ++dj) {
    const int jj = j + dj;
    if (jj < 0 || jj >= C) continue;
    for (int di = -1; di <= 1; ++di) {
      const int ii = i + di;
      if (ii < 0 || ii >= R || (di == 0 && dj == 0)) continue;
      const bool diag = di != 0 && dj != 0;
      if (diag && four) continue;
      f(ii, jj, diag);
    }
  }
}

// Ordinary cells: merged row (columns ascending, duplicates summed in neighbour order). FILL = false: row length only.
template <class T, bool FILL>
__global__ __launch_bounds__(256) void poly_cell_rows_kernel(int R, int C, int four, int avg_res,
                                                             const T* __restrict__ cond, const int* __restrict__ node,
                                                             const int* __restrict__ node_poly, const int* __restrict__ label,
                                                             int* __restrict__ rowlen, const int* __restrict__ rp,
                                                             int* __restrict__ ci, T* __restrict__ va) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    if (label[k] != (int)k) continue;          // not a node-carrying cell
    const int me = node[k];
    if (node_poly[me] >= 0) continue;          // polygon rows are assembled from the candidate lists
    const int i = (int)(k % R), j = (int)(k / R);
    const double g0 = (double)cond[(size_t)i * C + j];
    int cols[8];
    double w[8];
    int m = 0;
    for_each_neighbour(i, j, R, C, four, [&](int ii, int jj, bool diag) {
      const int nb = node[(int64_t)jj * R + ii];
      if (nb < 0) return;
      const double wt = raster_edge(g0, (double)cond[(size_t)ii * C + jj], diag, avg_res != 0);
      int q = 0;
      while (q < m && cols[q] != nb) ++q;
      if (q < m) {
        w[q] += wt;  // another cell of the same polygon: neighbour order = ascending cell index
      } else {
        cols[m] = nb;
        w[m] = wt;
        ++m;
      }
    });
    if (!FILL) {
      rowlen[me] = m + 1;
      continue;
    }
    // insertion sort by column, diagonal inserted in place
    for (int a = 1; a < m; ++a) {
      const int c = cols[a];
      const double x = w[a];
      int b = a - 1;
      while (b >= 0 && cols[b] > c) {
        cols[b + 1] = cols[b];
        w[b + 1] = w[b];
        --b;
      }
      cols[b + 1] = c;
      w[b + 1] = x;
    }
    double deg = 0.0;
    for (int a = 0; a < m; ++a) deg += w[a];
    int o = rp[me];
    bool dput = false;
    for (int a = 0; a < m; ++a) {
      if (!dput && cols[a] > me) {
        ci[o] = me;
        va[o++] = (T)deg;
        dput = true;
      }
      ci[o] = cols[a];
      va[o++] = (T)(-w[a]);
    }
    if (!dput) {
      ci[o] = me;
      va[o] = (T)deg;
    }
  }
}

// Polygon member cells: one candidate per (member, outside neighbour) pair. COUNT: cnt[polygon] += 1; else fill at the
// polygon's cursor. key = column << 33 | lower cell index << 2 | direction (0: +1, 1: +R-1, 2: +R, 3: +R+1).
template <class T, bool COUNT>
__global__ __launch_bounds__(256) void poly_candidates_kernel(int R, int C, int four, int avg_res,
                                                              const T* __restrict__ cond, const int* __restrict__ node,
                                                              const int* __restrict__ node_poly,
                                                              const int* __restrict__ label, int* __restrict__ cnt,
                                                              const int64_t* __restrict__ seg_off,
                                                              unsigned long long* __restrict__ key,
                                                              double* __restrict__ val) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int me = node[k];
    if (me < 0) continue;
    const int pd = node_poly[me];
    if (pd < 0) continue;
    const int i = (int)(k % R), j = (int)(k / R);
    const double g0 = (double)cond[(size_t)i * C + j];
    for_each_neighbour(i, j, R, C, four, [&](int ii, int jj, bool diag) {
      const int64_t kn = (int64_t)jj * R + ii;
      const int nb = node[kn];
      if (nb < 0 || nb == me) return;
      const int pos = atomicAdd(&cnt[pd], 1);
      if (COUNT) return;
      const int64_t lo = kn < k ? kn : k, d = (kn < k ? k - kn : kn - k);
      const unsigned long long dir = d == 1 ? 0ull : d == (int64_t)R - 1 ? 1ull : d == (int64_t)R ? 2ull : 3ull;
      key[seg_off[pd] + pos] = ((unsigned long long)nb << 33) | ((unsigned long long)lo << 2) | dir;
      val[seg_off[pd] + pos] = raster_edge(g0, (double)cond[(size_t)ii * C + jj], diag, avg_res != 0);
    });
  }
}

// padded (power of two) segment lengths
__global__ __launch_bounds__(256) void poly_pad_kernel(int npoly, const int* __restrict__ cnt, int* __restrict__ padded) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p <= npoly; p += gridDim.x * 256) {
    int L = 1;
    const int c = p < npoly ? cnt[p] : 0;
    while (L < c) L <<= 1;
    padded[p] = p < npoly ? (c > 0 ? L : 0) : 0;
  }
}

// one workgroup per polygon: bitonic sort of its (padded) segment by key
__global__ __launch_bounds__(256) void poly_sort_kernel(const int64_t* __restrict__ seg_off, unsigned long long* __restrict__ key,
                                                        double* __restrict__ val) {
  const int64_t base = seg_off[blockIdx.x];
  const int64_t L = seg_off[blockIdx.x + 1] - base;
  for (int64_t k2 = 2; k2 <= L; k2 <<= 1)
    for (int64_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int64_t e = threadIdx.x; e < L; e += 256) {
        const int64_t f = e ^ j2;
        if (f > e) {
          const bool up = (e & k2) == 0;
          const unsigned long long a = key[base + e], b = key[base + f];
          if ((a > b) == up) {
            key[base + e] = b;
            key[base + f] = a;
            const double t = val[base + e];
            val[base + e] = val[base + f];
            val[base + f] = t;
          }
        }
      }
      __syncthreads();
    }
}

// unique columns per polygon (COUNT) / write the merged row (one workgroup per polygon, thread 0 walks the sorted
// segment: perimeter-sized, and the order of the additions is the key order)
template <class T, bool COUNT>
__global__ __launch_bounds__(64) void poly_rows_kernel(const int64_t* __restrict__ seg_off, const int* __restrict__ cnt,
                                                       const unsigned long long* __restrict__ key,
                                                       const double* __restrict__ val, const int* __restrict__ poly_node,
                                                       int* __restrict__ rowlen, const int* __restrict__ rp,
                                                       int* __restrict__ ci, T* __restrict__ va) {
  if (threadIdx.x != 0) return;
  const int pd = blockIdx.x;
  const int64_t base = seg_off[pd];
  const int m = cnt[pd];
  const int me = poly_node[pd];
  if (COUNT) {
    int u = 0;
    for (int e = 0; e < m; ++e)
      if (e == 0 || (key[base + e] >> 33) != (key[base + e - 1] >> 33)) ++u;
    rowlen[me] = u + 1;
    return;
  }
  int o = rp[me];
  bool dput = false;
  int e = 0;
  double degree = 0.0;
  while (e < m) {
    const int col = (int)(key[base + e] >> 33);
    double s = 0.0;
    while (e < m && (int)(key[base + e] >> 33) == col) s += val[base + e++];
    if (!dput && col > me) {
      ci[o] = me;
      va[o++] = T(0);  // degree written after the walk
      dput = true;
    }
    ci[o] = col;
    va[o++] = (T)(-s);
    degree += s;
  }
  if (!dput) {
    ci[o] = me;
    va[o] = T(0);
  }
  // diagonal = sum of the merged off-diagonal weights (laplacian!: sum_off_diag)
  for (int q = rp[me]; q < rp[me + 1]; ++q)
    if (ci[q] == me) va[q] = (T)degree;
}

}  // namespace csgpu
