// lattice_setup.h -- level 0 of the hierarchy of a raster built straight from the lattice form: no CSR matrix, no
// SpGEMM, no transpose.
//
// GPU counterpart, for the matrices the reference's raster path produces (construct_graph, src/raster/pairwise.jl:
// 316-362; laplacian!, src/core.jl:608-634), of the first coarsening step of `smoothed_aggregation(matrix; ...)`
// (reference call site src/core.jl:164-167; algorithm restated in SURVEY.md section 2.3 and amg_setup.h):
//
//   raster -> A in lattice form (5 values per cell, stencil.h)                          raster_dia_kernel
//   row statistics (diagonal, sum |a_ij|, Gershgorin bound)                             dia_stats_kernel
//   aggregates = regular 3x3 tiles (+ piece analysis on rasters with NODATA cells)      lattice_tile_kernel, dia_pieces_kernel
//   T (tentative prolongator), P = T - w_p Dl^-1 A T                                     lattice_p_kernel      [n][9]
//   A P and Q = P - w D^-1 A P                                                           lattice_ap_q_kernel   [n][9]
//   A_c = P^T (A P)  (Galerkin operator of level 1, CSR)                                 lattice_galerkin_kernel
//
// Every operator of the level is index-free: row (i, j) of P, A P and Q only reaches the 3 x 3 block of tiles around the
// cell's own tile (LatticeQ in common.h), and A_c couples a tile to its neighbours within two tiles (one on all-valid
// rasters). The general CSR pipeline of amg_setup.h produces the same operators (tests compare the two); it stays in
// charge of matrices handed over in CSR form, of rasters with polygons and of every level below this one.
//
// Why: at 10000 x 10000 the CSR pipeline spends ~180 ms of device time and ~15 GB of transient memory on level 0 (CSR
// build 45 ms, lattice detection 11 ms, A T / A P / R A P 95 ms, transpose 25 ms), the lattice one ~1/4 of that; and a
// raster above 2^31 / 9 = 238 M cells has no int32 CSR form at all (the reference documents 437 M cells,
// docs/src/compute.md:3).
#pragma once
#include "amg_setup.h"
#include "lattice.h"
#include "raster.h"
#include "enrich.h"

namespace csgpu {

// k-th entry (ascending column order, k = 0..8) of row i of a lattice matrix: column j (-1 when outside the matrix)
// and value (0 where the entry is absent). U: storage precision, returned as double-convertible U.
template <class U>
__device__ __forceinline__ U dia_row_entry(const U* __restrict__ rows, int64_t n, int R, int64_t i, int k, int64_t& j) {
  switch (k) {
    case 0: j = i - R - 1; return j >= 0 ? rows[j * 5 + 4] : U(0);
    case 1: j = i - R; return j >= 0 ? rows[j * 5 + 3] : U(0);
    case 2: j = i - R + 1; return j >= 0 ? rows[j * 5 + 2] : U(0);
    case 3: j = i - 1; return j >= 0 ? rows[j * 5 + 1] : U(0);
    case 4: j = i; return rows[i * 5 + 0];
    case 5: j = i + 1; return j < n ? rows[i * 5 + 1] : U(0);
    case 6: j = i + R - 1; return j < n ? rows[i * 5 + 2] : U(0);
    case 7: j = i + R; return j < n ? rows[i * 5 + 3] : U(0);
    default: j = i + R + 1; return j < n ? rows[i * 5 + 4] : U(0);
  }
}

// ---- raster -> lattice form ------------------------------------------------------------------------------------------
// One thread per cell (column-major id = j*R + i). Same arithmetic as raster_fill_kernel (raster.h): weights in double,
// diagonal = sum of the weights of the valid neighbours (+ finite ground), off-diagonals -w, rounded to T once.
// NODATA cells (cell space): all five values 0 here; raster_dia_finish_kernel sets their diagonal to 1.
// part[block] = sum of squares of the entries the CSR form would store (diagonal once, every coupling in both rows);
// cnt[block] = number of those entries (the nnz the caller reports).
template <class T>
__global__ __launch_bounds__(256) void raster_dia_kernel(int R, int C, int four, int avg_res, const T* __restrict__ cond,
                                                         const T* __restrict__ ground, T* __restrict__ rows,
                                                         T* __restrict__ ground_node, double* __restrict__ part,
                                                         unsigned long long* __restrict__ cnt, int colmajor = 0) {
  __shared__ double sm[4];
  __shared__ unsigned long long smc[4];
  const int64_t n = (int64_t)R * C;
  // (colmajor: cond / ground are the transposed copies -- cell (i, j) at j*R + i; same values, same order of operations)
  const size_t si = colmajor ? (size_t)1 : (size_t)C, sj = colmajor ? (size_t)R : (size_t)1;
  double ss = 0.0;
  unsigned long long c = 0;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    int i, j;
    cell_rc(id, R, i, j);
    const double g0 = (double)cond[(size_t)i * si + (size_t)j * sj];
    T out[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (g0 > 0.0) {
      double deg = 0.0;
      // neighbours in column-major order, like the CSR row (raster_fill_kernel): the degree is summed in that order
      for (int dj = -1; dj <= 1; ++dj) {
        const int jj = j + dj;
        if (jj < 0 || jj >= C) continue;
        for (int di = -1; di <= 1; ++di) {
          const int ii = i + di;
          if (ii < 0 || ii >= R || (di == 0 && dj == 0)) continue;
          const bool diag = (di != 0 && dj != 0);
          if (diag && four) continue;
          const double g1 = (double)cond[(size_t)ii * si + (size_t)jj * sj];
          if (!(g1 > 0.0)) continue;
          const double w = raster_edge(g0, g1, diag, avg_res != 0);
          deg += w;
          const T v = (T)(-w);
          ss += (double)v * (double)v;
          ++c;
          if (dj == 0 && di == 1) out[1] = v;
          else if (dj == 1 && di == -1) out[2] = v;
          else if (dj == 1 && di == 0) out[3] = v;
          else if (dj == 1 && di == 1) out[4] = v;
        }
      }
      const double gnd = ground ? (double)ground[(size_t)i * si + (size_t)j * sj] : 0.0;
      out[0] = (T)(deg + gnd);
      ss += (double)out[0] * (double)out[0];
      ++c;
      if (ground_node) ground_node[id] = (T)gnd;
    } else if (ground_node) {
      ground_node[id] = T(0);
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) rows[id * 5 + s] = out[s];
  }
  ss = block_sum_256(ss, sm);
  c = block_sum_256(c, smc);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ss;
    cnt[blockIdx.x] = c;
  }
}

// regularisation nzval .+= eps(T) * norm(nzval) of the entries the CSR form stores (src/core.jl:161): the diagonal of
// every valid cell and every coupling that exists (value != 0: conductances are positive); NODATA diagonal = 1;
// size0[cell] = 1 for a cell with a node, 0 otherwise (may be null)
template <class T>
__global__ __launch_bounds__(256) void raster_dia_finish_kernel(int R, int C, const T* __restrict__ cond, T* __restrict__ rows,
                                                                const double* __restrict__ part, int nparts, double eps,
                                                                long long* __restrict__ size0, int colmajor = 0) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = block_sum_256(s, sm);
  const T shift = (T)(eps * sqrt(s));
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    bool valid;
    if (colmajor) {
      valid = cond[id] > T(0);
    } else {
      int i, j;
      cell_rc(id, R, i, j);
      valid = cond[(size_t)i * C + j] > T(0);
    }
    if (size0) size0[id] = valid ? 1 : 0;
    if (!valid) {
      rows[id * 5] = T(1);
      continue;
    }
    if (eps != 0.0) {
      rows[id * 5] += shift;
#pragma unroll
      for (int q = 1; q < 5; ++q)
        if (rows[id * 5 + q] != T(0)) rows[id * 5 + q] += shift;
    }
  }
}

// ---- cell space from a CSR matrix with raster coordinates (the Julia host path) ------------------------------------------
// A host that built its graph itself (construct_graph, src/raster/pairwise.jl:316-362) hands over a CSR Laplacian in the
// compact numbering plus the raster cell of every node (csgpu_opts.node_row / node_col). When every node sits on a cell
// of its own and every coupling joins lattice neighbours (no polygons), the matrix is scattered into the lattice form of
// the R x C raster spanned by the coordinates -- the same cell-space matrix csgpu_raster_setup builds from the raster.
__global__ __launch_bounds__(256) void coord_range_kernel(int n, const int* __restrict__ row, const int* __restrict__ col,
                                                          int* __restrict__ mm) {  // mm = {max row, max col, min row, min col}
  int r = -0x7fffffff, c = -0x7fffffff, r0 = 0x7fffffff, c0 = 0x7fffffff;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    r = max(r, row[i]);
    c = max(c, col[i]);
    r0 = min(r0, row[i]);
    c0 = min(c0, col[i]);
  }
  atomicMax(&mm[0], r);
  atomicMax(&mm[1], c);
  atomicMin(&mm[2], r0);
  atomicMin(&mm[3], c0);
}

// node2cell[i] = column-major cell id inside the bounding box (origin r0, c0; height R); cell2node[cell] = i + 1 (0 = no
// node); bad when two nodes share a cell
__global__ __launch_bounds__(256) void csr_cells_kernel(int n, int R, int r0, int c0, const int* __restrict__ row,
                                                        const int* __restrict__ col, int* __restrict__ node2cell,
                                                        int* __restrict__ cell2node, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t c = (int64_t)(col[i] - c0) * R + (row[i] - r0);
    node2cell[i] = (int)c;
    if (atomicCAS(&cell2node[c], 0, i + 1) != 0) atomicOr(bad, 2);
  }
}

// rows[cell] of the lattice form from the CSR rows of the nodes (upper triangle + diagonal; the matrix is symmetric);
// bad when a coupling does not join lattice neighbours
template <class T>
__global__ __launch_bounds__(256) void csr_to_cell_dia_kernel(int n, int R, const int* __restrict__ rp, const int* __restrict__ ci,
                                                              const T* __restrict__ va, const int* __restrict__ row,
                                                              const int* __restrict__ col, const int* __restrict__ node2cell,
                                                              T* __restrict__ rows, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t c = node2cell[i];
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int j = ci[k];
      if (j == i) {
        rows[c * 5] = va[k];
        continue;
      }
      const int dr = row[j] - row[i], dc = col[j] - col[i];
      if (dr < -1 || dr > 1 || dc < -1 || dc > 1) {
        atomicOr(bad, 4);
        continue;
      }
      const int d = dc * R + dr;  // 1, R-1, R, R+1 or their negatives (never 0: two nodes never share a cell)
      if (d < 0) continue;        // (the mirror image lives in row j)
      const int slot = d == 1 ? 1 : (d == R - 1 ? 2 : (d == R ? 3 : 4));
      rows[c * 5 + slot] = va[k];
    }
  }
}

// The same scatter for a BLOCK of CSR rows (nodes i0 .. i0 + nloc) whose row pointers were rebased to the block (rp[0] = 0):
// the streamed set-up of host matrices with 2^31 stored entries and more (csgpu.hip, setup_from_host_streamed), which never
// holds the whole CSR form on the device. Node ids, coordinates and node2cell are global.
template <class T>
__global__ __launch_bounds__(256) void csr_block_to_cell_dia_kernel(int i0, int nloc, int R, const int* __restrict__ rp,
                                                                    const int* __restrict__ ci, const T* __restrict__ va,
                                                                    const int* __restrict__ row, const int* __restrict__ col,
                                                                    const int* __restrict__ node2cell, T* __restrict__ rows,
                                                                    int* __restrict__ bad) {
  for (int il = blockIdx.x * 256 + threadIdx.x; il < nloc; il += gridDim.x * 256) {
    const int i = i0 + il;
    const int64_t c = node2cell[i];
    bool diag = false;
    for (int k = rp[il]; k < rp[il + 1]; ++k) {
      const int j = ci[k];
      if (j == i) {
        rows[c * 5] = va[k];
        diag = true;
        continue;
      }
      const int dr = row[j] - row[i], dc = col[j] - col[i];
      if (dr < -1 || dr > 1 || dc < -1 || dc > 1) {
        atomicOr(bad, 4);
        continue;
      }
      const int d = dc * R + dr;
      if (d < 0) continue;
      const int slot = d == 1 ? 1 : (d == R - 1 ? 2 : (d == R ? 3 : 4));
      rows[c * 5 + slot] = va[k];
    }
    if (!diag) atomicOr(bad, 16);  // (a Laplacian row without its diagonal: not a matrix this path can take)
  }
}

// all-valid raster handed over as a matrix: the handle keeps the caller's numbering only if it IS the column-major one
__global__ __launch_bounds__(256) void identity_numbering_kernel(int n, const int* __restrict__ node2cell, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (node2cell[i] != i) atomicOr(bad, 8);
}

// cells without a node: identity rows; size0 = 1 for a cell with a node
template <class T>
__global__ __launch_bounds__(256) void cell_identity_kernel(int64_t ncells, const int* __restrict__ cell2node, T* __restrict__ rows,
                                                            long long* __restrict__ size0) {
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncells; c += (int64_t)gridDim.x * 256) {
    const bool node = cell2node[c] != 0;
    size0[c] = node ? 1 : 0;
    if (!node) rows[c * 5] = T(1);
  }
}

// node numbering of the reference for a cell-space / all-valid raster: nodemap (row-major, 1-based node id), cellmap
// (row-major, 1-based row id of the device matrix), node2cell, cell2node (see Solver in csgpu.hip). node = exclusive scan
// of the valid flags in column-major order.
template <class T>
__global__ __launch_bounds__(256) void raster_maps_kernel(int R, int C, const T* __restrict__ cond, const int* __restrict__ node,
                                                          int* __restrict__ nodemap, int* __restrict__ cellmap,
                                                          int* __restrict__ node2cell, int* __restrict__ cell2node,
                                                          int colmajor = 0) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    int i, j;
    cell_rc(id, R, i, j);
    const bool valid = (colmajor ? cond[id] : cond[(size_t)i * C + j]) > T(0);
    nodemap[(size_t)i * C + j] = valid ? node[id] + 1 : 0;
    if (cellmap) cellmap[(size_t)i * C + j] = valid ? (int)id + 1 : 0;
    if (node2cell && valid) node2cell[node[id]] = (int)id;
    if (cell2node) cell2node[id] = valid ? node[id] + 1 : 0;
  }
}

// ---- row statistics (row_stats_kernel + dinv_kernel of amg_setup.h on the lattice form) -------------------------------
template <class U, class T>
__global__ __launch_bounds__(256) void dia_stats_kernel(int64_t n, int R, const U* __restrict__ rows, T* __restrict__ labs,
                                                        double* __restrict__ part_max, double* __restrict__ part_dmax) {
  __shared__ double sm[4], smd[4];
  double mx = 0.0, dmx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T l = T(0), d = T(0);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      int64_t j;
      const T v = (T)dia_row_entry(rows, n, R, i, k, j);
      if (k == 4) d = v;
      l += v < T(0) ? -v : v;
    }
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

// dinv_kernel of amg_setup.h (see there: no pivot on an isolated row below the floor) on the lattice form
template <class U, class T>
__global__ __launch_bounds__(256) void dia_dinv_kernel(int64_t n, const U* __restrict__ rows, const T* __restrict__ labs,
                                                       double floor_, T* __restrict__ dinv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const T d = (T)rows[i * 5];
    const bool isolated = labs[i] <= (d < T(0) ? -d : d);
    dinv[i] = (d > T(0) && !(isolated && (double)d <= floor_)) ? T(1) / d : T(0);
  }
}

// ---- aggregates: the regular tiles -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lattice_tile_kernel(int64_t n, int R, int Rc, int Cc, int* __restrict__ agg,
                                                           int* __restrict__ crow, int* __restrict__ ccol) {
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const int I = lat_tile(i, Rc), J = lat_tile(j, Cc);
    const int a = J * Rc + I;
    agg[id] = a;
    if (i == 3 * I && j == 3 * J) {  // one writer per tile
      crow[a] = I;
      ccol[a] = J;
    }
  }
}

// a coupling below theta * sqrt(a_ii a_jj) (TileStrength, amg_setup.h); round32: the hierarchy computes on the values
// rounded to fp32 (the CSR pipeline applies the test to its fp32 copy of the matrix: same decisions)
template <class U>
__device__ __forceinline__ bool dia_weak(U a, U di, U dj, double theta2, int round32) {
  const double x = round32 ? (double)(float)a : (double)a;
  const double p = round32 ? (double)(float)di : (double)di, q = round32 ? (double)(float)dj : (double)dj;
  return x * x < theta2 * p * q;
}

// piece analysis of amg_setup.h (tile_pieces_kernel) on the lattice form; see the comment there
template <class U, int PASS>
__global__ __launch_bounds__(256) void dia_pieces_kernel(int R, int C, int Rc, int Cc, const U* __restrict__ rows,
                                                         long long* __restrict__ size_f, signed char* __restrict__ piece,
                                                         signed char* __restrict__ mainlab, int* __restrict__ agg,
                                                         double theta2, int round32, int stride) {
  const int ntiles = Rc * Cc;
  const int64_t n = (int64_t)R * C;
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
      if (nvalid > 0 && (nvalid < h * w || theta2 > 0.0)) {  // (a full tile is connected: adjacent valid cells are always coupled)
        for (int sweep = 0; sweep < 16; ++sweep) {
          bool changed = false;
          for (int kc = 0; kc < w; ++kc)
            for (int kr = 0; kr < h; ++kr) {
              int& me = lab[kc * h + kr];
              if (me < 0) continue;
              const int64_t cell = (int64_t)(c0 + kc) * R + r0 + kr;
              for (int k = 0; k < 9; ++k) {
                if (k == 4) continue;
                int64_t nb;
                const U av = dia_row_entry(rows, n, R, cell, k, nb);
                if (av == U(0)) continue;
                if (theta2 > 0.0 && dia_weak(av, rows[cell * 5], rows[nb * 5], theta2, round32)) continue;
                const int ni = (int)(nb % R) - r0, nj = (int)(nb / R) - c0;
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
      int best = -1, bestcnt = 0;
      for (int q = 0; q < h * w; ++q) {
        int cntq = 0;
        for (int k = 0; k < h * w; ++k) cntq += lab[k] == q ? 1 : 0;
        if (cntq > bestcnt) {
          bestcnt = cntq;
          best = q;
        }
      }
      mainlab[tile] = (signed char)best;
      for (int kc = 0; kc < w; ++kc)
        for (int kr = 0; kr < h; ++kr) piece[(int64_t)(c0 + kc) * R + r0 + kr] = (signed char)lab[kc * h + kr];
    } else {
      // every cell outside the tile's main piece joins the tile of the main-piece cell it is most strongly coupled to
      // (ties: the first in column order); without such a coupling it weighs 0
      const int mainq = mainlab[tile];
      for (int kc = 0; kc < w; ++kc)
        for (int kr = 0; kr < h; ++kr) {
          const int64_t cell = (int64_t)(c0 + kc) * R + r0 + kr;
          if (piece[cell] < 0 || piece[cell] == mainq) continue;
          int target = -1;
          double best = 0.0;
          for (int k = 0; k < 9; ++k) {
            if (k == 4) continue;
            int64_t nb;
            const U v = dia_row_entry(rows, n, R, cell, k, nb);
            if (v == U(0)) continue;
            if (theta2 > 0.0 && dia_weak(v, rows[cell * 5], rows[nb * 5], theta2, round32)) continue;
            const int ni = (int)(nb % R), nj = (int)(nb / R);
            if (ni >= r0 && ni < r1 && nj >= c0 && nj < c1) continue;  // inside this tile
            const int nt = lat_tile(nj, Cc) * Rc + lat_tile(ni, Rc);
            if (piece[nb] < 0 || piece[nb] != mainlab[nt]) continue;
            const double a = fabs((double)v);
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

// pass 3 of the piece analysis (rounds 1, 2, ...; see tile_orphans_kernel in amg_setup.h) on the lattice form. An orphan
// may only adopt an aggregate that keeps Q inside the 3 x 3 block of tiles of every row that sees it: rows within two
// cells of the orphan lie in the tile above when the orphan sits in the first two rows of its tile, in the tile below
// when it sits in the last two; the adopted tile must be within one tile of all of them (same for columns).
template <class U>
__global__ __launch_bounds__(256) void dia_orphans_kernel(int R, int C, int Rc, int Cc, const U* __restrict__ rows,
                                                          long long* __restrict__ size_f, signed char* __restrict__ piece,
                                                          const signed char* __restrict__ mainlab, int* __restrict__ agg,
                                                          int round, int last, double theta2, int round32) {
  const int64_t n = (int64_t)R * C;
  for (int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x; cell < n; cell += (int64_t)gridDim.x * 256) {
    const int pc = piece[cell];
    if (pc < 0 || pc >= kPieceAttached) continue;
    const int i = (int)(cell % R), j = (int)(cell / R);
    const int I = lat_tile(i, Rc), J = lat_tile(j, Cc);
    if (pc == mainlab[J * Rc + I]) continue;
    int r0, r1, c0, c1;
    tile_extent(I, Rc, R, r0, r1);
    tile_extent(J, Cc, C, c0, c1);
    const bool up = i - r0 <= 1 && I > 0, dn = r1 - 1 - i <= 1 && I < Rc - 1;
    const bool lf = j - c0 <= 1 && J > 0, rt = c1 - 1 - j <= 1 && J < Cc - 1;
    const int Ilo = dn ? I : I - 1, Ihi = up ? I : I + 1, Jlo = rt ? J : J - 1, Jhi = lf ? J : J + 1;
    int target = -1;
    double best = 0.0;
    bool coupled = false;
    for (int k = 0; k < 9; ++k) {
      if (k == 4) continue;
      int64_t nb;
      const U v = dia_row_entry(rows, n, R, cell, k, nb);
      if (v == U(0)) continue;
      coupled = true;
      if (theta2 > 0.0 && dia_weak(v, rows[cell * 5], rows[nb * 5], theta2, round32)) continue;
      const int pn = piece[nb];
      if (pn < kPieceAttached || pn >= kPieceAttached + round) continue;  // attached in an EARLIER round (deterministic)
      const int ta = agg[nb], tI = ta % Rc, tJ = ta / Rc;
      if (tI < Ilo || tI > Ihi || tJ < Jlo || tJ > Jhi) continue;
      const double a = fabs((double)v);
      if (a > best) {
        best = a;
        target = ta;
      }
    }
    if (target >= 0) {
      agg[cell] = target;
      piece[cell] = (signed char)(kPieceAttached + round);
    } else if (last && !coupled) {
      size_f[cell] = 0;  // an island of one cell: no aggregate (a COUPLED cell left over keeps its tile's)
    }
  }
}

// t[i] = sqrt(size_f[i] / size_c[agg[i]])   (tentative_kernel of amg_setup.h; size_f null = all ones)
template <class T>
__global__ __launch_bounds__(256) void lattice_tentative_kernel(int64_t n, const int* __restrict__ agg,
                                                                const long long* __restrict__ size_f,
                                                                const unsigned long long* __restrict__ size_c,
                                                                T* __restrict__ t) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double sf = size_f ? (double)size_f[i] : 1.0;
    const double sc = (double)size_c[agg[i]];
    t[i] = sc > 0.0 ? (T)sqrt(sf / sc) : T(0);
  }
}

__global__ __launch_bounds__(256) void lattice_sizes_kernel(int64_t n, const int* __restrict__ agg,
                                                            const long long* __restrict__ size_f,
                                                            unsigned long long* __restrict__ size_c) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    atomicAdd(&size_c[agg[i]], (unsigned long long)(size_f ? size_f[i] : 1));
}

// slot (0..8) of aggregate a in the 3 x 3 block of tiles around tile (I, J); -1 when it lies outside
__device__ __forceinline__ int lat_slot(int a, int Rc, int I, int J) {
  const int dI = a % Rc - I, dJ = a / Rc - J;
  if (dI < -1 || dI > 1 || dJ < -1 || dJ > 1) return -1;
  return (dJ + 1) * 3 + (dI + 1);
}

// P = T - (w_p / labs) A T in index-free form: pl[i][slot] = P[i, aggregate at `slot` of the block around tile(i)].
// Same arithmetic as spgemm_tentative_kernel + smooth_prolongator_kernel (amg_setup.h): products summed per aggregate in
// column order in T, the smoothing step in double.
template <class U, class T>
__global__ __launch_bounds__(256) void lattice_p_kernel(int64_t n, int R, int Rc, int Cc, const U* __restrict__ rows,
                                                        const int* __restrict__ agg, const T* __restrict__ t,
                                                        const T* __restrict__ labs, double omega_p, T* __restrict__ pl,
                                                        int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int ri, ci;
    cell_rc(i, R, ri, ci);
    const int I = lat_tile(ri, Rc), J = lat_tile(ci, Cc);
    T acc[9];
    bool has[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      acc[s] = T(0);
      has[s] = false;
    }
    // (all loads of the row first -- entries, then aggregates and tentative values of the neighbours that exist -- so that
    // their latencies overlap instead of following one another behind the branches below)
    T av[9], tj[9];
    int aj[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      int64_t j;
      av[k] = (T)dia_row_entry(rows, n, R, i, k, j);
      const bool in = j >= 0 && j < n;
      aj[k] = in ? agg[j] : 0;
      tj[k] = in ? t[j] : T(0);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const T a = av[k];
      if (a == T(0) && k != 4) continue;  // (the diagonal is always a stored entry)
      const int slot = lat_slot(aj[k], Rc, I, J);
      if (slot < 0) {
        atomicOr(bad, 1);
        continue;
      }
      const T v = a * tj[k];
#pragma unroll
      for (int s = 0; s < 9; ++s)
        if (s == slot) {
          acc[s] += v;
          has[s] = true;
        }
    }
    const double l = (double)labs[i];
    const double w = l != 0.0 ? omega_p / l : 0.0;
    const int own = lat_slot(agg[i], Rc, I, J);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      double v = 0.0;
      if (has[s]) {
        v = -w * (double)acc[s];
        if (s == own) v += (double)t[i];
      }
      pl[i * 9 + s] = (T)v;
    }
  }
}

// A P and Q = P - w D^-1 A P, both index-free ([n][9]); ap may be null (only Q wanted).
// A workgroup owns 256 consecutive cells. The rows of `pl` it needs -- the cells themselves and their eight lattice
// neighbours -- are three contiguous runs of 258 rows (raster columns j-1, j, j+1), staged in LDS with coalesced loads:
// read directly, every lane would fetch nine 72-byte rows at a 72-byte lane stride, 81 load instructions that each touch
// 36 cache lines (measured at 1e8 cells, fp64: 52 ms -- the largest kernel of the setup; staged: see profiles/r3_setup_*).
// NT (cells per workgroup): the kernel is three phases between barriers (stage, multiply, store through LDS), so what hides
// one workgroup's loads is the NEXT workgroup on the CU -- at 256 cells the 56 KB of fp64 runs leave room for two. Measured
// at 10000^2 fp64 (round 6): 128 or 64 cells per workgroup change nothing (22.2 / 23.8 / 24 ms) -- occupancy is not the bound.
template <class U, class T, int NT = 256>
__global__ __launch_bounds__(NT) void lattice_ap_q_kernel(int64_t n, int R, int Rc, int Cc, const U* __restrict__ rows,
                                                          const T* __restrict__ pl, const T* __restrict__ dinv, T omega,
                                                          T* __restrict__ ap, T* __restrict__ q, int* __restrict__ bad,
                                                          const T* __restrict__ base) {
  constexpr int RUN = (NT + 2) * 9;
  __shared__ T s_x[3][RUN];
  const int tid = threadIdx.x;
  for (int64_t i0 = (int64_t)blockIdx.x * NT; i0 < n; i0 += (int64_t)gridDim.x * NT) {
    // Every global load of the round is issued before the first one is waited for (the staged runs into registers, the nine
    // entries of the cell's row of A, its scaling): with two workgroups per CU the round used to be ~36 load latencies in a
    // row -- a loop of ten load-then-store-to-LDS steps per run, then one load per visited entry of A behind a branch.
    constexpr int NQ = (RUN + NT - 1) / NT;
    T v[3][NQ];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int64_t c0 = i0 + (int64_t)(r - 1) * R - 1;  // first cell of the run
#pragma unroll
      for (int qd = 0; qd < NQ; ++qd) {
        const int e = tid + qd * NT;
        const int64_t cell = c0 + e / 9;
        v[r][qd] = (e < RUN && cell >= 0 && cell < n) ? pl[c0 * 9 + e] : T(0);
      }
    }
    const int64_t i = i0 + tid;
    const bool on = i < n;
    T av[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      int64_t j;
      av[k] = on ? (T)dia_row_entry(rows, n, R, i, k, j) : T(0);
    }
    const T wq = on ? (dinv ? omega * dinv[i] : omega) : T(0);
    __syncthreads();  // (the previous round's reads)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int qd = 0; qd < NQ; ++qd) {
        const int e = tid + qd * NT;
        if (e < RUN) s_x[r][e] = v[r][qd];
      }
    __syncthreads();
    int ri = 0, ci = 0;
    if (on) cell_rc(i, R, ri, ci);
    const int I = lat_tile(ri, Rc), J = lat_tile(ci, Cc);
    T acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) acc[s] = T(0);
#pragma unroll
    for (int k = 0; k < 9 && on; ++k) {
      const T a = av[k];
      if (a == T(0)) continue;
      // (row / column of j = i + (k / 3 - 1) R + (k % 3 - 1) from the cell's own: no division -- the kernel used to spend
      // eighteen 64-bit divisions per cell here)
      int rj = ri + (k % 3 - 1), cj = ci + (k / 3 - 1);
      if (rj < 0) {
        rj += R;
        --cj;
      } else if (rj >= R) {
        rj -= R;
        ++cj;
      }
      const int Ij = lat_tile(rj, Rc), Jj = lat_tile(cj, Cc);
      const int sI = Ij - I, sJ = Jj - J;  // tile of j relative to tile of i: -1, 0, 1
      const T* xr = &s_x[k / 3][(tid + k % 3) * 9];  // row j of pl
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const T pv = xr[s];
        if (pv == T(0)) continue;
        const int dI = s % 3 - 1 + sI, dJ = s / 3 - 1 + sJ;
        if (dI < -1 || dI > 1 || dJ < -1 || dJ > 1) {
          atomicOr(bad, 2);
          continue;
        }
        const int so = (dJ + 1) * 3 + (dI + 1);
#pragma unroll
        for (int s2 = 0; s2 < 9; ++s2)
          if (s2 == so) acc[s2] += a * pv;
      }
    }
    // q = base - w (M pl) with w = omega * dinv (dinv null: w = omega) and base = pl unless given; both results leave
    // through LDS so that the stores are contiguous too
    T qv[9];
    if (on) {
      const T w = wq;
#pragma unroll
      for (int s = 0; s < 9; ++s) qv[s] = -w * acc[s] + (base ? base[i * 9 + s] : s_x[1][(tid + 1) * 9 + s]);
    }
    __syncthreads();  // every lane has taken its rows out of the staged runs
    if (on) {
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        s_x[0][tid * 9 + s] = acc[s];
        s_x[2][tid * 9 + s] = qv[s];
      }
    }
    __syncthreads();
    const int cnt = (int)(n - i0 < NT ? n - i0 : NT) * 9;
    for (int e = tid; e < cnt; e += NT) {
      if (ap) ap[i0 * 9 + e] = s_x[0][e];
      if (q) q[i0 * 9 + e] = s_x[2][e];
    }
  }
}

template <class U, class T>
inline void lattice_ap_q(int64_t n, int R, int Rc, int Cc, const U* rows, const T* pl, const T* dinv, T omega, T* ap, T* q,
                         int* bad, const T* base, hipStream_t st) {
  const int nt = knobs().apq_nt;
#define CS_APQ(NT_)                                                                                                      \
  hipLaunchKernelGGL((lattice_ap_q_kernel<U, T, NT_>), dim3((int)std::min<int64_t>(ceil_div(n, (int64_t)NT_), kMaxGrid)), \
                     dim3(NT_), 0, st, n, R, Rc, Cc, rows, pl, dinv, omega, ap, q, bad, base)
  if (nt == 256) CS_APQ(256);
  else if (nt == 64) CS_APQ(64);
  else CS_APQ(128);
#undef CS_APQ
}

// ---- Galerkin operator A_c = P^T (A P) ---------------------------------------------------------------------------------
// One thread per coarse node a = (Ia, Ja): walks the cells of the 3 x 3 block of tiles around its tile in cell order
// (deterministic), adds P[i, a] * (A P)[i, b] into a 5 x 5 window of coarse columns b around a (two tiles reach: see the
// header), and writes the non-zero entries (the diagonal always) in ascending column order into a padded row of 25
// slots; lattice_galerkin_compact_kernel packs the rows into CSR.
// (Round 6, measured and taken out again: the column's <= 10 values of P loaded up front into registers, the way the staging
// loads of lattice_ap_q_kernel now are -- 18.9 -> 34.2 ms in fp64 (the twelve-fold unrolled body), 4 ms better in fp32.)
template <class T>
__global__ __launch_bounds__(128) void lattice_galerkin_kernel(int R, int C, int Rc, int Cc, const T* __restrict__ pl,
                                                               const T* __restrict__ ap, int* __restrict__ count,
                                                               int* __restrict__ pcol, T* __restrict__ pval) {
  const int nc = Rc * Cc;
  for (int a = blockIdx.x * 128 + threadIdx.x; a < nc; a += gridDim.x * 128) {
    const int Ia = a % Rc, Ja = a / Rc;
    T acc[25];
#pragma unroll
    for (int s = 0; s < 25; ++s) acc[s] = T(0);
    for (int tJ = max(Ja - 1, 0); tJ <= min(Ja + 1, Cc - 1); ++tJ) {
      int c0, c1;
      tile_extent(tJ, Cc, C, c0, c1);
      for (int jc = c0; jc < c1; ++jc)
        for (int tI = max(Ia - 1, 0); tI <= min(Ia + 1, Rc - 1); ++tI) {
          int r0, r1;
          tile_extent(tI, Rc, R, r0, r1);
          const int sa = (Ja - tJ + 1) * 3 + (Ia - tI + 1);  // slot of a in the block around tile (tI, tJ)
          for (int ir = r0; ir < r1; ++ir) {
            const int64_t i = (int64_t)jc * R + ir;
            const T pv = pl[i * 9 + sa];
            if (pv == T(0)) continue;
#pragma unroll
            for (int s = 0; s < 9; ++s) {
              const T v = ap[i * 9 + s];
              // coarse column b = tile (tI + s%3 - 1, tJ + s/3 - 1); window index relative to a
              const int wI = tI + s % 3 - 1 - Ia + 2, wJ = tJ + s / 3 - 1 - Ja + 2;  // 0..4
              const int wi = wJ * 5 + wI;
#pragma unroll
              for (int s2 = 0; s2 < 25; ++s2)
                if (s2 == wi) acc[s2] += pv * v;
            }
          }
        }
    }
    int m = 0;
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      const int bI = Ia + s % 5 - 2, bJ = Ja + s / 5 - 2;
      if (bI < 0 || bI >= Rc || bJ < 0 || bJ >= Cc) continue;
      if (acc[s] == T(0) && s != 12) continue;
      pcol[(size_t)a * 25 + m] = bJ * Rc + bI;
      pval[(size_t)a * 25 + m] = acc[s];
      ++m;
    }
    count[a] = m;
  }
}

// The same sums with the fine rows staged in LDS. A workgroup owns TI consecutive coarse rows of ONE coarse column Ja and
// walks the fine raster columns of the tile columns Ja-1 .. Ja+1 in ascending order; per fine column the rows of `pl` and
// `ap` its coarse rows reach (tiles Ia0-1 .. Ia0+TI) are two contiguous runs, loaded with coalesced accesses. Read straight
// from memory (kernel above) every lane fetches one value of a 72-byte row per load at a lane stride of 216 bytes: 81 such
// loads per coarse node, each touching 64 cache lines (19.4 ms at 10000^2 fp64, 0.1 of the HBM roofline for its 14.4 GB).
// Summation order per coarse node unchanged (fine columns ascending, tiles and rows ascending, slots ascending): same bits.
// MEASURED SLOWER (round 6, 10000^2 fp64: 36.7 ms against 19.4; one wave per workgroup and two barriers per fine column leave
// five waves per CU to hide the staging loads) -- kept behind the debug switch CSGPU_GALERKIN_STAGED for the next attempt.
// Work items are dealt so that the workgroups of one XCD (blockIdx % 8) walk neighbouring coarse columns of one strip --
// a fine column is staged by three of them and the second and third find it in that XCD's L2.
template <class T, int TI>
__global__ __launch_bounds__(TI) void lattice_galerkin_staged_kernel(int R, int C, int Rc, int Cc, const T* __restrict__ pl,
                                                                      const T* __restrict__ ap, int* __restrict__ count,
                                                                      int* __restrict__ pcol, T* __restrict__ pval) {
  constexpr int ROWS = 3 * (TI + 2) + 2;  // (the raster's last tile holds up to 4 rows)
  __shared__ T s_p[ROWS * 9];
  __shared__ T s_a[ROWS * 9];
  const int tid = threadIdx.x;
  const int nstrips = (Rc + TI - 1) / TI;
  const int64_t items = (int64_t)nstrips * Cc;
  const int64_t chunk = (items + 7) / 8;
  for (int64_t w = blockIdx.x; w < chunk * 8; w += gridDim.x) {
    const int64_t item = (w % 8) * chunk + w / 8;
    if (item >= items) continue;  // (uniform over the workgroup)
    const int strip = (int)(item / Cc), Ja = (int)(item % Cc);
    const int Ia0 = strip * TI;
    const int Ia = Ia0 + tid;
    const bool on = Ia < Rc;
    int lo, hi, dummy;
    tile_extent(max(Ia0 - 1, 0), Rc, R, lo, dummy);
    tile_extent(min(Ia0 + TI, Rc - 1), Rc, R, dummy, hi);
    const int len = (hi - lo) * 9;
    T acc[25];
#pragma unroll
    for (int s = 0; s < 25; ++s) acc[s] = T(0);
    for (int tJ = max(Ja - 1, 0); tJ <= min(Ja + 1, Cc - 1); ++tJ) {
      int c0, c1;
      tile_extent(tJ, Cc, C, c0, c1);
      for (int jc = c0; jc < c1; ++jc) {
        __syncthreads();  // (the previous column's reads)
        const size_t base = ((size_t)jc * R + lo) * 9;
        for (int e = tid; e < len; e += TI) {
          s_p[e] = pl[base + e];
          s_a[e] = ap[base + e];
        }
        __syncthreads();
        if (!on) continue;
        for (int tI = max(Ia - 1, 0); tI <= min(Ia + 1, Rc - 1); ++tI) {
          int r0, r1;
          tile_extent(tI, Rc, R, r0, r1);
          const int sa = (Ja - tJ + 1) * 3 + (Ia - tI + 1);  // slot of a in the block around tile (tI, tJ)
          for (int ir = r0; ir < r1; ++ir) {
            const T pv = s_p[(ir - lo) * 9 + sa];
            if (pv == T(0)) continue;
#pragma unroll
            for (int s = 0; s < 9; ++s) {
              const T v = s_a[(ir - lo) * 9 + s];
              const int wI = tI + s % 3 - 1 - Ia + 2, wJ = tJ + s / 3 - 1 - Ja + 2;  // 0..4
              const int wi = wJ * 5 + wI;
#pragma unroll
              for (int s2 = 0; s2 < 25; ++s2)
                if (s2 == wi) acc[s2] += pv * v;
            }
          }
        }
      }
    }
    if (!on) continue;
    const int a = Ja * Rc + Ia;
    int m = 0;
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      const int bI = Ia + s % 5 - 2, bJ = Ja + s / 5 - 2;
      if (bI < 0 || bI >= Rc || bJ < 0 || bJ >= Cc) continue;
      if (acc[s] == T(0) && s != 12) continue;
      pcol[(size_t)a * 25 + m] = bJ * Rc + bI;
      pval[(size_t)a * 25 + m] = acc[s];
      ++m;
    }
    count[a] = m;
  }
}

template <class T>
__global__ __launch_bounds__(256) void lattice_galerkin_compact_kernel(int nc, const int* __restrict__ rp,
                                                                       const int* __restrict__ pcol, const T* __restrict__ pval,
                                                                       int* __restrict__ ci, T* __restrict__ va) {
  for (int a = blockIdx.x * 256 + threadIdx.x; a < nc; a += gridDim.x * 256) {
    const int b = rp[a], m = rp[a + 1] - b;
    for (int e = 0; e < m; ++e) {
      ci[b + e] = pcol[(size_t)a * 25 + e];
      va[b + e] = pval[(size_t)a * 25 + e];
    }
  }
}

// ---- the lattice form back to CSR (on demand: current maps, explicit residual checks, test hooks) ---------------------
template <class T, bool FILL>
__global__ __launch_bounds__(256) void dia_to_csr_kernel(int64_t n, int R, const T* __restrict__ rows, int* __restrict__ rp,
                                                         int* __restrict__ ci, T* __restrict__ va) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int m = 0;
    const int o = FILL ? rp[i] : 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      int64_t j;
      const T a = dia_row_entry(rows, n, R, i, k, j);
      if (a == T(0) && k != 4) continue;
      if (FILL) {
        ci[o + m] = (int)j;
        va[o + m] = a;
      }
      ++m;
    }
    if (!FILL) rp[i] = m;
  }
}

template <class T>
inline void dia_to_csr(const Dia<T>& D, Csr<T>& A, hipStream_t st) {
  const int64_t n = D.n;
  A.nrows = A.ncols = (int)n;
  A.rowptr.alloc((size_t)(n + 1) * sizeof(int));
  CS_HIP(hipMemsetAsync(A.rp(), 0, (size_t)(n + 1) * sizeof(int), st));
  hipLaunchKernelGGL((dia_to_csr_kernel<T, false>), dim3(grid_for(n)), dim3(256), 0, st, n, D.R, D.data(), A.rp(), (int*)nullptr,
                     (T*)nullptr);
  // the total may exceed int32 on rasters above 238 M cells: check in 64 bits before the scan
  DBuf total = dalloc<int>(1);
  exclusive_scan_i32(A.rp(), n + 1, st, dptr<int>(total));
  const int nnz = read_int(dptr<int>(total), st);
  CS_REQUIRE(nnz >= 0, CSGPU_BAD_ARGS, "raster too large for the CSR form this call needs (2^31 stored entries)");
  A.nnz = nnz;
  A.col.alloc((size_t)std::max(nnz, 1) * sizeof(int));
  A.val.alloc((size_t)std::max(nnz, 1) * sizeof(T));
  hipLaunchKernelGGL((dia_to_csr_kernel<T, true>), dim3(grid_for(n)), dim3(256), 0, st, n, D.R, D.data(), A.rp(), A.ci(), A.va());
  check_launch("lattice form -> CSR");
}

// Level 0 of H from the lattice form A0 (storage precision U; the hierarchy computes in T on the rounded values) of an
// R x C raster; weights `size0` (device, long long, may be null = every cell is a node; modified by the piece analysis).
// Leaves H.levels = {level 0 (index-free: Ql, Sdia, dinv), level 1 (A = Galerkin operator in CSR)} and the carry for the
// level loop. Returns false (H untouched) when the lattice pipeline does not apply.
template <class U, class T>
inline bool lattice_level0_setup(Hierarchy<T>& H, const Dia<U>& A0, int R, int C, long long* size0, const SetupParams& sp,
                                 SetupCarry& carry, hipStream_t st) {
  const int64_t n = A0.n;
  const int Rc = (R + 1) / 3, Cc = (C + 1) / 3;
  if (!sp.two_product || sp.theta != 0.0 || sp.aggregation == CSGPU_AGG_MIS2 || R < 6 || C < 6 || n != (int64_t)R * C ||
      n <= sp.max_coarse || sp.max_levels < 2)
    return false;
  const int64_t nc = (int64_t)Rc * Cc;
  const int g = grid_for(n);
  Level<T> L;
  L.A.nrows = L.A.ncols = (int)n;  // (no CSR arrays: the level is index-free)
  L.n = (int)n;
  // row statistics
  DBuf labs((size_t)n * sizeof(T)), part = dalloc<double>(2 * (size_t)g);
  L.dinv.alloc((size_t)n * sizeof(T));
  hipLaunchKernelGGL((dia_stats_kernel<U, T>), dim3(g), dim3(256), 0, st, n, R, A0.data(), dptr<T>(labs), dptr<double>(part),
                     dptr<double>(part) + g);
  std::vector<double> hp(2 * (size_t)g);
  CS_HIP(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  double rho = 0, dmax = 0;
  for (int b = 0; b < g; ++b) {
    rho = std::max(rho, hp[b]);
    dmax = std::max(dmax, hp[(size_t)g + b]);
  }
  hipLaunchKernelGGL((dia_dinv_kernel<U, T>), dim3(g), dim3(256), 0, st, n, A0.data(), (const T*)dptr<T>(labs),
                     64.0 * (double)std::numeric_limits<T>::epsilon() * dmax, dptr<T>(L.dinv));
  if (!(rho > 0)) rho = 1.0;
  L.rho = rho;
  L.omega = sp.omega_s / rho;
  // aggregates
  DBuf agg = dalloc<int>((size_t)n);
  carry.crow.alloc((size_t)nc * sizeof(int));
  carry.ccol.alloc((size_t)nc * sizeof(int));
  hipLaunchKernelGGL(lattice_tile_kernel, dim3(g), dim3(256), 0, st, n, R, Rc, Cc, dptr<int>(agg), dptr<int>(carry.crow),
                     dptr<int>(carry.ccol));
  const bool no_pieces = !knobs().tile_pieces;  // A/B knob
  // strength filter of the piece analysis (TileStrength in amg_setup.h): decided here for the whole hierarchy
  DBuf ones;  // unit weights of an all-valid raster that turns out heterogeneous
  double th2 = 0.0;
  carry.tile_theta = 0.0;
  if ((size0 || sp.tile_theta > 0.0) && !no_pieces) {
    DBuf piece((size_t)n), mainlab((size_t)nc);
    const int gt = grid_for(nc);
    const bool unit = size0 == nullptr;
    if (unit) {
      ones.alloc((size_t)n * sizeof(long long));
      hipLaunchKernelGGL(fill_ll_kernel, dim3(g), dim3(256), 0, st, dptr<long long>(ones), n, 1LL);
      size0 = dptr<long long>(ones);
    }
    const int r32 = sizeof(T) < sizeof(U) ? 1 : 0;
    auto pass1 = [&](double t2, int stride) {
      hipLaunchKernelGGL((dia_pieces_kernel<U, 1>), dim3(grid_for(ceil_div(nc, stride))), dim3(256), 0, st, R, C, Rc, Cc,
                         A0.data(), size0, (signed char*)piece.p, (signed char*)mainlab.p, dptr<int>(agg), t2, r32, stride);
    };
    const double t2 = sp.tile_theta * sp.tile_theta;
    const int stride = tile_sample_stride(nc);  // the test looks at a sample of the tiles (amg_setup.h, aggregate)
    int64_t valid = 0, out0 = 0, out1 = 0;
    if (t2 > 0.0) {
      if (!unit) {
        pass1(0.0, 1);
        piece_counts(n, R, Rc, Cc, piece, mainlab, valid, out0, st, stride);
      }
      pass1(t2, stride);
      piece_counts(n, R, Rc, Cc, piece, mainlab, valid, out1, st, stride);
    }
    const bool hetero = t2 > 0.0 && (double)(out1 - out0) > sp.tile_split_min * (double)std::max<int64_t>(valid, 1);
    if (t2 > 0.0) H.hetero_frac = (double)(out1 - out0) / (double)std::max<int64_t>(valid, 1);
    if (knobs().verbose)
      fprintf(stderr, "csgpu: tile strength test: %lld of %lld cells leave their tile at theta %.3g (%lld without): %s\n",
              (long long)out1, (long long)valid, sp.tile_theta, (long long)out0, hetero ? "filter ON" : "filter off");
    if (hetero) {
      th2 = t2;
      carry.tile_theta = sp.tile_theta;
      carry.weighted = true;  // (no effect when the caller's weights exist anyway)
      pass1(t2, 1);
    } else if (unit) {
      size0 = nullptr;  // regular tiles, every cell a node: nothing to analyse
      ones.release();
    } else if (t2 > 0.0) {
      pass1(0.0, stride);  // (the sampled tiles hold the labels of the filtered pass)
    } else {
      pass1(0.0, 1);
    }
    if (size0) {
      hipLaunchKernelGGL((dia_pieces_kernel<U, 2>), dim3(gt), dim3(256), 0, st, R, C, Rc, Cc, A0.data(), size0,
                         (signed char*)piece.p, (signed char*)mainlab.p, dptr<int>(agg), th2, r32, 1);
      for (int round = 1; round <= kOrphanRounds; ++round)
        hipLaunchKernelGGL((dia_orphans_kernel<U>), dim3(g), dim3(256), 0, st, R, C, Rc, Cc, A0.data(), size0,
                           (signed char*)piece.p, (const signed char*)mainlab.p, dptr<int>(agg), round,
                           round == kOrphanRounds ? 1 : 0, th2, r32);
      check_launch("tile pieces (lattice)");
      CS_HIP(hipStreamSynchronize(st));
    }
  }
  DBuf size_c = dalloc<unsigned long long>((size_t)nc);
  CS_HIP(hipMemsetAsync(size_c.p, 0, size_c.bytes, st));
  hipLaunchKernelGGL(lattice_sizes_kernel, dim3(g), dim3(256), 0, st, n, (const int*)dptr<int>(agg), (const long long*)size0,
                     dptr<unsigned long long>(size_c));
  if (size0 && knobs().verbose) {
    std::vector<unsigned long long> hs((size_t)nc);
    CS_HIP(hipMemcpy(hs.data(), size_c.p, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tot = 0, empty = 0;
    for (unsigned long long v : hs) {
      tot += v;
      empty += v == 0 ? 1 : 0;
    }
    fprintf(stderr, "csgpu: cell space: %lld nodes, %lld of them without an aggregate (weight 0), %llu of %lld tiles empty\n",
            (long long)sp.n_real, (long long)sp.n_real - (long long)tot, empty, (long long)nc);
  }
  DBuf tv((size_t)n * sizeof(T));
  hipLaunchKernelGGL((lattice_tentative_kernel<T>), dim3(g), dim3(256), 0, st, n, (const int*)dptr<int>(agg),
                     (const long long*)size0, (const unsigned long long*)dptr<unsigned long long>(size_c), dptr<T>(tv));
  // P, A P, Q
  DBuf pl((size_t)n * 9 * sizeof(T)), apl((size_t)n * 9 * sizeof(T)), ql((size_t)n * 9 * sizeof(T)), bad = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  hipLaunchKernelGGL((lattice_p_kernel<U, T>), dim3(g), dim3(256), 0, st, n, R, Rc, Cc, A0.data(), (const int*)dptr<int>(agg),
                     (const T*)dptr<T>(tv), (const T*)dptr<T>(labs), sp.omega_p, dptr<T>(pl), dptr<int>(bad));
  lattice_ap_q<U, T>(n, R, Rc, Cc, A0.data(), (const T*)dptr<T>(pl), (const T*)dptr<T>(L.dinv), (T)L.omega, dptr<T>(apl),
                     dptr<T>(ql), dptr<int>(bad), (const T*)nullptr, st);
  check_launch("lattice P / A P / Q");
  if (read_int(dptr<int>(bad), st) != 0) return false;  // (cannot happen for tile aggregates; the CSR pipeline takes over)
  tv.release();
  labs.release();
  // refined tiles (NODATA cells, strength-aware tiles): a second coarse function on the badly shaped aggregates (enrich.h)
  H.enr = Enrich();
  if (size0 && n < 0x7fffffffLL) {
    // (optional: its n-sized temporaries come on top of pl / apl / ql -- a raster that fitted without it must not fail with
    // it)
    try {
      enrich_setup<U, T>(H.enr, A0.data(), R, C, Rc, Cc, (const long long*)size0, (const int*)dptr<int>(agg),
                         (const unsigned long long*)dptr<unsigned long long>(size_c), st);
      if (knobs().enrich_fused) enrich_coarse_setup<T>(H.enr, (const T*)dptr<T>(ql), R, Rc, Cc, st);
    } catch (const Error& e) {
      if (e.code != CSGPU_OOM) throw;
      (void)hipGetLastError();
      H.enr = Enrich();
      if (knobs().verbose) fprintf(stderr, "csgpu: enrichment skipped (out of device memory during its set-up)\n");
    }
  }
  agg.release();
  // Galerkin operator of level 1
  Csr<T> Ac;
  Ac.nrows = Ac.ncols = (int)nc;
  Ac.rowptr.alloc((size_t)(nc + 1) * sizeof(int));
  CS_HIP(hipMemsetAsync(Ac.rp(), 0, (size_t)(nc + 1) * sizeof(int), st));
  {
    DBuf pcol = dalloc<int>((size_t)nc * 25), pval((size_t)nc * 25 * sizeof(T)), total = dalloc<int>(1);
    int gg = ceil_div(nc, 128);
    if (gg > 65536) gg = 65536;
    if (knobs().galerkin_staged) {
      constexpr int TI = 64;
      const int64_t items = (int64_t)ceil_div(Rc, TI) * Cc;
      hipLaunchKernelGGL((lattice_galerkin_staged_kernel<T, TI>), dim3((int)std::min<int64_t>(((items + 7) / 8) * 8, 16384)),
                         dim3(TI), 0, st, R, C, Rc, Cc, (const T*)dptr<T>(pl), (const T*)dptr<T>(apl), Ac.rp(), dptr<int>(pcol),
                         dptr<T>(pval));
    } else {
      hipLaunchKernelGGL((lattice_galerkin_kernel<T>), dim3(gg), dim3(128), 0, st, R, C, Rc, Cc, (const T*)dptr<T>(pl),
                         (const T*)dptr<T>(apl), Ac.rp(), dptr<int>(pcol), dptr<T>(pval));
    }
    exclusive_scan_i32(Ac.rp(), nc + 1, st, dptr<int>(total));
    Ac.nnz = read_int(dptr<int>(total), st);
    Ac.col.alloc((size_t)std::max<int64_t>(Ac.nnz, 1) * sizeof(int));
    Ac.val.alloc((size_t)std::max<int64_t>(Ac.nnz, 1) * sizeof(T));
    hipLaunchKernelGGL((lattice_galerkin_compact_kernel<T>), dim3(grid_for(nc)), dim3(256), 0, st, (int)nc, (const int*)Ac.rp(),
                       (const int*)dptr<int>(pcol), (const T*)dptr<T>(pval), Ac.ci(), Ac.va());
    check_launch("lattice Galerkin");
    CS_HIP(hipStreamSynchronize(st));
  }
  pl.release();
  apl.release();
  // the two-product level: index-free Q and S in lattice form
  L.Ql.n = n;
  L.Ql.R = R;
  L.Ql.C = C;
  L.Ql.Rc = Rc;
  L.Ql.Cc = Cc;
  L.Ql.q = std::move(ql);
  dia_build_s(A0, (const T*)dptr<T>(L.dinv), L.omega, L.Sdia, st);
  H.levels.clear();
  H.levels.push_back(std::move(L));
  H.levels.emplace_back();
  H.levels.back().A = std::move(Ac);
  carry.size = std::move(size_c);
  carry.gridR = Rc;
  carry.gridC = Cc;
  return true;
}

// ---- level 1 of a raster hierarchy in lattice form ---------------------------------------------------------------------
// Level 1 of a full raster is again a nine-point lattice (R1 x C1 = the tiles of level 0) with regular 3 x 3 aggregates,
// and it is where a V-cycle spends most of its time below the fine level (10000^2, K = 16, fp64: 6.1 of 34 ms -- seven
// CSR products moving 15 vector passes). With w0, w1 the weights of its two Jacobi sweeps (Chebyshev or damped),
//     S  = (w0 + w1) D^-1 - w0 w1 D^-1 A D^-1       two sweeps from a zero guess are x = S b; two sweeps from x are
//                                                    x + S (b - A x)                      (nine-point, symmetric)
//     Q2 = (I - S A) P                               restriction of the residual after pre-smoothing: R (b - A S b) = Q2' b
// the level collapses to FOUR marching products without a column index (vcycle, pcg.h):
//     x = S b ;  b_c = Q2' b ;  [levels below] ;  t = b - A x ;  out = x + S t + Q2 x_c
// (out = [x + P x_c] + S (b - A [x + P x_c]), i.e. prolongation + the two post-sweeps). Q2 reaches three cells beyond a
// row's own tile -- still inside the 3 x 3 block of tiles, so the index-free form of level 0's Q (LatticeQ) holds it.
// Built from the CSR operators of the level (kept: test hooks, and the generic branch when the sweep count differs).
// Declined (level untouched) when A is not a nine-point lattice of period R or P leaves the tile block -- cell-space
// hierarchies whose tiles were refined by the piece analysis.
template <class T>
inline void lattice_level1_setup(Level<T>& L, const int* agg, int R, int C, int nagg, hipStream_t st) {
  const bool off = !knobs().lattice_l1;  // A/B knob
  const int min_rows = knobs().lattice_l1_min_rows;
  const int64_t n = (int64_t)R * C;
  const int Rc = (R + 1) / 3, Cc = (C + 1) / 3;
  if (off || n < min_rows || L.A.nrows != n || (int64_t)Rc * Cc != nagg || R < 6 || C < 6) return;
  Dia<T> Ad;
  // (the Galerkin operator is symmetric up to rounding, not bit for bit: the lattice form keeps the upper triangle)
  if (!dia_from_csr(L.A, R, Ad, st, true)) return;
  LatticeQ<T> Pl;
  if (!lattice_q_from_csr(L.P, agg, R, C, Pl, st)) return;
  const double w0 = L.weights.empty() ? L.omega : L.weights[0], w1 = L.weights.empty() ? L.omega : L.weights[1];
  Dia<T> Sd;
  dia_build_s(Ad, (const T*)dptr<T>(L.dinv), w0, Sd, st, w1);
  DBuf apl((size_t)n * 9 * sizeof(T)), ql((size_t)n * 9 * sizeof(T)), bad = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  const int g = grid_for(n);
  lattice_ap_q<T, T>(n, R, Rc, Cc, Ad.data(), Pl.data(), (const T*)nullptr, T(0), dptr<T>(apl), (T*)nullptr, dptr<int>(bad),
                     (const T*)nullptr, st);                                                                          // A P
  lattice_ap_q<T, T>(n, R, Rc, Cc, Sd.data(), (const T*)dptr<T>(apl), (const T*)nullptr, T(1), (T*)nullptr, dptr<T>(ql),
                     dptr<int>(bad), Pl.data(), st);                                                                  // P - S (A P)
  check_launch("lattice level 1");
  if (read_int(dptr<int>(bad), st) != 0) return;
  L.Adia = std::move(Ad);
  L.Sdia = std::move(Sd);
  L.Ql.n = n;
  L.Ql.R = R;
  L.Ql.C = C;
  L.Ql.Rc = Rc;
  L.Ql.Cc = Cc;
  L.Ql.q = std::move(ql);
  if (knobs().verbose) fprintf(stderr, "csgpu: level 1 (%d x %d) in lattice form\n", R, C);
}

}  // namespace csgpu
