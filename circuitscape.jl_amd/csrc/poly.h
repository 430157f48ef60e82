// poly.h -- rasters WITH short-circuit polygons on the index-free lattice path ("projected lattice", round 4).
//
// The reference merges every cell of a polygon into ONE node before it builds the graph (construct_node_map with a
// polymap, src/raster/pairwise.jl:276-301; construct_graph :316-362), which destroys the lattice structure of the matrix;
// csgpu_raster_setup_poly therefore used to build the merged CSR graph and run the CSR kernels on a MIS(2) / coordinate
// hierarchy -- 2.2-2.35x slower per batch than the same raster without polygons (profiles/r4_polygons_5000_csr_path.jsonl).
//
// A merged polygon is an EQUIPOTENTIAL. With E the cell -> node incidence (one 1 per row) the merged Laplacian is
// E'AE, and y solves (E'AE) y = b_m exactly when x = E y solves the cell-space system restricted to the subspace
// S = range(E) = { vectors that are constant on every polygon }:
//     Pi A Pi x = Pi b_c,   Pi = E (E'E)^-1 E'  (the average over each polygon's cells),   E' b_c = b_m.
// So the solve stays on the R x C lattice, every cell keeps its row, and PCG runs in S:
//   * r <- Pi r after every residual update, z <- Pi z after every V-cycle (p = z + beta p and x stay in S by
//     induction); the CG scalars need no change: p'(A p) = p'(Pi A p) and r'z = r'(Pi z) for p, r in S;
//   * edges BETWEEN two cells of one polygon carry no current for any vector of S, so their weights do not enter
//     Pi A Pi at all -- they are free parameters of the PRECONDITIONER. Multiplied by 100 (CSGPU_POLY_STRENGTH) they make
//     the hierarchy see each polygon as the nearly rigid body it is: the strength-aware tiles of round 3 (always on for
//     these handles) cut the aggregates along the polygon boundaries. Measured (emulator, 180^2 raster, 12 polygons of
//     up to 22 cells across): 23.6 iterations with the plain projection, 12.0 with the strengthened interiors, 9 for the
//     raster without polygons, 10.9 on the merged CSR graph (whose iterations cost 2.2x as much); with the EXACT inverse of
//     the unstrengthened lattice matrix as preconditioner the projected system still needs 19 iterations, with that of
//     the strengthened one 3 -- the gap to the polygon-free count is the hierarchy's, not the projection's;
//   * a NODATA cell inside a polygon belongs to the polygon's node in the reference (nodemap != 0 there) and its edges
//     to valid neighbours exist with the weight construct_graph computes from its stored value (cond_avg(g, 0) = g / 2):
//     such cells keep a row here too.
// The projection is two small kernels over the polygons' member cells (fixed summation order: bit-reproducible).
#pragma once
#include "blas1.h"
#include "raster.h"

namespace csgpu {

static const int kPolyChunk = 1024;  // member cells per workgroup of the projection kernels

// Member lists of the polygons (device): cells[ptr[p] .. ptr[p+1]) ascending; chunks of at most kPolyChunk cells of ONE
// polygon: chunk_first[q] = first member index of chunk q, chunk_poly[q] its polygon; poly_chunk0[p] = first chunk of p.
struct PolyProj {
  int npoly = 0, nchunks = 0;
  const int* ptr = nullptr;
  const int* cells = nullptr;
  const int* chunk_first = nullptr;
  const int* chunk_poly = nullptr;
  const int* poly_chunk0 = nullptr;  // [npoly + 1]
  double* chunk_sum = nullptr;       // [nchunks][kMaxK]
};

// pass 1: chunk_sum[q][c] = sum over the chunk's cells of v[cell][c], cells ascending
template <class V, int K>
__global__ __launch_bounds__(256) void poly_chunk_sum_kernel(PolyProj pp, const V* __restrict__ v, const int* skip) {
  if (skip && *skip) return;
  __shared__ double s_part[256];
  const int q = blockIdx.x;
  const int p = pp.chunk_poly[q];
  const int lo = pp.chunk_first[q], hi = min(pp.ptr[p + 1], lo + kPolyChunk);
  constexpr int G = 256 / K;  // threads per column
  const int c = threadIdx.x % K, g = threadIdx.x / K;
  double s = 0.0;
  if (g < G)
    for (int m = lo + g; m < hi; m += G) s += (double)v[(size_t)pp.cells[m] * K + c];
  s_part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0.0;
    for (int gg = 0; gg < G; ++gg) t += s_part[gg * K + threadIdx.x];
    pp.chunk_sum[(size_t)q * kMaxK + threadIdx.x] = t;
  }
}

// pass 2: v[cell][c] = (sum over the polygon's chunks, in order) / |polygon| for the chunk's cells; optional copy in a
// second precision (the preconditioner-precision copy of r)
template <class V, class V2, int K>
__global__ __launch_bounds__(256) void poly_apply_kernel(PolyProj pp, V* __restrict__ v, V2* __restrict__ v2, const int* skip) {
  if (skip && *skip) return;
  __shared__ double s_mean[K];
  const int q = blockIdx.x;
  const int p = pp.chunk_poly[q];
  const int lo = pp.chunk_first[q], hi = min(pp.ptr[p + 1], lo + kPolyChunk);
  if (threadIdx.x < K) {
    double t = 0.0;
    for (int qq = pp.poly_chunk0[p]; qq < pp.poly_chunk0[p + 1]; ++qq) t += pp.chunk_sum[(size_t)qq * kMaxK + threadIdx.x];
    s_mean[threadIdx.x] = t / (double)(pp.ptr[p + 1] - pp.ptr[p]);
  }
  __syncthreads();
  constexpr int G = 256 / K;
  const int c = threadIdx.x % K, g = threadIdx.x / K;
  if (g < G)
    for (int m = lo + g; m < hi; m += G) {
      const size_t at = (size_t)pp.cells[m] * K + c;
      v[at] = (V)s_mean[c];
      if (v2) v2[at] = (V2)s_mean[c];
    }
}

// v <- Pi v (and v2 <- the same values, may be null)
template <class V, class V2, int K>
inline void poly_project(const PolyProj& pp, V* v, V2* v2, const int* skip, hipStream_t st) {
  if (pp.nchunks <= 0) return;
  hipLaunchKernelGGL((poly_chunk_sum_kernel<V, K>), dim3(pp.nchunks), dim3(256), 0, st, pp, (const V*)v, skip);
  hipLaunchKernelGGL((poly_apply_kernel<V, V2, K>), dim3(pp.nchunks), dim3(256), 0, st, pp, v, v2, skip);
}

// Residual norms in NODE space (the reference's check is ||A x - b|| / ||b|| of the MERGED system,
// src/core.jl:640-641). A polygon of s cells whose projected residual is the constant rho per cell carries the node residual
// s rho in the merged system: ||r_m||^2 = sum_ordinary r^2 + sum_polygons (s rho)^2, while the cell-space norm of Pi r holds
// s rho^2 for it. The difference, sum_p (s^2 - s) rho_p^2, is one extra row of r'r partials:
//   row[c] = sum_p s_p (s_p - 1) mean_p[c]^2,   mean_p = (chunk sums of p, in order) / s_p
// from the chunk sums the LAST projection left behind (poly_project, or poly_chunk_sum_kernel alone). One workgroup, fixed
// summation order (polygons strided over 256 / K thread groups, groups combined in order): bit-reproducible.
template <int K>
__global__ __launch_bounds__(256) void poly_norm_corr_kernel(PolyProj pp, double* __restrict__ row, const int* skip) {
  if (skip && *skip) return;
  __shared__ double s_part[256];
  constexpr int G = 256 / K;
  const int c = threadIdx.x % K, g = threadIdx.x / K;
  double acc = 0.0;
  if (g < G)
    for (int p = g; p < pp.npoly; p += G) {
      const double s = (double)(pp.ptr[p + 1] - pp.ptr[p]);
      if (!(s > 1.0)) continue;
      double t = 0.0;
      for (int q = pp.poly_chunk0[p]; q < pp.poly_chunk0[p + 1]; ++q) t += pp.chunk_sum[(size_t)q * kMaxK + c];
      const double mean = t / s;
      acc += s * (s - 1.0) * mean * mean;
    }
  s_part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0.0;
    for (int gg = 0; gg < G; ++gg) t += s_part[gg * K + threadIdx.x];
    row[threadIdx.x] = t;
  }
}

// ---- raster + polygon labels -> lattice form -------------------------------------------------------------------------
// label[k] (column-major cell id k): the polygon's representative cell for every cell of a merged polygon (NODATA cells
// included), k itself for an ordinary valid cell, -1 for a cell without a node (poly_label_kernel, raster.h).
// Every cell with label >= 0 keeps a row. Edge weights as construct_graph computes them from the STORED values of the two
// cells (raster_edge); an edge inside a polygon is multiplied by strength[polygon] (a zero weight there -- two NODATA
// cells -- is replaced by the conductance of the polygon's representative cell first), see the header comment.
// part / cnt: sum of squares / number of the entries the MERGED matrix stores, up to how a polygon's row is summed: the
// diagonal of a polygon cell counts its external edges only, internal edges do not count (regularisation, core.jl:161).
template <class T>
__global__ __launch_bounds__(256) void raster_dia_poly_kernel(int R, int C, int four, int avg_res, const T* __restrict__ cond,
                                                              const int* __restrict__ label,
                                                              const int* __restrict__ cell_poly,
                                                              const double* __restrict__ strength,
                                                              T* __restrict__ rows, double* __restrict__ part,
                                                              unsigned long long* __restrict__ cnt) {
  __shared__ double sm[4];
  __shared__ unsigned long long smc[4];
  const int64_t n = (int64_t)R * C;
  double ss = 0.0;
  unsigned long long c = 0;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const int lab = label[id];
    T out[5] = {T(0), T(0), T(0), T(0), T(0)};
    if (lab >= 0) {
      const double g0 = (double)cond[(size_t)i * C + j];
      double deg = 0.0, deg_ext = 0.0;
      for (int dj = -1; dj <= 1; ++dj) {
        const int jj = j + dj;
        if (jj < 0 || jj >= C) continue;
        for (int di = -1; di <= 1; ++di) {
          const int ii = i + di;
          if (ii < 0 || ii >= R || (di == 0 && dj == 0)) continue;
          const bool diag = (di != 0 && dj != 0);
          if (diag && four) continue;
          const int64_t kn = (int64_t)jj * R + ii;
          const int ln = label[kn];
          if (ln < 0) continue;
          const double g1 = (double)cond[(size_t)ii * C + jj];
          double w = raster_edge(g0, g1, diag, avg_res != 0);
          const bool interior = ln == lab;  // two cells of one polygon (an ordinary cell's label is its own id)
          if (interior) {
            if (!(w > 0.0)) {
              const int rk = lab;
              const double gr = (double)cond[(size_t)(rk % R) * C + rk / R];
              w = raster_edge(gr, gr, diag, avg_res != 0);
            }
            w *= strength[cell_poly[id]];  // (per polygon: the member count decides, csgpu.hip setup_poly_lattice)
          } else {
            deg_ext += w;
            const T v = (T)(-w);
            ss += (double)v * (double)v;
            ++c;
          }
          deg += w;
          const T v = (T)(-w);
          if (dj == 0 && di == 1) out[1] = v;
          else if (dj == 1 && di == -1) out[2] = v;
          else if (dj == 1 && di == 0) out[3] = v;
          else if (dj == 1 && di == 1) out[4] = v;
        }
      }
      out[0] = (T)deg;
      ss += deg_ext * deg_ext;
      ++c;
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

// regularisation shift + identity rows of the cells without a node + weights (raster_dia_finish_kernel with the row
// mask taken from the labels instead of the conductances)
template <class T>
__global__ __launch_bounds__(256) void raster_dia_poly_finish_kernel(int R, int C, const int* __restrict__ label,
                                                                     T* __restrict__ rows, const double* __restrict__ part,
                                                                     int nparts, double eps, long long* __restrict__ size0) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = block_sum_256(s, sm);
  const T shift = (T)(eps * sqrt(s));
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const bool valid = label[id] >= 0;
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

// cell-space maps of a polygon handle: cellmap (row-major, 1-based ROW id = cell id + 1, 0: no row), cell2node
// (column-major, 1-based node id, 0: none), cell_poly (column-major: dense polygon index of a merged cell, -1 otherwise)
__global__ __launch_bounds__(256) void poly_cell_maps_kernel(int R, int C, const int* __restrict__ label,
                                                             const int* __restrict__ node, const int* __restrict__ node_poly,
                                                             int* __restrict__ cellmap, int* __restrict__ cell2node,
                                                             int* __restrict__ cell_poly) {
  const int64_t n = (int64_t)R * C;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
    const int i = (int)(k % R), j = (int)(k / R);
    const bool row = label[k] >= 0;
    cellmap[(size_t)i * C + j] = row ? (int)k + 1 : 0;
    cell2node[k] = row ? node[k] + 1 : 0;
    cell_poly[k] = row ? node_poly[node[k]] : -1;
  }
}

// node2cell[node] = column-major id of the cell that carries the node (its raster coordinates)
__global__ __launch_bounds__(256) void poly_node2cell_kernel(int64_t nnode, int R, const int* __restrict__ nrow,
                                                             const int* __restrict__ ncol, int* __restrict__ node2cell) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nnode; k += (int64_t)gridDim.x * 256)
    node2cell[k] = ncol[k] * R + nrow[k];
}

}  // namespace csgpu
