// raster.h -- device-side construction of the raster graph Laplacian (scope row N4, used by bench.py and tests).
//
// GPU counterpart, for an ALL-VALID raster without polygons, of
//   construct_node_map   src/raster/pairwise.jl:271-301  (column-major numbering of cells with conductance > 0)
//   construct_graph      src/raster/pairwise.jl:316-362  (E, S, SE, NE neighbours; cond_avg / res_avg, diagonals / sqrt 2)
//   laplacian!           src/core.jl:608-634
// producing the CSR Laplacian directly in HBM (no COO, no host transient): one thread per cell writes its own
// sorted row. Also emits each node's (row, col) for the tile-seeded aggregation.
#pragma once
#include "prims.h"

namespace csgpu {

__device__ __forceinline__ double raster_edge(double x, double y, bool diag, bool avg_res) {
  double v = avg_res ? 1.0 / ((1.0 / x + 1.0 / y) * 0.5) : (x + y) * 0.5;
  return diag ? v / 1.4142135623730951 : v;
}

__global__ __launch_bounds__(256) void raster_count_kernel(int R, int C, int four, int* __restrict__ counts) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    const int up = i > 0, dn = i < R - 1, lf = j > 0, rt = j < C - 1;
    int cnt = 1 + up + dn + lf + rt;
    if (!four) cnt += (up & lf) + (dn & lf) + (up & rt) + (dn & rt);
    counts[id] = cnt;
  }
}

// cond: device array, row-major [R][C] (cell (i,j) at i*C + j), all entries > 0.
template <class T>
__global__ __launch_bounds__(256) void raster_fill_kernel(int R, int C, int four, int avg_res,
                                                          const T* __restrict__ cond, const int* __restrict__ rp,
                                                          int* __restrict__ ci, T* __restrict__ va,
                                                          int* __restrict__ nrow, int* __restrict__ ncol) {
  const int64_t n = (int64_t)R * C;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (int64_t)gridDim.x * 256) {
    const int i = (int)(id % R), j = (int)(id / R);
    nrow[id] = i;
    ncol[id] = j;
    const double g0 = (double)cond[(size_t)i * C + j];
    int k = rp[id];
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
        ci[k] = (int)((int64_t)jj * R + ii);
        if (self) {
          kdiag = k;
        } else {
          const double w = raster_edge(g0, (double)cond[(size_t)ii * C + jj], diag, avg_res != 0);
          va[k] = (T)(-w);
          deg += w;
        }
        ++k;
      }
    }
    va[kdiag] = (T)deg;
  }
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

}  // namespace csgpu
