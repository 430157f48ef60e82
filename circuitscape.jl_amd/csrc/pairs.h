// pairs.h -- K9: pair batching around the solver (right-hand sides, grounding shift, resistances, gathers).
//
// GPU counterpart of the per-pair body of solve(prob, ::AMGSolver, ...) in the reference:
//   current[src] = -1 ; current[dst] = +1                          src/core.jl:224-226
//   voltages .-= voltages[src] ; resistance = v[dst] - v[src]       src/core.jl:231-232
//   focal voltages consumed by the shortcut (update_voltmatrix!)    src/core.jl:685-703
// and of the batched layout of the direct-solver driver (n x batch right-hand sides, src/core.jl:455-472).
#pragma once
#include "prims.h"

namespace csgpu {

// b (interleaved n x K, pre-zeroed): column c gets -1 at src[c], +1 at dst[c]
template <class T, int K>
__global__ __launch_bounds__(64) void pairs_rhs_kernel(T* __restrict__ b, const int* __restrict__ src,
                                                       const int* __restrict__ dst, int ncols) {
  const int c = threadIdx.x;
  if (c < ncols && c < K && src[c] != dst[c]) {  // src == dst: zero right-hand side, R = 0 (the reference skips it, core.jl:210)
    b[(size_t)src[c] * K + c] = T(-1);
    b[(size_t)dst[c] * K + c] = T(1);
  }
}

// resist[c] = x[dst_c] - x[src_c]; gathered[c*ngather + g] = x[gather[g]] - x[src_c]
template <class T, int K>
__global__ __launch_bounds__(256) void pairs_extract_kernel(const T* __restrict__ x, const int* __restrict__ src,
                                                            const int* __restrict__ dst, int ncols,
                                                            const int* __restrict__ gather, int ngather,
                                                            T* __restrict__ resist, T* __restrict__ gathered) {
  const int64_t total = (int64_t)ncols * (ngather + 1);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t / (ngather + 1));
    const int g = (int)(t % (ngather + 1));
    const T vs = x[(size_t)src[c] * K + c];
    if (g == ngather)
      resist[c] = x[(size_t)dst[c] * K + c] - vs;
    else
      gathered[(size_t)c * ngather + g] = x[(size_t)gather[g] * K + c] - vs;
  }
}

// the same from the focal-node accumulation of a resistance-only solve: xf is [ngather + 2K][K] with the gathered
// nodes first, then the K source nodes, then the K destination nodes (PcgWork::set_focal)
template <class T, int K>
__global__ __launch_bounds__(256) void pairs_extract_focal_kernel(const T* __restrict__ xf, int ncols, int ngather,
                                                                  T* __restrict__ resist, T* __restrict__ gathered) {
  const int64_t total = (int64_t)ncols * (ngather + 1);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t / (ngather + 1));
    const int g = (int)(t % (ngather + 1));
    const T vs = xf[(size_t)(ngather + c) * K + c];
    if (g == ngather)
      resist[c] = xf[(size_t)(ngather + K + c) * K + c] - vs;
    else
      gathered[(size_t)c * ngather + g] = xf[(size_t)g * K + c] - vs;
  }
}

// out (column-major n x ncols) = x[:, c] - x[src_c, c]     (de-interleave + grounding shift)
template <class T, int K>
__global__ __launch_bounds__(256) void pairs_volt_kernel(int64_t n, const T* __restrict__ x,
                                                         const int* __restrict__ src, int ncols,
                                                         T* __restrict__ out) {
  const int64_t total = n * ncols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t / n);
    const int64_t i = t % n;
    out[t] = x[(size_t)i * K + c] - x[(size_t)src[c] * K + c];
  }
}

// interleave / de-interleave general right-hand sides (column-major n x ncols <-> [n][K])
template <class T, int K>
__global__ __launch_bounds__(256) void interleave_kernel(int64_t n, const T* __restrict__ colmajor, int ncols,
                                                         T* __restrict__ inter) {
  const int64_t total = n * K;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % K);
    const int64_t i = e / K;
    inter[e] = c < ncols ? colmajor[(size_t)c * n + i] : T(0);
  }
}
template <class T, int K>
__global__ __launch_bounds__(256) void deinterleave_kernel(int64_t n, const T* __restrict__ inter, int ncols,
                                                           T* __restrict__ colmajor) {
  const int64_t total = n * ncols;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int c = (int)(t / n);
    const int64_t i = t % n;
    colmajor[t] = inter[(size_t)i * K + c];
  }
}

// sparse right-hand sides (csgpu_solve_sources): b (interleaved n x K, pre-zeroed) gets val[e] at (row[e], col[e]); the host
// has merged duplicates, so every (row, col) occurs once -- plain stores, deterministic
template <class T, int K>
__global__ __launch_bounds__(256) void sparse_rhs_kernel(int cnt, const int* __restrict__ row, const int* __restrict__ col,
                                                         const T* __restrict__ val, T* __restrict__ b) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < cnt) b[(size_t)row[e] * K + col[e]] = val[e];
}

// out[c] = x[node[c]][c] for the columns of a batch (node < 0: 0) -- the voltage of a one-to-all column's check node
// (`res[i] = v[1]`, src/raster/onetoall.jl:141)
template <class T, int K>
__global__ __launch_bounds__(64) void gather_columns_kernel(const T* __restrict__ x, const int* __restrict__ node, int ncols,
                                                            T* __restrict__ out) {
  const int c = threadIdx.x;
  if (c < K) out[c] = (c < ncols && node[c] >= 0) ? x[(size_t)node[c] * K + c] : T(0);
}

}  // namespace csgpu
