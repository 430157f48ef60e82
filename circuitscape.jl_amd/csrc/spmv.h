// spmv.h -- K1: CSR SpMV / multi-RHS SpMM for gfx950 (CSR-stream: coalesced matrix streaming through LDS).
//
// GPU counterpart of `A*p` inside Krylov.cg (reference call site src/core.jl:639) and of every
// residual / restriction / prolongation product inside the AMG V-cycle (AlgebraicMultigrid.jl __solve!).
//
// Layout: vectors for a batch of K right-hand sides are interleaved, X[node*K + c]; the K values a gather
// needs are one contiguous K*sizeof(T) segment, and the matrix is streamed ONCE for all K columns.
//
// One workgroup (256 threads = 4 waves) owns kRows consecutive rows:
//   1. row pointers -> LDS;
//   2. the rows' nonzeros are streamed from HBM fully coalesced, tile by tile:
//        K == 1: lane k loads val[k], col[k], gathers x[col[k]] and parks the PRODUCT in LDS;
//        K  > 1: val[k] / col[k] are parked in LDS (the gather is done in phase 3 by K adjacent lanes);
//   3. K == 1: thread r sums row r's products from LDS (stride = row length: odd on rasters, no bank conflicts);
//      K  > 1: K adjacent lanes own one row (lane c = column c) and walk its nonzeros from LDS, gathering
//              the contiguous segment x[col*K .. col*K+K);
//   4. fused epilogue (residual, damped-Jacobi update, prolongation add) and optional fused dot-product
//      partials (wave shuffle -> LDS -> one partial per workgroup and column; deterministic, no atomics).
// Rows longer than the LDS tile simply span several tiles (each thread accumulates the part of its row
// inside the current tile), so any row-length distribution is handled.
//
// Algorithmic bytes per launch: nnz*(sizeof(T)+4) + (n+1)*4 + 2*n*K*sizeof(T).
#pragma once
#include "prims.h"

namespace csgpu {

enum SpmvEpi {
  EPI_PLAIN = 0,   // y = A x
  EPI_RESID = 1,   // y = b - A x
  EPI_JACOBI = 2,  // y = x + omega * dinv .* (b - A x)
  EPI_ADD = 3      // y = xadd + A x          (prolongation: xadd may alias y)
};

static const int kSpmvRows = 256;   // rows per workgroup pass
static const int kSpmvTile = 2560;  // nonzeros staged in LDS per tile

template <class T>
struct SpmvArgs {
  int nrows;
  const int* rowptr;
  const int* col;
  const T* val;
  const T* x;       // input, interleaved [ncols][K]
  T* y;             // output, interleaved [nrows][K]
  const T* b;       // EPI_RESID / EPI_JACOBI
  const T* xadd;    // EPI_ADD (may alias y); EPI_JACOBI reads x itself for the row's own value
  const T* dinv;    // EPI_JACOBI: 1 / a_ii
  T omega;          // EPI_JACOBI
  const T* dotw;    // DOT: partial[c] += dotw[row*K+c] * y[row*K+c]
  double* partials; // DOT: [gridDim.x][K]
};

template <class T, int K, int EPI, bool DOT>
__global__ __launch_bounds__(256) void spmv_kernel(SpmvArgs<T> a) {
  __shared__ int s_rp[kSpmvRows + 1];
  __shared__ T s_val[kSpmvTile];
  __shared__ int s_col[K > 1 ? kSpmvTile : 1];
  __shared__ double s_red[4 * (K > 1 ? K : 1)];

  const int tid = threadIdx.x;
  constexpr int RPP = kSpmvRows / K;  // rows handled per pass when K lanes share a row
  const int c = K > 1 ? tid % K : 0;
  double dot_acc = 0.0;

  const int nblocks = (a.nrows + kSpmvRows - 1) / kSpmvRows;
  for (int rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
    const int row0 = rb * kSpmvRows;
    const int nr = min(kSpmvRows, a.nrows - row0);
    __syncthreads();  // previous pass finished with s_rp / tiles
    for (int t = tid; t <= nr; t += 256) s_rp[t] = a.rowptr[row0 + t];
    __syncthreads();
    const int kbeg = s_rp[0], kend = s_rp[nr];

    T acc[K];
#pragma unroll
    for (int p = 0; p < K; ++p) acc[p] = T(0);

    for (int ts = kbeg; ts < kend; ts += kSpmvTile) {
      const int te = min(kend, ts + kSpmvTile);
      if (ts != kbeg) __syncthreads();
      // ---- phase 2: coalesced stream of the tile
      for (int k = ts + tid; k < te; k += 256) {
        if (K == 1) {
          s_val[k - ts] = a.val[k] * a.x[a.col[k]];
        } else {
          s_val[k - ts] = a.val[k];
          s_col[k - ts] = a.col[k];
        }
      }
      __syncthreads();
      // ---- phase 3: per-row reduction out of LDS
      if (K == 1) {
        if (tid < nr) {
          const int lo = max(s_rp[tid], ts), hi = min(s_rp[tid + 1], te);
          T s = T(0);
          for (int k = lo; k < hi; ++k) s += s_val[k - ts];
          acc[0] += s;
        }
      } else {
#pragma unroll
        for (int p = 0; p < K; ++p) {
          const int r = tid / K + p * RPP;
          if (r < nr) {
            const int lo = max(s_rp[r], ts), hi = min(s_rp[r + 1], te);
            T s = T(0);
            for (int k = lo; k < hi; ++k) s += s_val[k - ts] * a.x[(size_t)s_col[k - ts] * K + c];
            acc[p] += s;
          }
        }
      }
    }
    // ---- phase 4: epilogue
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const int r = K == 1 ? tid : tid / K + p * RPP;
      if (K == 1 && p > 0) break;
      if (r < nr) {
        const size_t row = (size_t)(row0 + r);
        const size_t e = row * K + c;
        T v = acc[p];
        if (EPI == EPI_RESID) v = a.b[e] - v;
        if (EPI == EPI_JACOBI) v = a.x[e] + a.omega * a.dinv[row] * (a.b[e] - v);
        if (EPI == EPI_ADD) v = a.xadd[e] + v;
        a.y[e] = v;
        if (DOT) dot_acc += (double)a.dotw[e] * (double)v;
      }
    }
  }

  if (DOT) {
    // lanes with equal c sit K apart: reduce over lane bits >= log2(K), then across the 4 waves via LDS.
    double v = dot_acc;
#pragma unroll
    for (int o = 32; o >= (K > 1 ? K : 1); o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = tid & 63, w = tid >> 6;
    __syncthreads();
    if (lane < K) s_red[w * K + lane] = v;
    __syncthreads();
    if (tid < K) a.partials[(size_t)blockIdx.x * K + tid] = s_red[tid] + s_red[K + tid] + s_red[2 * K + tid] + s_red[3 * K + tid];
  }
}

// Number of workgroups launched for an nrows-row product (also the number of dot partials per column).
inline int spmv_grid(int nrows) {
  int nb = ceil_div(nrows, kSpmvRows);
  if (nb < 1) nb = 1;
  return nb < 4096 ? nb : 4096;
}

template <class T, int K, int EPI, bool DOT>
inline void spmv_launch_t(const SpmvArgs<T>& a, hipStream_t st) {
  hipLaunchKernelGGL((spmv_kernel<T, K, EPI, DOT>), dim3(spmv_grid(a.nrows)), dim3(256), 0, st, a);
}

template <class T, int K>
inline void spmv_launch(const SpmvArgs<T>& a, int epi, bool dot, hipStream_t st) {
  if (a.nrows <= 0) return;
  switch (epi) {
    case EPI_PLAIN:
      dot ? spmv_launch_t<T, K, EPI_PLAIN, true>(a, st) : spmv_launch_t<T, K, EPI_PLAIN, false>(a, st);
      break;
    case EPI_RESID:
      spmv_launch_t<T, K, EPI_RESID, false>(a, st);
      break;
    case EPI_JACOBI:
      dot ? spmv_launch_t<T, K, EPI_JACOBI, true>(a, st) : spmv_launch_t<T, K, EPI_JACOBI, false>(a, st);
      break;
    case EPI_ADD:
      spmv_launch_t<T, K, EPI_ADD, false>(a, st);
      break;
  }
}

// Convenience: y = A x (+ epilogue) for a Csr<T>.
template <class T>
inline SpmvArgs<T> spmv_args(const Csr<T>& A, const T* x, T* y) {
  SpmvArgs<T> a;
  a.nrows = A.nrows;
  a.rowptr = A.rp();
  a.col = A.ci();
  a.val = A.va();
  a.x = x;
  a.y = y;
  a.b = nullptr;
  a.xadd = nullptr;
  a.dinv = nullptr;
  a.omega = T(0);
  a.dotw = nullptr;
  a.partials = nullptr;
  return a;
}

}  // namespace csgpu
