// dia25.h -- 25-point lattice form of a coarse-level operator and its marching product (round 4).
//
// Level 1 of a raster hierarchy is a lattice again (the 3x3 tiles of level 0 in column-major order). When the tiles are
// regular its Galerkin operator is nine-point and the level runs in the collapsed lattice form of lattice_setup.h
// (lattice_level1_setup). When they are REFINED -- cell-space rasters with NODATA cells, strength-aware tiles: an
// aggregate then holds cells of a neighbouring tile -- the operator couples tiles up to two apart (5 x 5 window, 25
// points), lattice_level1_setup declines, and the level used to run its three products with A (two Jacobi sweeps and the
// residual of the V(2,2) level) through the CSR SpMM: 304 B of matrix per row and gathered x rows.
// Here the operator is stored index-free, 25 values per row (slot (dJ + 2) * 5 + (dI + 2) = A[(i, j), (i + dI, j + dJ)],
// 0 where absent; the full window, not the symmetric half: the kernel then needs no neighbour's matrix row), and the
// product marches like the nine-point kernels of stencil.h: a workgroup owns TI rows x SEG columns of the level's lattice,
// streams a column's x entries (two halo rows above and below) into an 8-slot LDS ring one step ahead and the column's
// 25 * TI matrix values into a double-buffered LDS tile, and every lane takes its 25 products out of LDS. All HBM accesses
// are contiguous column segments; matrix bytes per row 200 instead of 304, no column indices, no gathers.
// GPU counterpart of the smoother / residual products of AlgebraicMultigrid.jl's V-cycle on that level (reference call
// sites src/core.jl:164-167, 178).
#pragma once
#include "spmv.h"

namespace csgpu {

template <class T>
struct Dia25 {
  int64_t n = 0;
  int R = 0;   // lattice period (rows of the level's lattice)
  DBuf rows;   // [n][25] of T
  const T* data() const { return rows.as<T>(); }
  size_t device_bytes() const { return rows.bytes; }
};

// scatter the CSR entries of a lattice operator into the 25-slot rows; bad |= 1 when an entry lies outside the 5 x 5 window
template <class T>
__global__ __launch_bounds__(256) void dia25_fill_kernel(int n, int R, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const T* __restrict__ va, T* __restrict__ rows, int* __restrict__ bad) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int ri = i % R, cj = i / R;
    for (int k = rp[i]; k < rp[i + 1]; ++k) {
      const int c = ci[k];
      const int dI = c % R - ri, dJ = c / R - cj;
      if (dI < -2 || dI > 2 || dJ < -2 || dJ > 2) {
        atomicOr(bad, 1);
        continue;
      }
      rows[(size_t)i * 25 + (dJ + 2) * 5 + (dI + 2)] = va[k];
    }
  }
}

// smallest level that takes the 25-point form. A/B knob CSGPU_DIA25: 0 = never, n > 1 = levels with at least n rows
// (default 16384: below that the level sits in the coarse tail kernel, tail.h, or costs microseconds either way)
inline int64_t dia25_min_rows() {
  const char* e = getenv("CSGPU_DIA25");  // (read at every set-up: the tests switch it inside one process)
  if (!e || atoll(e) == 1) return 16384;
  return atoll(e) <= 0 ? (int64_t)0x7fffffffffffLL : (int64_t)atoll(e);
}

template <class T>
inline bool dia25_from_csr(const Csr<T>& A, int R, Dia25<T>& out, hipStream_t st) {
  const int n = A.nrows;
  if (A.nrows != A.ncols || R < 6 || n < 6 * R || (n % R) != 0) return false;
  DBuf rows((size_t)n * 25 * sizeof(T));
  DBuf bad = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(rows.p, 0, rows.bytes, st));
  CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  hipLaunchKernelGGL((dia25_fill_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, n, R, A.rp(), A.ci(), A.va(), dptr<T>(rows),
                     dptr<int>(bad));
  check_launch("25-point lattice form");
  if (read_int(dptr<int>(bad), st) != 0) return false;
  out.n = n;
  out.R = R;
  out.rows = std::move(rows);
  return true;
}

enum Dia25Epi { D25_PLAIN = 0, D25_RESID = 1, D25_JACOBI = 2 };

template <class T>
struct Dia25Args {
  int64_t n;
  int R, C;
  int nstrips, nseg, seg;
  const T* rows;   // [n][25]
  const T* x;      // [n][K]
  T* y;            // [n][K]
  const T* b;      // RESID / JACOBI
  const T* dinv;   // JACOBI
  T omega;         // JACOBI: y = x + omega dinv (b - A x)
  const int* skip;
};

// y = A x | b - A x | x + omega dinv (b - A x), A in 25-point lattice form
template <class T, int K, int EPI>
__global__ __launch_bounds__(256) void dia25_kernel(Dia25Args<T> a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPL = K < VEC ? K : VEC;
  constexpr int LPR = K / CPL;
  constexpr int TI = 256 / LPR;
  constexpr int HR = TI + 4;                       // rows staged per column (two halo rows above / below)
  constexpr int XU = (HR * LPR + 255) / 256;       // 16-byte loads of x per lane and column
  constexpr int MU = (25 * TI + 255) / 256;        // matrix values per lane and column
  typedef SpmvVec<T, CPL> XV;
  __shared__ XV s_x[8][HR * LPR];
  __shared__ T s_m[2][25 * TI];
  if (a.skip && *a.skip) return;
  const int tid = threadIdx.x;
  const int t = tid / LPR, lq = tid % LPR, c0 = lq * CPL;
  const int ntiles = a.nstrips * a.nseg;
  int t_first = blockIdx.x, t_last = ntiles, t_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {  // XCD-aware tile walk (see dia_cg_kernel)
    const int xcd = blockIdx.x & 7, chunk = (ntiles + 7) >> 3;
    t_first = xcd * chunk + (blockIdx.x >> 3);
    t_last = min(ntiles, (xcd + 1) * chunk);
    t_step = gridDim.x >> 3;
  }
  for (int tile = t_first; tile < t_last; tile += t_step) {
    const int si = tile % a.nstrips, sj = tile / a.nstrips;
    const int i0 = si * TI;
    const int j0 = sj * a.seg, j1 = min(a.C, j0 + a.seg);
    const bool row_on = i0 + t < a.R;
    XV xreg[XU];
    T mreg[MU];
    // column jc of x into registers: staged rows i0 - 2 .. i0 + TI + 1, zero outside the lattice
    auto load_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        XV v;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v.e[q] = T(0);
        if (e < HR * LPR && jc >= 0 && jc < a.C) {
          const int row = i0 - 2 + e / LPR;
          if (row >= 0 && row < a.R) v = *reinterpret_cast<const XV*>(a.x + ((size_t)jc * a.R + row) * K + (e % LPR) * CPL);
        }
        xreg[u] = v;
      }
    };
    auto store_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        if (e < HR * LPR) s_x[jc & 7][e] = xreg[u];
      }
    };
    // matrix values of column jc, rows i0 .. i0 + TI - 1 (contiguous: 25 per row)
    auto load_m = [&](int jc) {
      const int64_t base = ((int64_t)jc * a.R + i0) * 25;
      const int nval = 25 * min(TI, a.R - i0);
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        mreg[u] = (e < nval && jc < a.C) ? a.rows[base + e] : T(0);
      }
    };
    auto store_m = [&](int jc) {
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        if (e < 25 * TI) s_m[jc & 1][e] = mreg[u];
      }
    };
    __syncthreads();  // previous tile finished with the ring
    // prologue: x columns j0 - 2 .. j0 + 2 and the matrix of column j0 into LDS; x column j0 + 3 / matrix j0 + 1 in flight
    for (int jc = j0 - 2; jc <= j0 + 2; ++jc) {
      load_x(jc);
      store_x(jc);
    }
    load_m(j0);
    store_m(j0);
    load_x(j0 + 3);
    load_m(j0 + 1);
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
      if (row_on) {
        const T* m = s_m[j & 1] + 25 * t;
        T acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = T(0);
#pragma unroll
        for (int dj = 0; dj < 5; ++dj) {
          const XV* xc = s_x[(j + dj - 2) & 7] + (size_t)t * LPR + lq;  // staged row of lattice row i0 + t - 2
#pragma unroll
          for (int di = 0; di < 5; ++di) {
            const T w = m[dj * 5 + di];
            const XV xv = xc[di * LPR];
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[q] = fma(w, xv.e[q], acc[q]);
          }
        }
        const size_t e0 = ((size_t)j * a.R + i0 + t) * K + c0;
        XV out;
        if (EPI == D25_PLAIN) {
#pragma unroll
          for (int q = 0; q < CPL; ++q) out.e[q] = acc[q];
        } else {
          const XV bv = *reinterpret_cast<const XV*>(a.b + e0);
          if (EPI == D25_RESID) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) out.e[q] = bv.e[q] - acc[q];
          } else {
            const XV xs = s_x[j & 7][(size_t)(t + 2) * LPR + lq];
            const T sc = a.omega * a.dinv[(size_t)j * a.R + i0 + t];
#pragma unroll
            for (int q = 0; q < CPL; ++q) out.e[q] = xs.e[q] + sc * (bv.e[q] - acc[q]);
          }
        }
        *reinterpret_cast<XV*>(a.y + e0) = out;
      }
      __syncthreads();            // everybody is done with x column j - 2 and the matrix tile of column j
      store_x(j + 3);             // (slot (j + 3) & 7 = that of column j - 5: long since free)
      store_m(j + 1);             // (slot (j + 1) & 1 = that of column j - 1)
      if (j + 4 <= j1 + 1) load_x(j + 4);
      if (j + 2 < j1) load_m(j + 2);
      __syncthreads();
    }
  }
}

// The same product with the 5 x 5 window of x in REGISTERS (25 16-byte vectors per lane): marching one column on, a lane
// keeps 20 of them and reads the 5 of the new column from LDS -- 5 LDS reads of x per lane and column instead of 25. (With
// the ring kernel above the LDS pipe is as busy as HBM: per wavefront and column 25 x 8 clocks for x + 25 x 4 for the
// matrix row against 4 nodes x 968 B of HBM traffic at K = 32 fp64.) The x ring shrinks to two slots (the column being
// read, the column being written) and one barrier per column is enough. The column loop is unrolled five-fold so the
// window's slots are compile-time registers.
template <class T, int K, int EPI>
__global__ __launch_bounds__(256) void dia25w_kernel(Dia25Args<T> a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPL = K < VEC ? K : VEC;
  constexpr int LPR = K / CPL;
  constexpr int TI = 256 / LPR;
  constexpr int HR = TI + 4;
  constexpr int XU = (HR * LPR + 255) / 256;
  constexpr int MU = (25 * TI + 255) / 256;
  typedef SpmvVec<T, CPL> XV;
  __shared__ XV s_x[2][HR * LPR];
  __shared__ T s_m[2][25 * TI];
  if (a.skip && *a.skip) return;
  const int tid = threadIdx.x;
  const int t = tid / LPR, lq = tid % LPR, c0 = lq * CPL;
  const int ntiles = a.nstrips * a.nseg;
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
    const bool row_on = i0 + t < a.R;
    XV xreg[XU];
    T mreg[MU];
    XV win[5][5];  // win[s][di]: x(i0 + t + di - 2, column with (column - j0 + 2) % 5 == s)
    auto load_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        XV v;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v.e[q] = T(0);
        if (e < HR * LPR && jc >= 0 && jc < a.C) {
          const int row = i0 - 2 + e / LPR;
          if (row >= 0 && row < a.R) v = *reinterpret_cast<const XV*>(a.x + ((size_t)jc * a.R + row) * K + (e % LPR) * CPL);
        }
        xreg[u] = v;
      }
    };
    auto store_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        if (e < HR * LPR) s_x[jc & 1][e] = xreg[u];
      }
    };
    auto load_m = [&](int jc) {
      const int64_t base = ((int64_t)jc * a.R + i0) * 25;
      const int nval = 25 * min(TI, a.R - i0);
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        mreg[u] = (e < nval && jc < a.C) ? a.rows[base + e] : T(0);
      }
    };
    auto store_m = [&](int jc) {
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * 256;
        if (e < 25 * TI) s_m[jc & 1][e] = mreg[u];
      }
    };
    // prologue: columns j0 - 2 .. j0 + 1 through LDS into window slots 0 .. 3
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int jc = j0 - 2 + s;
      load_x(jc);
      __syncthreads();  // (previous tile / previous prologue step finished with the slot)
      store_x(jc);
      __syncthreads();
#pragma unroll
      for (int di = 0; di < 5; ++di) win[s][di] = s_x[jc & 1][(size_t)(t + di) * LPR + lq];
    }
    __syncthreads();
    load_x(j0 + 2);
    load_m(j0);
    store_x(j0 + 2);
    store_m(j0);
    load_x(j0 + 3);
    load_m(j0 + 1);
    __syncthreads();
    for (int jb = j0; jb < j1; jb += 5) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int j = jb + u;
        if (j < j1) {  // (uniform over the workgroup)
          // newest column of the window: j + 2 -> slot (u + 4) % 5
#pragma unroll
          for (int di = 0; di < 5; ++di) win[(u + 4) % 5][di] = s_x[(j + 2) & 1][(size_t)(t + di) * LPR + lq];
          if (row_on) {
            const T* m = s_m[j & 1] + 25 * t;
            T acc[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[q] = T(0);
#pragma unroll
            for (int dj = 0; dj < 5; ++dj) {
#pragma unroll
              for (int di = 0; di < 5; ++di) {
                const T w = m[dj * 5 + di];
#pragma unroll
                for (int q = 0; q < CPL; ++q) acc[q] = fma(w, win[(u + dj) % 5][di].e[q], acc[q]);
              }
            }
            const size_t e0 = ((size_t)j * a.R + i0 + t) * K + c0;
            XV out;
            if (EPI == D25_PLAIN) {
#pragma unroll
              for (int q = 0; q < CPL; ++q) out.e[q] = acc[q];
            } else {
              const XV bv = *reinterpret_cast<const XV*>(a.b + e0);
              if (EPI == D25_RESID) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) out.e[q] = bv.e[q] - acc[q];
              } else {
                const T sc = a.omega * a.dinv[(size_t)j * a.R + i0 + t];
#pragma unroll
                for (int q = 0; q < CPL; ++q) out.e[q] = win[(u + 2) % 5][2].e[q] + sc * (bv.e[q] - acc[q]);
              }
            }
            *reinterpret_cast<XV*>(a.y + e0) = out;
          }
          // column j + 3 / matrix j + 1 into the slots nobody reads during this step; next loads in flight
          store_x(j + 3);
          store_m(j + 1);
          if (j + 4 <= j1 + 1) load_x(j + 4);
          if (j + 2 < j1) load_m(j + 2);
          __syncthreads();
        }
      }
    }
  }
}

// kernel choice (A/B knob CSGPU_DIA25_KERNEL=ring|window)
inline bool dia25_window() {
  const char* e = getenv("CSGPU_DIA25_KERNEL");  // (read at every launch: the tests switch it inside one process)
  return e ? (e[0] == 'w') : true;
}

template <class T, int K>
inline void dia25_launch(const Dia25<T>& D, int epi, const T* x, T* y, const T* b, const T* dinv, T omega, const int* skip,
                         hipStream_t st) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPL = K < VEC ? K : VEC;
  constexpr int TI = 256 / (K / CPL);
  Dia25Args<T> a;
  a.n = D.n;
  a.R = D.R;
  a.C = (int)(D.n / D.R);
  a.seg = std::min(32, std::max(a.C, 1));
  a.nstrips = ceil_div(D.R, TI);
  a.nseg = ceil_div(a.C, a.seg);
  a.rows = D.data();
  a.x = x;
  a.y = y;
  a.b = b;
  a.dinv = dinv;
  a.omega = omega;
  a.skip = skip;
  int64_t g = (int64_t)a.nstrips * a.nseg;
  if (g > 65536) g = 65536;
  if (g >= 64) g &= ~(int64_t)7;
  const dim3 grid((int)std::max<int64_t>(g, 1));
  if (dia25_window()) {
    if (epi == D25_PLAIN)
      hipLaunchKernelGGL((dia25w_kernel<T, K, D25_PLAIN>), grid, dim3(256), 0, st, a);
    else if (epi == D25_RESID)
      hipLaunchKernelGGL((dia25w_kernel<T, K, D25_RESID>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((dia25w_kernel<T, K, D25_JACOBI>), grid, dim3(256), 0, st, a);
    return;
  }
  if (epi == D25_PLAIN)
    hipLaunchKernelGGL((dia25_kernel<T, K, D25_PLAIN>), grid, dim3(256), 0, st, a);
  else if (epi == D25_RESID)
    hipLaunchKernelGGL((dia25_kernel<T, K, D25_RESID>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((dia25_kernel<T, K, D25_JACOBI>), grid, dim3(256), 0, st, a);
}

}  // namespace csgpu
