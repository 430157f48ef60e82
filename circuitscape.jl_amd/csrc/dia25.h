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
// streams a column's x entries (two halo rows above and below) through a two-slot LDS ring one step ahead and the column's
// 25 * TI matrix values into a double-buffered LDS tile; a lane keeps the 5 x 5 window of its row in registers. All HBM
// accesses are contiguous column segments; matrix bytes per row 200 instead of 304, no column indices, no gathers.
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

// smallest level that takes the 25-point form. Knobs::dia25_min_rows (csgpu_opts.dia25_min_rows): -1 = never, n = levels with at least n rows
// (default 16384: below that the level sits in the coarse tail kernel, tail.h, or costs microseconds either way)
inline int64_t dia25_min_rows() {
  const int64_t v = knobs().dia25_min_rows;
  return v < 0 ? (int64_t)0x7fffffffffffLL : v;
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

// JACOBI0: the first TWO sweeps of a level from a zero guess in one pass -- x1 = omega0 D^-1 b is formed while b is staged
// into the window (never written), y = x1 + omega D^-1 (b - A x1). Replaces scale_dinv_kernel + the first JACOBI pass of the
// V-cycle's pre-smoothing: three n x K vector passes less per level and cycle (write x1, read x1, read b twice -> once).
enum Dia25Epi { D25_PLAIN = 0, D25_RESID = 1, D25_JACOBI = 2, D25_JACOBI0 = 3 };

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
  T omega0;        // JACOBI0: weight of the sweep from zero that is folded into the load of the window
  const int* skip;
};

// y = A x | b - A x | x + omega dinv (b - A x), A in 25-point lattice form.
// The 5 x 5 window of x lives in REGISTERS (25 16-byte vectors per lane): marching one column on, a lane keeps 20 of them
// and reads the 5 of the new column from LDS. (Round-4 measurement, profiles/r4_nodata_10000_k32_dia25_ab.json: with the
// whole window read from an LDS ring -- 25 reads of x per lane and column -- the LDS pipe was as busy as HBM, per
// wavefront and column 25 x 8 clocks for x + 25 x 4 for the matrix row against 4 nodes x 968 B of HBM traffic at K = 32
// fp64; 486 ms per 16 pairs against 464 with the register window.) The x ring has two slots (the column being read, the
// column being written) and one barrier per column is enough. The column loop is unrolled five-fold so the window's slots
// are compile-time registers. PF: the column's b / dinv entries are loaded one column ahead as well (A/B knob
// CSGPU_DIA25_PF) -- without it every column waits out the latency of its own b load. WV: waves per SIMD the register
// allocation is held to (3: 168 VGPRs, a few spilled dwords in some instantiations; 1: no limit; A/B knob CSGPU_DIA25_WAVES).
template <class T, int K, int EPI, bool PF, int WV>
__global__ __launch_bounds__(256, WV) void dia25w_kernel(Dia25Args<T> a) {
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
    T dreg[XU];   // JACOBI0: D^-1 of the staged entries
    T mreg[MU];
    XV win[5][5];  // win[s][di]: x(i0 + t + di - 2, column with (column - j0 + 2) % 5 == s)
    auto load_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        XV v;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v.e[q] = T(0);
        if (EPI == D25_JACOBI0) dreg[u] = T(0);
        if (e < HR * LPR && jc >= 0 && jc < a.C) {
          const int row = i0 - 2 + e / LPR;
          if (row >= 0 && row < a.R) {
            if (EPI == D25_JACOBI0) {
              // (b and D^-1 only LOADED here; scaled when the registers go to LDS one column later -- scaling here would wait
              // for both loads on the spot and take the prefetch out of the pipeline: 1.00 instead of 0.72 ms per launch
              // at 4.0 M rows, profiles/r5_dia25_level1_6000.txt)
              v = *reinterpret_cast<const XV*>(a.b + ((size_t)jc * a.R + row) * K + (e % LPR) * CPL);
              dreg[u] = a.dinv[(size_t)jc * a.R + row];
            } else {
              v = *reinterpret_cast<const XV*>(a.x + ((size_t)jc * a.R + row) * K + (e % LPR) * CPL);
            }
          }
        }
        xreg[u] = v;
      }
    };
    auto store_x = [&](int jc) {
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int e = tid + u * 256;
        if (e < HR * LPR) {
          if (EPI == D25_JACOBI0) {
            // x1 = (omega0 * dinv) * b: the arithmetic of scale_dinv_kernel, bit for bit
            const T sc = a.omega0 * dreg[u];
            XV v = xreg[u];
#pragma unroll
            for (int q = 0; q < CPL; ++q) v.e[q] = sc * v.e[q];
            s_x[jc & 1][e] = v;
          } else {
            s_x[jc & 1][e] = xreg[u];
          }
        }
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
    XV b_pre;
    T d_pre = T(0);
#pragma unroll
    for (int q = 0; q < CPL; ++q) b_pre.e[q] = T(0);
    auto load_b = [&](int jc) {
      if (EPI != D25_PLAIN && row_on && jc < j1) {
        b_pre = *reinterpret_cast<const XV*>(a.b + ((size_t)jc * a.R + i0 + t) * K + c0);
        if (EPI == D25_JACOBI || EPI == D25_JACOBI0) d_pre = a.dinv[(size_t)jc * a.R + i0 + t];
      }
    };
    if (PF) load_b(j0);
    __syncthreads();
    for (int jb = j0; jb < j1; jb += 5) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int j = jb + u;
        if (j < j1) {  // (uniform over the workgroup)
          if (!PF) load_b(j);
          const XV bv = b_pre;
          const T dv = d_pre;
          if (PF) load_b(j + 1);
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
              if (EPI == D25_RESID) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) out.e[q] = bv.e[q] - acc[q];
              } else {
                const T sc = a.omega * dv;
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

inline bool dia25_prefetch_b() { return knobs().dia25_prefetch; }

// Register bound of the launch: 3 waves per SIMD where that costs at most a few spilled dwords (8 or more lanes per node:
// K = 32, K = 16 fp64; measured at K = 32: 325.3 against 327.7 ms per 16 pairs on the mixed path), no bound for the
// narrower batches (K = 8 fp64 would spill 52 B per lane under the bound; without it 2 waves per SIMD and no scratch).
inline int dia25_waves(int lanes_per_node) {
  const int w = knobs().dia25_waves;
  return w > 0 ? w : (lanes_per_node >= 8 ? 3 : 1);
}

template <class T, int K>
inline void dia25_launch(const Dia25<T>& D, int epi, const T* x, T* y, const T* b, const T* dinv, T omega, const int* skip,
                         hipStream_t st, T omega0 = T(0)) {
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
  a.omega0 = omega0;
  a.skip = skip;
  int64_t g = (int64_t)a.nstrips * a.nseg;
  if (g > 65536) g = 65536;
  if (g >= 64) g &= ~(int64_t)7;
  const dim3 grid((int)std::max<int64_t>(g, 1));
  const bool pf = dia25_prefetch_b(), w3 = dia25_waves(K / CPL) >= 3;
#define CS_D25_LAUNCH(E, P, W) hipLaunchKernelGGL((dia25w_kernel<T, K, E, P, W>), grid, dim3(256), 0, st, a)
#define CS_D25_EPI(E)                  \
  do {                                 \
    if (pf && w3) {                    \
      CS_D25_LAUNCH(E, true, 3);       \
    } else if (pf) {                   \
      CS_D25_LAUNCH(E, true, 1);       \
    } else if (w3) {                   \
      CS_D25_LAUNCH(E, false, 3);      \
    } else {                           \
      CS_D25_LAUNCH(E, false, 1);      \
    }                                  \
  } while (0)
  if (epi == D25_PLAIN) {
    CS_D25_LAUNCH(D25_PLAIN, true, 3);
  } else if (epi == D25_RESID) {
    CS_D25_EPI(D25_RESID);
  } else if (epi == D25_JACOBI0) {
    CS_D25_EPI(D25_JACOBI0);
  } else {
    CS_D25_EPI(D25_JACOBI);
  }
#undef CS_D25_EPI
#undef CS_D25_LAUNCH
}

}  // namespace csgpu
