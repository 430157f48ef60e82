// lattice.h -- index-free transfer operators of the two-product V(1,1) level on a raster lattice.
//
// On an all-valid raster the aggregation of amg_setup.h produces the regular 3x3 tiles (the pattern the reference's
// greedy StandardAggregation yields on rasters), so row (i, j) of Q = P - w D^-1 A P couples the fine
// cell to the 3 x 3 block of tiles around its own tile and nothing else. Q is then stored as 9 values per fine node
// with NO indices (LatticeQ, common.h), and both products of the level,
//     b_c = Q^T b                     (restriction;   lattice_restrict_kernel below)
//     out = S b + Q x_c               (second product; DIA_SQE mode of dia_cg_kernel, stencil.h)
// run as marching kernels whose every HBM access is a contiguous raster-column segment: the gathered operands (x_c
// for the second product, b for the restriction) are staged once per tile in LDS instead of being fetched through
// column indices. GPU counterpart of the restriction / prolongation products inside AlgebraicMultigrid.jl's V-cycle
// (reference call sites src/core.jl:164-167, 178).
//
// Algorithmic bytes: restriction n*K*sizeof(T) [b] + n*9*sizeof(T) [Q] + n_c*K*sizeof(T) [b_c];
// second product n*5*sizeof(T) [S] + n*9*sizeof(T) [Q] + 2*n*K*sizeof(T) [b, out] + n_c*K*sizeof(T) [x_c].
#pragma once
#include "stencil.h"

namespace csgpu {

__device__ __forceinline__ int lat_tile(int i, int nc) {  // tile index of fine row / column i (last tile absorbs the rest)
  const int t = i / 3;
  return t < nc ? t : nc - 1;
}

// agg[node] must be the row's own tile or one next to it; every Q entry must lie in the 3x3 tile block of its row
template <class T>
__global__ __launch_bounds__(256) void lattice_q_fill_kernel(int64_t n, int R, int Rc, int Cc, const int* __restrict__ agg,
                                                             const int* __restrict__ qrp, const int* __restrict__ qci,
                                                             const T* __restrict__ qva, T* __restrict__ q,
                                                             int* __restrict__ bad) {
  for (int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x; node < n; node += (int64_t)gridDim.x * 256) {
    const int i = (int)(node % R), j = (int)(node / R);
    const int I = lat_tile(i, Rc), J = lat_tile(j, Cc);
    {  // the row's aggregate: its own tile, or (cell-space rasters, amg_setup.h tile_pieces_kernel) a neighbouring one
      const int a = agg[node];
      const int dI = a % Rc - I, dJ = a / Rc - J;
      if (dI < -1 || dI > 1 || dJ < -1 || dJ > 1) {
        atomicOr(bad, 1);
        continue;
      }
    }
    T row[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) row[s] = T(0);
    for (int k = qrp[node]; k < qrp[node + 1]; ++k) {
      const int c = qci[k];
      const int dI = c % Rc - I, dJ = c / Rc - J;
      if (dI < -1 || dI > 1 || dJ < -1 || dJ > 1) {
        atomicOr(bad, 2);
        continue;
      }
      const int slot = (dJ + 1) * 3 + (dI + 1);
#pragma unroll
      for (int s = 0; s < 9; ++s)
        if (s == slot) row[s] = qva[k];
    }
#pragma unroll
    for (int s = 0; s < 9; ++s) q[node * 9 + s] = row[s];
  }
}

// Index-free form of Q (fine lattice R x C, aggregates `agg`, nc coarse nodes); false when the aggregates are not the
// regular tiles or Q reaches beyond the neighbouring tiles (the CSR kernels are used then).
template <class T>
inline bool lattice_q_from_csr(const Csr<T>& Q, const int* agg, int R, int C, LatticeQ<T>& out, hipStream_t st) {
  const int64_t n = (int64_t)R * C;
  const int Rc = (R + 1) / 3, Cc = (C + 1) / 3;
  if (Q.nrows != n || Rc < 2 || Cc < 2 || (int64_t)Rc * Cc != Q.ncols) return false;
  DBuf q((size_t)n * 9 * sizeof(T));
  DBuf bad = dalloc<int>(1);
  CS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
  hipLaunchKernelGGL((lattice_q_fill_kernel<T>), dim3(grid_for(n)), dim3(256), 0, st, n, R, Rc, Cc, agg, Q.rp(), Q.ci(),
                     Q.va(), dptr<T>(q), dptr<int>(bad));
  check_launch("index-free Q");
  if (read_int(dptr<int>(bad), st) != 0) return false;
  out.n = n;
  out.R = R;
  out.C = C;
  out.Rc = Rc;
  out.Cc = Cc;
  out.q = std::move(q);
  return true;
}

// ---- restriction b_c = Q^T b --------------------------------------------------------------------------------------
// A workgroup of NT threads owns TIC = NT / LPR coarse rows (LPR lanes cover the K columns of one coarse node) of a
// range of coarse columns and marches through the FINE raster columns that feed them: fine column f contributes to
// the coarse columns J(f)-1, J(f), J(f)+1, whose running sums live in three register accumulators that rotate when
// the march enters the next tile column. Per fine column the tile's rows of b (K-wide) and of Q (9 values) are
// streamed into a 3-slot LDS ring with contiguous loads; every lane then walks the rows of its three neighbouring
// tiles. Fixed summation order (fine columns ascending, rows ascending): deterministic.
template <class T>
struct RestrictArgs {
  int R, C, Rc, Cc;
  int nstrips, nseg, segc;  // tiles: nstrips strips of TIC coarse rows x nseg segments of segc coarse columns
  const T* q;               // [n][9]
  const T* b;               // [n][K]
  T* bc;                    // [Rc*Cc][K]
  const int* skip;
};

template <class T, int K, int NT>
struct RestrictShape {
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int CPL = K < VEC ? K : VEC;
  static constexpr int LPR = K / CPL;
  static constexpr int TIC = NT / LPR;       // coarse rows per workgroup
  static constexpr int FR = 3 * TIC + 8;     // fine rows staged per column (one tile above, one below, last-tile slack)
  static constexpr int BU = (FR * LPR + NT - 1) / NT;  // 16-byte loads of b per thread and fine column
  static constexpr int QU = (FR * 9 + NT - 1) / NT;    // 4/8-byte loads of Q per thread and fine column
};

template <class T, int K, int NT>
__global__ __launch_bounds__(NT) void lattice_restrict_kernel(RestrictArgs<T> a) {
  typedef RestrictShape<T, K, NT> SH;
  constexpr int CPL = SH::CPL, LPR = SH::LPR, TIC = SH::TIC, FR = SH::FR, BU = SH::BU, QU = SH::QU;
  typedef SpmvVec<T, CPL> XV;
  __shared__ XV s_b[3][FR * LPR];
  __shared__ T s_q[3][FR * 9];
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
    const int Ic0 = si * TIC;
    const int Jc0 = sj * a.segc, Jc1 = min(a.Cc, Jc0 + a.segc);
    const int Ic = Ic0 + t;                         // this lane's coarse row
    const bool row_on = Ic < a.Rc;
    const int f0 = max(3 * (Ic0 - 1), 0);           // first fine row staged
    // fine rows up to the end of tile Ic0 + TIC (the tile below the strip's last row), clipped to the raster
    const int last_tile = min(Ic0 + TIC, a.Rc - 1);
    const int f1 = last_tile >= a.Rc - 1 ? a.R : 3 * last_tile + 3;
    const int nfr = f1 - f0;                        // <= FR
    // fine columns feeding coarse columns [Jc0, Jc1): tiles Jc0-1 .. Jc1
    const int fc0 = max(3 * (Jc0 - 1), 0);
    const int fc1 = Jc1 >= a.Cc - 1 ? a.C : 3 * Jc1 + 3;
    // rows of this lane's three tiles: [rlo[d], rhi[d]) for tile Ic - 1 + d, as offsets into the staged column
    int rlo[3], rhi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tI = Ic - 1 + d;
      const bool on = row_on && tI >= 0 && tI < a.Rc;
      rlo[d] = on ? 3 * tI - f0 : 0;
      rhi[d] = on ? (tI >= a.Rc - 1 ? a.R : 3 * tI + 3) - f0 : 0;
    }
    XV breg[BU];
    T qreg[QU];
    auto load_column = [&](int f) {
      const int64_t base = (int64_t)f * a.R + f0;
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int e = tid + u * NT;
        XV v;
#pragma unroll
        for (int q = 0; q < CPL; ++q) v.e[q] = T(0);
        if (e < nfr * LPR) v = *reinterpret_cast<const XV*>(a.b + (size_t)(base + e / LPR) * K + (e % LPR) * CPL);
        breg[u] = v;
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int e = tid + u * NT;
        qreg[u] = e < nfr * 9 ? a.q[base * 9 + e] : T(0);
      }
    };
    auto store_column = [&](int f) {
      const int slot = f % 3;
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int e = tid + u * NT;
        if (e < FR * LPR) s_b[slot][e] = breg[u];
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int e = tid + u * NT;
        if (e < FR * 9) s_q[slot][e] = qreg[u];
      }
    };
    T acc[3][CPL];  // running sums of coarse columns Jf - 1, Jf, Jf + 1 (Jf = tile column of the current fine column)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[d][q] = T(0);
    auto emit = [&](int Jc, const T* v) {
      if (row_on && Jc >= Jc0 && Jc < Jc1) {
        XV o;
#pragma unroll
        for (int q = 0; q < CPL; ++q) o.e[q] = v[q];
        *reinterpret_cast<XV*>(a.bc + ((size_t)Jc * a.Rc + Ic) * K + c0) = o;
      }
    };
    __syncthreads();  // previous tile finished with the ring
    load_column(fc0);
    int Jf = lat_tile(fc0, a.Cc);
    for (int f = fc0; f < fc1; ++f) {
      store_column(f);
      if (f + 1 < fc1) load_column(f + 1);
      __syncthreads();
      const int Jn = lat_tile(f, a.Cc);
      if (Jn != Jf) {  // entered the next tile column: coarse column Jf - 1 is complete
        emit(Jf - 1, acc[0]);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          acc[0][q] = acc[1][q];
          acc[1][q] = acc[2][q];
          acc[2][q] = T(0);
        }
        Jf = Jn;
      }
      const XV* sb = s_b[f % 3];
      const T* sq = s_q[f % 3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // rows of tile Ic - 1 + d see this lane's coarse row at dI = Ic - tile = 1 - d
        const int sI = (1 - d) + 1;
        for (int r = rlo[d]; r < rhi[d]; ++r) {
          const XV bv = sb[r * LPR + lq];
          const T* qr = sq + r * 9;
          const T wm = qr[0 * 3 + sI];  // dJ = -1: coarse column Jf - 1
          const T w0 = qr[1 * 3 + sI];  // dJ =  0
          const T wp = qr[2 * 3 + sI];  // dJ = +1
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            acc[0][q] = fma(wm, bv.e[q], acc[0][q]);
            acc[1][q] = fma(w0, bv.e[q], acc[1][q]);
            acc[2][q] = fma(wp, bv.e[q], acc[2][q]);
          }
        }
      }
    }
    emit(Jf - 1, acc[0]);
    emit(Jf, acc[1]);
    emit(Jf + 1, acc[2]);
  }
}

inline int lattice_segc() {  // coarse columns per restriction tile (tuning knob Knobs::restrict_seg)
  const int v = knobs().restrict_seg;
  return v < 2 ? 2 : v;
}

// bc = Q^T b
template <class T, int K>
inline void lattice_restrict(const LatticeQ<T>& Q, const T* b, T* bc, const int* skip, hipStream_t st) {
  constexpr int NT = 128;
  RestrictArgs<T> a;
  a.R = Q.R;
  a.C = Q.C;
  a.Rc = Q.Rc;
  a.Cc = Q.Cc;
  a.segc = std::min(lattice_segc(), Q.Cc);
  a.nstrips = ceil_div(Q.Rc, RestrictShape<T, K, NT>::TIC);
  a.nseg = ceil_div(Q.Cc, a.segc);
  a.q = Q.data();
  a.b = b;
  a.bc = bc;
  a.skip = skip;
  int64_t g = (int64_t)a.nstrips * a.nseg;
  if (g > 65536) g = 65536;
  if (g >= 64) g &= ~(int64_t)7;
  hipLaunchKernelGGL((lattice_restrict_kernel<T, K, NT>), dim3((int)std::max<int64_t>(g, 1)), dim3(NT), 0, st, a);
}

}  // namespace csgpu
