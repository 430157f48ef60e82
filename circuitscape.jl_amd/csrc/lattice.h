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

// ---- residual update + restriction in ONE marching pass (round 6; VERDICT r5 item 4: measured) --
// r_out = r_in - alpha (A p)  AND  b_c = Q^T r_out, so that the restriction does not read r again (8 of ~80 bytes per node and
// column of a PCG iteration). A workgroup owns TIC coarse rows x segc coarse columns as in lattice_restrict_kernel and
// marches through the fine columns that feed them; per fine column it stages p (FR + 2 rows, four-slot ring: the nine-point
// product needs three columns) and the matrix rows, computes the updated residual of ALL FR staged rows -- its own and the
// one-tile halo above and below, whose values the neighbouring workgroups compute as well, bit for bit the same -- into an
// LDS slot, writes the OWNED entries to r_out, and accumulates the coarse sums from the slot. Because halo rows are
// recomputed from r_in, the update cannot be in place: r_in and r_out are different buffers (pcg.h ping-pongs them).
// Two barriers per fine column (ring ready -> compute; slot ready -> accumulate), which is also what lets the ring hold three
// columns and the residual slot one. 512 threads: TIC = 32 coarse rows at K = 32 (halo 8 of 104 staged rows), 130 KB of LDS.
// The arithmetic of an entry is that of dia_cg_kernel<..., DIA_RUPD> (same products, same order): r_out is bit-identical to
// the unfused update's; the coarse sums are lattice_restrict_kernel's, bit for bit. Only the order in which the partials of
// r'r are summed differs.
template <class T>
struct RupdRestrictArgs {
  int64_t n;
  int R, C, Rc, Cc;
  int nstrips, nseg, segc;
  const T* rows;        // lattice form of A, [n][5]
  const T* q;           // index-free Q, [n][9]
  const T* p;           // search direction, [n][K]
  const T* r_in;        // [n][K]
  T* r_out;             // [n][K], must not alias r_in
  T* bc;                // [Rc*Cc][K]
  const CgScalars* S;   // alpha[c], all_done
  double* partials;     // [gridDim.x][K] partials of r_out'r_out (may be null)
  // streaming pair solves (pcg_stream_pairs): a column with restart[c] != 0 takes a new pair -- its new residual is the
  // pair's right-hand side, -1 at node src[c] and +1 at dst[c] (nothing when they coincide), which stream_restart_kernel
  // writes AFTER the two-pass update; here the coarse sums need it at once
  const int* restart = nullptr;
  const int* src = nullptr;
  const int* dst = nullptr;
  // first update of a batch whose r0 was never materialised (pcg.h): r_in is not read, column c of it is -1 at node
  // isrc[c] and +1 at idst[c] (columns >= icols and pairs with isrc == idst: zero)
  const int* isrc = nullptr;
  const int* idst = nullptr;
  int icols = 0;
  // APPLY form of the kernel (a lattice V(2,2) level, pcg.h): r_out = D p and bc = Q^T p in one pass over p -- `rows` is the
  // lattice form of D (the level's two-sweep operator S), r_in / S / partials are unused, `skip` is the cycle's skip flag
  const int* skip = nullptr;
};

template <class T, int K, int NT>
struct RupdRestrictShape {
  typedef RestrictShape<T, K, NT> RS;
  static constexpr int CPL = RS::CPL, LPR = RS::LPR, TIC = RS::TIC, FR = RS::FR;
  static constexpr int PR = FR + 2;                         // rows of p / of the matrix staged per fine column
  // A thread updates BU ADJACENT rows of a fine column (rows tq * BU ... of its row group tq = tid / LPR): their 3 x 3
  // windows of p overlap, 3 (BU + 2) reads of the ring instead of 9 BU. The kernel is as close to the LDS's bandwidth as to
  // HBM's -- measured: section 8 of DESIGN.md, strided rows (one entry every NT / LPR rows) against adjacent ones.
  static constexpr int RG = NT / LPR;                       // row groups
  static constexpr int BU = (FR + RG - 1) / RG;             // residual entries (16-byte vectors) per thread and fine column
  static constexpr int PU = (PR * LPR + NT - 1) / NT;       // loads of p per thread and fine column
  static constexpr int MU = (PR * 5 + NT - 1) / NT;
  static constexpr int QU = (FR * 9 + NT - 1) / NT;
  static constexpr size_t lds_bytes = (size_t)3 * PR * LPR * 16 + (size_t)3 * PR * 5 * sizeof(T) + (size_t)FR * LPR * 16 +
                                      (size_t)FR * 9 * sizeof(T) + (size_t)(NT / 64) * K * sizeof(double);
};

template <class T, int K, int NT, bool APPLY = false>
__global__ __launch_bounds__(NT) void lattice_rupd_restrict_kernel(RupdRestrictArgs<T> a) {
  typedef RupdRestrictShape<T, K, NT> SH;
  constexpr int CPL = SH::CPL, LPR = SH::LPR, TIC = SH::TIC, FR = SH::FR, PR = SH::PR, BU = SH::BU, PU = SH::PU, MU = SH::MU,
                QU = SH::QU;
  typedef SpmvVec<T, CPL> XV;
  __shared__ XV s_p[3][PR * LPR];
  __shared__ T s_m[3][PR * 5];
  __shared__ XV s_b[FR * LPR];
  __shared__ T s_q[FR * 9];
  __shared__ double s_red[APPLY ? 1 : (NT / 64) * K];
  if (a.S && a.S->all_done) return;
  if (a.skip && *a.skip) return;
  const int tid = threadIdx.x;
  const int t = tid / LPR, lq = tid % LPR, c0 = lq * CPL;
  const int row0 = t * BU;  // first of this thread's adjacent rows in the update
  T alpha[CPL];
#pragma unroll
  for (int qq = 0; qq < CPL; ++qq) alpha[qq] = APPLY ? T(0) : (T)a.S->alpha[c0 + qq];
  double dot_acc[CPL];
  bool fresh[CPL];
  int64_t nsrc[CPL], ndst[CPL];
#pragma unroll
  for (int qq = 0; qq < CPL; ++qq) {
    dot_acc[qq] = 0.0;
    fresh[qq] = !APPLY && a.restart && a.restart[c0 + qq] != 0;
    nsrc[qq] = fresh[qq] ? a.src[c0 + qq] : -1;
    ndst[qq] = fresh[qq] ? a.dst[c0 + qq] : -1;
    if (nsrc[qq] == ndst[qq]) nsrc[qq] = ndst[qq] = -1;
  }
  const bool synth_in = !APPLY && a.isrc != nullptr;
  int64_t is_[CPL], id_[CPL];
#pragma unroll
  for (int qq = 0; qq < CPL; ++qq) {
    const bool on = synth_in && c0 + qq < a.icols && a.isrc[c0 + qq] != a.idst[c0 + qq];
    is_[qq] = on ? a.isrc[c0 + qq] : -1;
    id_[qq] = on ? a.idst[c0 + qq] : -1;
  }
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
    const int Ic = Ic0 + t;
    const bool row_on = Ic < a.Rc;
    const int f0 = max(3 * (Ic0 - 1), 0);
    const int last_tile = min(Ic0 + TIC, a.Rc - 1);
    const int f1 = last_tile >= a.Rc - 1 ? a.R : 3 * last_tile + 3;
    const int nfr = f1 - f0;  // <= FR
    const int fc0 = max(3 * (Jc0 - 1), 0);
    const int fc1 = Jc1 >= a.Cc - 1 ? a.C : 3 * Jc1 + 3;
    // the fine cells this workgroup OWNS (writes r_out for): the tiles of its own coarse rows / columns
    const int o0 = 3 * Ic0, o1 = (Ic0 + TIC >= a.Rc) ? a.R : 3 * (Ic0 + TIC);
    const int oc0 = 3 * Jc0, oc1 = (Jc1 >= a.Cc) ? a.C : 3 * Jc1;
    int rlo[3], rhi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tI = Ic - 1 + d;
      const bool on = row_on && tI >= 0 && tI < a.Rc;
      rlo[d] = on ? 3 * tI - f0 : 0;
      rhi[d] = on ? (tI >= a.Rc - 1 ? a.R : 3 * tI + 3) - f0 : 0;
    }
    XV preg[PU];
    T mreg[MU];
    XV rreg[BU];
    T qreg[QU];
    auto load_pm = [&](int f) {  // column f of p and of the matrix: staged rows f0 - 1 .. f0 + nfr (ids outside [0, n): zero)
      const int64_t base = (int64_t)f * a.R + f0 - 1;
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int e = tid + u * NT;
        XV v;
#pragma unroll
        for (int qq = 0; qq < CPL; ++qq) v.e[qq] = T(0);
        const int64_t id = base + e / LPR;
        if (e < (nfr + 2) * LPR && id >= 0 && id < a.n) v = *reinterpret_cast<const XV*>(a.p + (size_t)id * K + (e % LPR) * CPL);
        preg[u] = v;
      }
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * NT;
        const int64_t g = base * 5 + e;
        mreg[u] = (e < (nfr + 2) * 5 && g >= 0 && g < a.n * 5) ? a.rows[g] : T(0);
      }
    };
    auto store_pm = [&](int slot) {
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int e = tid + u * NT;
        if (e < PR * LPR) s_p[slot][e] = preg[u];
      }
#pragma unroll
      for (int u = 0; u < MU; ++u) {
        const int e = tid + u * NT;
        if (e < PR * 5) s_m[slot][e] = mreg[u];
      }
    };
    auto load_rq = [&](int f) {  // column f of r_in and of Q: staged rows f0 .. f0 + nfr - 1
      const int64_t base = (int64_t)f * a.R + f0;
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int row = row0 + u;
        XV v;
#pragma unroll
        for (int qq = 0; qq < CPL; ++qq) v.e[qq] = T(0);
        if (!APPLY && f < fc1 && row < nfr) {
          const int64_t id = base + row;
          if (synth_in) {
#pragma unroll
            for (int qq = 0; qq < CPL; ++qq) v.e[qq] = id == is_[qq] ? T(-1) : (id == id_[qq] ? T(1) : T(0));
          } else {
            v = *reinterpret_cast<const XV*>(a.r_in + (size_t)id * K + c0);
          }
        }
        rreg[u] = v;
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int e = tid + u * NT;
        qreg[u] = (f < fc1 && e < nfr * 9) ? a.q[base * 9 + e] : T(0);
      }
    };
    T acc[3][CPL];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int qq = 0; qq < CPL; ++qq) acc[d][qq] = T(0);
    auto emit = [&](int Jc, const T* v) {
      if (row_on && Jc >= Jc0 && Jc < Jc1) {
        XV o;
#pragma unroll
        for (int qq = 0; qq < CPL; ++qq) o.e[qq] = v[qq];
        *reinterpret_cast<XV*>(a.bc + ((size_t)Jc * a.Rc + Ic) * K + c0) = o;
      }
    };
    __syncthreads();  // previous tile finished with the rings
    // prologue: columns fc0 - 1 and fc0 of p / the matrix in the ring, column fc0 + 1 in flight; column fc0 of r / Q in flight
    load_pm(fc0 - 1);
    store_pm(0);
    load_pm(fc0);
    store_pm(1);
    load_pm(fc0 + 1);
    load_rq(fc0);
    int Jf = lat_tile(fc0, a.Cc);
    int sm = 0, sc = 1, sp = 2;  // ring slots of fine columns f - 1, f, f + 1
    for (int f = fc0; f < fc1; ++f) {
      store_pm(sp);  // (the slot column f - 2 had: its last readers passed the second barrier of the previous step)
      load_pm(f + 2);
      XV rv[BU];
      T qv[QU];
#pragma unroll
      for (int u = 0; u < BU; ++u) rv[u] = rreg[u];
#pragma unroll
      for (int u = 0; u < QU; ++u) qv[u] = qreg[u];
      load_rq(f + 1);
      __syncthreads();  // column f + 1 of the ring is in place; s_b / s_q are free (the previous step's accumulation is over)
      const T* mp = s_m[sm];
      const T* mc = s_m[sc];
      const XV* xm = s_p[sm];
      const XV* x0 = s_p[sc];
      const XV* xp = s_p[sp];
      const bool col_owned = f >= oc0 && f < oc1;
      // nine-point products of this thread's BU adjacent rows, one ring column at a time (the order of an entry's fma chain is
      // dia_cg_kernel's: column f - 1 rows -1, 0, +1, then f, then f + 1)
      T sacc[BU][CPL];
      XV ctr[BU];  // p at the entries themselves (APPLY restricts the input vector)
#pragma unroll
      for (int dj = 0; dj < 3; ++dj) {
        const XV* xcol = dj == 0 ? xm : (dj == 1 ? x0 : xp);
        XV win[BU + 2];
#pragma unroll
        for (int k = 0; k < BU + 2; ++k) {
#pragma unroll
          for (int qq = 0; qq < CPL; ++qq) win[k].e[qq] = T(0);
          if (row0 + k < nfr + 2) win[k] = xcol[(row0 + k) * LPR + lq];
        }
#pragma unroll
        for (int u = 0; u < BU; ++u) {
          const int row = row0 + u;          // staged residual row; its p / matrix row is row + 1
          if (row < nfr) {
            const int me = 5 * (row + 1);
            T w0, w1, w2;
            if (dj == 0) {
              w0 = mp[me - 5 + 4];
              w1 = mp[me + 3];
              w2 = mp[me + 5 + 2];
            } else if (dj == 1) {
              w0 = mc[me - 5 + 1];
              w1 = mc[me + 0];
              w2 = mc[me + 1];
            } else {
              w0 = mc[me + 2];
              w1 = mc[me + 3];
              w2 = mc[me + 4];
            }
#pragma unroll
            for (int qq = 0; qq < CPL; ++qq) {
              T sv = dj == 0 ? w0 * win[u].e[qq] : fma(w0, win[u].e[qq], sacc[u][qq]);
              sv = fma(w1, win[u + 1].e[qq], sv);
              sacc[u][qq] = fma(w2, win[u + 2].e[qq], sv);
            }
            if (dj == 1) ctr[u] = win[u + 1];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int row = row0 + u;
        if (row < nfr) {
          XV rn;
#pragma unroll
          for (int qq = 0; qq < CPL; ++qq)
            rn.e[qq] = APPLY ? sacc[u][qq] : (fresh[qq] ? T(0) : rv[u].e[qq] - alpha[qq] * sacc[u][qq]);
          const int fr = f0 + row;
          const bool owned = col_owned && fr >= o0 && fr < o1;
          if (owned) {  // (partials as the two-pass update writes them: a restarting column contributes nothing)
#pragma unroll
            for (int qq = 0; qq < CPL; ++qq) dot_acc[qq] += (double)rn.e[qq] * (double)rn.e[qq];
          }
          const int64_t id = (int64_t)f * a.R + fr;
#pragma unroll
          for (int qq = 0; qq < CPL; ++qq)
            if (fresh[qq]) rn.e[qq] = id == nsrc[qq] ? T(-1) : (id == ndst[qq] ? T(1) : T(0));
          s_b[row * LPR + lq] = APPLY ? ctr[u] : rn;  // (APPLY: the restriction is of the INPUT vector)
          if (owned) dia_store(reinterpret_cast<XV*>(a.r_out + (size_t)id * K + c0), rn);
        }
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int e = tid + u * NT;
        if (e < FR * 9) s_q[e] = qv[u];
      }
      __syncthreads();
      const int Jn = lat_tile(f, a.Cc);
      if (Jn != Jf) {  // entered the next tile column: coarse column Jf - 1 is complete
        emit(Jf - 1, acc[0]);
#pragma unroll
        for (int qq = 0; qq < CPL; ++qq) {
          acc[0][qq] = acc[1][qq];
          acc[1][qq] = acc[2][qq];
          acc[2][qq] = T(0);
        }
        Jf = Jn;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int sI = (1 - d) + 1;
        for (int r = rlo[d]; r < rhi[d]; ++r) {
          const XV bv = s_b[r * LPR + lq];
          const T* qr = s_q + r * 9;
          const T wm = qr[0 * 3 + sI];
          const T w0 = qr[1 * 3 + sI];
          const T wp = qr[2 * 3 + sI];
#pragma unroll
          for (int qq = 0; qq < CPL; ++qq) {
            acc[0][qq] = fma(wm, bv.e[qq], acc[0][qq]);
            acc[1][qq] = fma(w0, bv.e[qq], acc[1][qq]);
            acc[2][qq] = fma(wp, bv.e[qq], acc[2][qq]);
          }
        }
      }
      const int s0 = sm;
      sm = sc;
      sc = sp;
      sp = s0;
    }
    emit(Jf - 1, acc[0]);
    emit(Jf, acc[1]);
    emit(Jf + 1, acc[2]);
  }
  if (APPLY || !a.partials) return;  // (block-uniform)
  if constexpr (!APPLY) {
    const int lane = tid & 63, w = tid >> 6;
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < CPL; ++qq) {
      double v = dot_acc[qq];
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane < LPR) s_red[w * K + lane * CPL + qq] = v;
    }
    __syncthreads();
    if (tid < K) {
      double ssum = 0.0;
#pragma unroll
      for (int ww = 0; ww < NT / 64; ++ww) ssum += s_red[ww * K + tid];
      a.partials[(size_t)blockIdx.x * K + tid] = ssum;
    }
  }
}

// Threads per workgroup of the fused pass (build-time A/B knob). Measured at 10000^2, K = 32, fp64, one box
// (profiles/r6_fused_restrict_shape_ab.json; two-pass path: 46.2 - 46.9 pair-solves/s): 128 threads (8 coarse rows per
// workgroup, 8 halo rows of 32 staged, 3 workgroups per CU) 45.7 - 46.0; 256 (16 rows, 8 of 56, 2 per CU) 48.4 - 48.7;
// 512 (32 rows, 8 of 104, ONE workgroup of 130 KB per CU) 49.0, and 49.3 - 49.4 with 64 coarse columns per tile instead of 32.
#ifndef CSGPU_FUSED_NT
#define CSGPU_FUSED_NT 512
#endif
constexpr int kFusedNT = CSGPU_FUSED_NT;
// does the fused pass exist for this batch width?
template <class T, int K>
constexpr bool lattice_rupd_restrict_fits() {
  return K >= 16 && RupdRestrictShape<T, K, kFusedNT>::lds_bytes <= (kFusedNT >= 512 ? 160 : 80) * 1024;  // (256 threads: two workgroups per CU)
}

template <class T, int K>
inline int lattice_rupd_restrict_grid(const LatticeQ<T>& Q, int& nstrips, int& nseg, int& segc) {
  constexpr int NT = kFusedNT;
  segc = std::min(std::max(knobs().fused_seg, 2), Q.Cc);
  nstrips = ceil_div(Q.Rc, RestrictShape<T, K, NT>::TIC);
  // (a coarser lattice: narrower tiles rather than fewer than four per CU)
  while (segc > 16 && (int64_t)nstrips * ceil_div(Q.Cc, segc) < 1024) segc /= 2;
  nseg = ceil_div(Q.Cc, segc);
  int64_t g = (int64_t)nstrips * nseg;
  if (g > 16384) g = 16384;  // (rows of dot partials: PcgWork::ensure)
  if (g >= 64) g &= ~(int64_t)7;
  return (int)std::max<int64_t>(g, 1);
}

// r_out = r_in - alpha (A p) and bc = Q^T r_out in one pass; returns the number of partial rows written (0: none asked for)
template <class T, int K>
inline int lattice_rupd_restrict(const Dia<T>& D, const LatticeQ<T>& Q, const CgScalars* S, const T* p, const T* r_in, T* r_out,
                                 T* bc, double* partials, hipStream_t st, const int* restart = nullptr,
                                 const int* src = nullptr, const int* dst = nullptr, const int* isrc = nullptr,
                                 const int* idst = nullptr, int icols = 0) {
  if constexpr (lattice_rupd_restrict_fits<T, K>()) {
    constexpr int NT = kFusedNT;
    RupdRestrictArgs<T> a;
    a.n = D.n;
    a.R = Q.R;
    a.C = Q.C;
    a.Rc = Q.Rc;
    a.Cc = Q.Cc;
    const int g = lattice_rupd_restrict_grid<T, K>(Q, a.nstrips, a.nseg, a.segc);
    a.rows = D.data();
    a.q = Q.data();
    a.p = p;
    a.r_in = r_in;
    a.r_out = r_out;
    a.bc = bc;
    a.S = S;
    a.partials = partials;
    a.restart = restart;
    a.src = src;
    a.dst = dst;
    a.isrc = isrc;
    a.idst = idst;
    a.icols = icols;
    hipLaunchKernelGGL((lattice_rupd_restrict_kernel<T, K, NT>), dim3(g), dim3(NT), 0, st, a);
    return partials ? g : 0;
  } else {
    (void)D; (void)Q; (void)S; (void)p; (void)r_in; (void)r_out; (void)bc; (void)partials; (void)st; (void)restart; (void)src; (void)dst; (void)isrc; (void)idst; (void)icols;
    return 0;
  }
}

// y = D x and bc = Q^T x in one pass over x (the first two products of a lattice V(2,2) level: x = S b, b_c = Q2' b);
// false = this batch width has no fused form (the caller runs dia_apply + lattice_restrict)
template <class T, int K>
inline bool lattice_apply_restrict(const Dia<T>& D, const LatticeQ<T>& Q, const T* x, T* y, T* bc, const int* skip,
                                   hipStream_t st) {
  if constexpr (lattice_rupd_restrict_fits<T, K>()) {
    constexpr int NT = kFusedNT;
    RupdRestrictArgs<T> a;
    a.n = D.n;
    a.R = Q.R;
    a.C = Q.C;
    a.Rc = Q.Rc;
    a.Cc = Q.Cc;
    const int g = lattice_rupd_restrict_grid<T, K>(Q, a.nstrips, a.nseg, a.segc);
    a.rows = D.data();
    a.q = Q.data();
    a.p = x;
    a.r_in = nullptr;
    a.r_out = y;
    a.bc = bc;
    a.S = nullptr;
    a.partials = nullptr;
    a.skip = skip;
    hipLaunchKernelGGL((lattice_rupd_restrict_kernel<T, K, NT, true>), dim3(g), dim3(NT), 0, st, a);
    return true;
  } else {
    (void)D; (void)Q; (void)x; (void)y; (void)bc; (void)skip; (void)st;
    return false;
  }
}

// bc = Q^T b for the right-hand sides of a batch of pair solves (column c: -1 at node src[c], +1 at dst[c]; nothing when they
// coincide or c >= ncols): at most 18 entries per column instead of a pass over n x K. bc is zeroed by the caller. One thread
// per column; a coarse entry that takes both nodes' contributions gets w_dst - w_src rounded once, which is what the marching
// restriction's fma chain over the (otherwise zero) fine cells leaves there: same bits.
template <class T, int K>
__global__ __launch_bounds__(64) void lattice_restrict_pairs_kernel(int R, int Rc, int Cc, const T* __restrict__ q,
                                                                    const int* __restrict__ src, const int* __restrict__ dst,
                                                                    int ncols, T* __restrict__ bc) {
  const int c = threadIdx.x;
  if (c >= ncols || c >= K || src[c] == dst[c]) return;
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int64_t node = which == 0 ? src[c] : dst[c];
    const T sign = which == 0 ? T(-1) : T(1);
    const int i = (int)(node % R), j = (int)(node / R);
    const int I = lat_tile(i, Rc), J = lat_tile(j, Cc);
    for (int dj = 0; dj < 3; ++dj)
      for (int di = 0; di < 3; ++di) {
        const int Ic = I + di - 1, Jc = J + dj - 1;
        if (Ic < 0 || Ic >= Rc || Jc < 0 || Jc >= Cc) continue;
        const T w = q[(size_t)node * 9 + dj * 3 + di];
        T* at = bc + ((size_t)Jc * Rc + Ic) * K + c;
        *at = fma(w, sign, *at);
      }
  }
}

template <class T, int K>
inline void lattice_restrict_pairs(const LatticeQ<T>& Q, const int* src, const int* dst, int ncols, T* bc, hipStream_t st) {
  CS_HIP(hipMemsetAsync(bc, 0, (size_t)Q.Rc * Q.Cc * K * sizeof(T), st));
  hipLaunchKernelGGL((lattice_restrict_pairs_kernel<T, K>), dim3(1), dim3(64), 0, st, Q.R, Q.Rc, Q.Cc, Q.data(), src, dst, ncols,
                     bc);
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
