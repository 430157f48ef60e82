// enrich.h -- spectral enrichment of the level-0 coarse space on perforated / refined-tile raster lattices (round 5).
//
// Why. On a raster with NODATA cells the 3x3 tiles stay the aggregates of level 0 (that is what keeps the transfer
// operators index-free), and the piece analysis keeps every aggregate CONNECTED. Connected is not enough: a short wall of
// NODATA cells inside a tile leaves a C- or U-shaped aggregate whose two arms hang together through a one-cell neck, a
// bare diagonal bridge, a corridor along the raster's edge. The smoothed-aggregation coarse space carries ONE function per
// aggregate -- (smoothed) constants -- and cannot represent the low-energy mode "one arm up, the other down"; Jacobi
// damps it slowly because the neck makes it smooth. Measured with a scipy reconstruction of this very hierarchy (400^2 and
// 1000^2, 15 % i.i.d. NODATA): the spectrum of M^-1 A has the same bulk edge as on the all-valid raster (0.29-0.30) and
// below it a tail of LOCALISED eigenvectors (10-40 cells each, lambda 0.15 ... 0.29), every one of them a jump across such
// a wall inside one aggregate; the tail is what costs the iterations (12.3 against 10.0 at 400^2; it fills up as the raster
// grows: 13.8-16.3 at 10000^2) -- the reference's own hierarchy (greedy aggregation + Gauss-Seidel, oracle/) needs 16.9 at
// 400^2. Re-dealing the cells over the tiles (splitting the pieces along their Fiedler vectors) only moves the defects.
//
// What. Every aggregate whose local generalised eigenproblem  L_agg phi = lambda D phi  (L_agg: the couplings INSIDE the
// aggregate, D: the full diagonal) has a second eigenvalue below tau gets a SECOND coarse function: an approximation phi_t
// of that Fiedler vector (start: the coordinate direction with the smallest Rayleigh quotient, then a few power steps),
// supported on the aggregate's cells. The vectors E = [phi_t] form an auxiliary coarse space handled OUTSIDE the lattice
// hierarchy, multiplicatively and symmetrically around the V-cycle:
//
//     c  = B E'r,   r1 = r - A E c,   z = V(r1),   c2 = B E'(r1 - A z),   z += E (c + c2),       B = diag(E'AE)^-1 (row-
//                                                                              abs-sum scaled: B^-1 >= E'AE)
//
// i.e. M^-1 = E(2B - B G B)E' + (I - E B E'A) V (I - A E B E'): symmetric positive definite for any symmetric A used in
// it. All of it touches only the members of the enriched aggregates and their neighbours (2-5 % of the cells): gathers /
// scatters over compact lists, no atomics, fixed summation orders -- bit-reproducible. The r'z partials of the V-cycle's
// last product are corrected by the scalars the passes produce anyway (r'z = r1'z_V + c'(AE)'z_V + t'(c + c2)).
// Prototype (two-grid, exact coarse solve, 400^2 / 15 % NODATA): 12.5 -> 10.6 iterations with 4 % of the tiles enriched
// (tau = 0.08), 9.75 with 23 % (tau = 0.15; the all-valid raster: 9.9). The solve-phase passes run from PRECOMPUTED sparse
// forms of E and A E (by columns for the gathers, by rows over the halo cells for the residual update): the first version
// searched the lattice neighbourhoods in every pass and cost 25 ms per iteration at 10000^2 -- more than it saved.
//
// GPU counterpart of nothing in the reference: AlgebraicMultigrid.jl has no such step (its unstructured aggregates adapt
// to the holes instead; call site of the hierarchy src/core.jl:164-167). Parity is solution-level, as for the rest of
// the preconditioner.
#pragma once
#include "amg_setup.h"

namespace csgpu {

static const int kEnrichMaxM = 20;   // members of an aggregate the local analysis handles (more: not enriched)
static const int kEnrichWin = 8;     // window edge: a tile (<= 4 cells) and two cells around it

// k-th entry (k = 0..8, ascending column order) of row i of a lattice matrix: column j and value (0 where absent)
template <class U>
__device__ __forceinline__ U enr_entry(const U* __restrict__ rows, int64_t n, int R, int64_t i, int k, int64_t& j) {
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

// Pass A, one thread per tile: members of the aggregate (window scan), local Fiedler approximation, decision.
// phi (dense, pre-zeroed) receives the vector of an enriched aggregate; flag[t] = 1, mcount[t] = members.
// (55 ms at 10000^2 in its first form -- as much as the rest of the set-up's level 0; now: tiles that hold exactly their own
// cells with at most one hole leave after nine loads, neighbours by coordinate offsets instead of j % R, j / R, fp32
// weights in the thread's scratch.)
template <class U, class T>
__global__ __launch_bounds__(64) void enrich_phi_kernel(int R, int C, int Rc, int Cc, const U* __restrict__ rows,
                                                        const long long* __restrict__ size0, const int* __restrict__ agg,
                                                        const unsigned long long* __restrict__ size_c, double tau,
                                                        int psteps, T* __restrict__ phi, int* __restrict__ flag,
                                                        int* __restrict__ mcount) {
  const int64_t n = (int64_t)R * C;
  const int ntiles = Rc * Cc;
  for (int t = blockIdx.x * 64 + threadIdx.x; t < ntiles; t += gridDim.x * 64) {
    flag[t] = 0;
    mcount[t] = 0;
    const int want = (int)size_c[t];
    if (want < 3 || want > kEnrichMaxM) continue;
    const int I = t % Rc, J = t / Rc;
    int r0, r1, c0, c1;
    tile_extent(I, Rc, R, r0, r1);
    tile_extent(J, Cc, C, c0, c1);
    {
      // a tile that holds exactly its own cells and has at most one hole is compact: not enriched, and most tiles are such
      int own = 0, validc = 0;
      for (int c = c0; c < c1; ++c)
        for (int r = r0; r < r1; ++r) {
          const int64_t cell = (int64_t)c * R + r;
          const bool v = !size0 || size0[cell] != 0;
          validc += v ? 1 : 0;
          own += (v && agg[cell] == t) ? 1 : 0;
        }
      // (one hole: the local Fiedler value of such a tile stays above 0.075 -- measured; a larger tau looks at them)
      if (own == want && validc == own && (r1 - r0) * (c1 - c0) - validc <= (tau < 0.075 ? 1 : 0)) continue;
    }
    const int wr0 = max(r0 - 2, 0), wr1 = min(r1 + 2, R), wc0 = max(c0 - 2, 0), wc1 = min(c1 + 2, C);
    signed char widx[kEnrichWin * kEnrichWin];
    signed char mr[kEnrichMaxM], mc[kEnrichMaxM];   // window coordinates of the members
    int m = 0;
    for (int c = wc0; c < wc1; ++c)
      for (int r = wr0; r < wr1; ++r) {
        const int64_t cell = (int64_t)c * R + r;
        const bool mem = agg[cell] == t && (!size0 || size0[cell] != 0);
        widx[(c - wc0) * kEnrichWin + (r - wr0)] = (signed char)(mem && m < kEnrichMaxM ? m : -1);
        if (mem) {
          if (m < kEnrichMaxM) {
            mr[m] = (signed char)(r - wr0);
            mc[m] = (signed char)(c - wc0);
          }
          ++m;
        }
      }
    if (m != want) continue;  // (members outside the window: the aggregate is left alone)
    // local graph: couplings between members (k-th neighbour of a cell: column offset k / 3 - 1, row offset k % 3 - 1)
    float d[kEnrichMaxM], w[kEnrichMaxM][8];
    signed char nb[kEnrichMaxM][8];
    float dsum = 0.f;
    const int wh = wr1 - wr0, ww = wc1 - wc0;
    for (int a = 0; a < m; ++a) {
      const int64_t cell = (int64_t)(wc0 + mc[a]) * R + wr0 + mr[a];
      d[a] = (float)rows[cell * 5];
      dsum += d[a];
      int q = 0;
      for (int k = 0; k < 9; ++k) {
        if (k == 4) continue;
        const int jr = mr[a] + (k % 3) - 1, jc = mc[a] + (k / 3) - 1;
        int b = -1;
        float v = 0.f;
        if (jr >= 0 && jr < wh && jc >= 0 && jc < ww) {
          b = widx[jc * kEnrichWin + jr];
          if (b >= 0) {
            int64_t j;
            v = (float)enr_entry(rows, n, R, cell, k, j);
            if (v == 0.f) b = -1;
          }
        }
        nb[a][q] = (signed char)b;
        w[a][q] = v < 0.f ? -v : v;
        ++q;
      }
    }
    if (!(dsum > 0.f)) continue;
    float v[kEnrichMaxM], lv[kEnrichMaxM];
    auto center = [&](float* x) {
      float s = 0.f;
      for (int a = 0; a < m; ++a) s += d[a] * x[a];
      s /= dsum;
      for (int a = 0; a < m; ++a) x[a] -= s;
    };
    auto apply = [&](const float* x, float* y) {  // y = L_agg x
      for (int a = 0; a < m; ++a) {
        float s = 0.f;
        for (int q = 0; q < 8; ++q)
          if (nb[a][q] >= 0) s += w[a][q] * (x[a] - x[nb[a][q]]);
        y[a] = s;
      }
    };
    auto rayleigh = [&](const float* x, const float* y) {
      float num = 0.f, den = 0.f;
      for (int a = 0; a < m; ++a) {
        num += x[a] * y[a];
        den += d[a] * x[a] * x[a];
      }
      return den > 0.f ? num / den : 1e30f;
    };
    auto direction = [&](int dir, int a) {
      return dir == 0 ? (float)mr[a] : dir == 1 ? (float)mc[a] : dir == 2 ? (float)(mr[a] + mc[a]) : (float)(mr[a] - mc[a]);
    };
    // start: the coordinate direction (row, column, the two diagonals) with the smallest Rayleigh quotient
    float best = 1e30f;
    int bestdir = -1;
    for (int dir = 0; dir < 4; ++dir) {
      for (int a = 0; a < m; ++a) v[a] = direction(dir, a);
      center(v);
      apply(v, lv);
      const float q = rayleigh(v, lv);
      if (q < best) {
        best = q;
        bestdir = dir;
      }
    }
    if (bestdir < 0 || best > 1e29f) continue;
    if (best > 7.0f * (float)tau) continue;   // (the power steps below lower the quotient, but not by this much)
    for (int a = 0; a < m; ++a) v[a] = direction(bestdir, a);
    center(v);
    for (int s = 0; s < psteps; ++s) {  // power steps on I - 0.6 D^-1 L_agg (deflated against the constant)
      apply(v, lv);
      for (int a = 0; a < m; ++a) v[a] -= 0.6f * lv[a] / d[a];
      center(v);
    }
    apply(v, lv);
    const float lam = rayleigh(v, lv);
    if (!(lam < (float)tau)) continue;
    float den = 0.f;
    for (int a = 0; a < m; ++a) den += d[a] * v[a] * v[a];
    if (!(den > 0.f)) continue;
    const float sc = 1.0f / sqrtf(den);
    for (int a = 0; a < m; ++a) phi[(int64_t)(wc0 + mc[a]) * R + wr0 + mr[a]] = (T)(v[a] * sc);
    flag[t] = 1;
    mcount[t] = m;
  }
}

// Pass B, one thread per tile (enriched ones work): member list in window order (cell, phi); G_vv = phi'A phi and the sum
// of |phi_i A_ij phi_j| over the couplings into OTHER enriched aggregates (row-abs-sum scaling of B); number of entries of
// the vector's column of A E (cells of the window whose row of A meets the members)
template <class U, class T>
__global__ __launch_bounds__(64) void enrich_lists_kernel(int R, int C, int Rc, int Cc, const U* __restrict__ rows,
                                                          const int* __restrict__ agg, const T* __restrict__ phi,
                                                          const int* __restrict__ tile_of_vec, int nvec, const int* __restrict__ moff,
                                                          int* __restrict__ vptr, int* __restrict__ vcell, T* __restrict__ vphi,
                                                          double* __restrict__ binv, int* __restrict__ acount) {
  const int64_t n = (int64_t)R * C;
  for (int v = blockIdx.x * 64 + threadIdx.x; v < nvec; v += gridDim.x * 64) {
    const int t = tile_of_vec[v];
    const int I = t % Rc, J = t / Rc;
    int r0, r1, c0, c1;
    tile_extent(I, Rc, R, r0, r1);
    tile_extent(J, Cc, C, c0, c1);
    const int wr0 = max(r0 - 3, 0), wr1 = min(r1 + 3, R), wc0 = max(c0 - 3, 0), wc1 = min(c1 + 3, C);
    int o = moff[t];
    vptr[v] = o;
    double gvv = 0.0, off = 0.0;
    int na = 0;
    for (int c = wc0; c < wc1; ++c)
      for (int r = wr0; r < wr1; ++r) {
        const int64_t cell = (int64_t)c * R + r;
        const double pi = (double)phi[cell];
        const bool mem = agg[cell] == t && pi != 0.0;   // (phi is non-zero exactly at the members of enriched aggregates;
        if (mem) {                                        //  a member whose phi rounds to 0 contributes nothing anywhere)
          vcell[o] = (int)cell;
          vphi[o] = (T)pi;
          ++o;
        }
        double coef = 0.0;
        for (int k = 0; k < 9; ++k) {
          int64_t j;
          const double a = (double)enr_entry(rows, n, R, cell, k, j);
          if (a == 0.0) continue;
          const double pj = (double)phi[j];
          if (pj == 0.0) continue;
          if (agg[j] == t) {
            coef += a * pj;
            if (mem) gvv += pi * a * pj;
          } else if (mem) {
            off += fabs(pi * a * pj);
          }
        }
        if (coef != 0.0) ++na;
      }
    for (const int mend = moff[t + 1]; o < mend; ++o) {  // (a member whose phi rounded to 0: padding entries without weight)
      vcell[o] = vcell[moff[t]];
      vphi[o] = T(0);
    }
    binv[v] = gvv + off > 0.0 ? 1.0 / (gvv + off) : 0.0;
    acount[v] = na;
  }
}

// column v of A E: (cell, coefficient) in window order -- the same arithmetic as the count above
template <class U, class T>
__global__ __launch_bounds__(64) void enrich_ae_fill_kernel(int R, int C, int Rc, int Cc, const U* __restrict__ rows,
                                                            const int* __restrict__ agg, const T* __restrict__ phi,
                                                            const int* __restrict__ tile_of_vec, int nvec, const int* __restrict__ aptr,
                                                            int* __restrict__ acell, double* __restrict__ acoef) {
  const int64_t n = (int64_t)R * C;
  for (int v = blockIdx.x * 64 + threadIdx.x; v < nvec; v += gridDim.x * 64) {
    const int t = tile_of_vec[v];
    const int I = t % Rc, J = t / Rc;
    int r0, r1, c0, c1;
    tile_extent(I, Rc, R, r0, r1);
    tile_extent(J, Cc, C, c0, c1);
    const int wr0 = max(r0 - 3, 0), wr1 = min(r1 + 3, R), wc0 = max(c0 - 3, 0), wc1 = min(c1 + 3, C);
    int o = aptr[v];
    for (int c = wc0; c < wc1; ++c)
      for (int r = wr0; r < wr1; ++r) {
        const int64_t cell = (int64_t)c * R + r;
        double coef = 0.0;
        for (int k = 0; k < 9; ++k) {
          int64_t j;
          const double a = (double)enr_entry(rows, n, R, cell, k, j);
          if (a == 0.0) continue;
          const double pj = (double)phi[j];
          if (pj != 0.0 && agg[j] == t) coef += a * pj;
        }
        if (coef != 0.0) {
          acell[o] = (int)cell;
          acoef[o] = coef;
          ++o;
        }
      }
  }
}

// halo cells (rows of A E that are not empty): flag; then per halo cell the (vector, coefficient) entries of its row, in
// the order the vectors first appear among the cell's neighbours (at most kEnrichRowMax distinct ones)
static const int kEnrichRowMax = 6;

template <class U, class T, int PASS>
__global__ __launch_bounds__(256) void enrich_halo_kernel(int64_t n, int R, const U* __restrict__ rows, const int* __restrict__ agg,
                                                          const T* __restrict__ phi, const int* __restrict__ vec_of_tile,
                                                          int* __restrict__ hflag, const int* __restrict__ hcell, int nhalo,
                                                          int* __restrict__ hcount, const int* __restrict__ hptr,
                                                          int* __restrict__ hvec, double* __restrict__ hcoef,
                                                          int* __restrict__ overflow) {
  const int64_t total = PASS == 0 ? n + 1 : (int64_t)nhalo;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
    const int64_t i = PASS == 0 ? q : (int64_t)hcell[q];
    int vs[kEnrichRowMax];
    double cs[kEnrichRowMax];
    int nv = 0;
    if (i < n) {
      for (int k = 0; k < 9; ++k) {
        int64_t j;
        const double a = (double)enr_entry(rows, n, R, i, k, j);
        if (a == 0.0) continue;
        const double pj = (double)phi[j];
        if (pj == 0.0) continue;
        const int v = vec_of_tile[agg[j]];
        int s = 0;
        while (s < nv && vs[s] != v) ++s;
        if (s == nv) {
          if (nv == kEnrichRowMax) {
            atomicOr(overflow, 1);
            continue;
          }
          vs[nv] = v;
          cs[nv] = 0.0;
          ++nv;
        }
        cs[s] += a * pj;
      }
    }
    if (PASS == 0) {
      hflag[q] = nv > 0 ? 1 : 0;
    } else if (PASS == 1) {
      int cnt = 0;
      for (int s = 0; s < nv; ++s) cnt += cs[s] != 0.0 ? 1 : 0;
      hcount[q] = cnt;
    } else {
      int o = hptr[q];
      for (int s = 0; s < nv; ++s)
        if (cs[s] != 0.0) {
          hvec[o] = vs[s];
          hcoef[o] = cs[s];
          ++o;
        }
    }
  }
}

__global__ __launch_bounds__(256) void enrich_halo_fill_kernel(int64_t n, const int* __restrict__ hoff, int* __restrict__ hcell) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (hoff[i + 1] != hoff[i]) hcell[hoff[i]] = (int)i;
}

// position of every member's cell in the (ascending) halo list
__global__ __launch_bounds__(256) void enrich_vhalo_kernel(int nmem, const int* __restrict__ vcell, const int* __restrict__ hcell,
                                                           int nhalo, int* __restrict__ vhalo, int* __restrict__ bad) {
  for (int m = blockIdx.x * 256 + threadIdx.x; m < nmem; m += gridDim.x * 256) {
    const int cell = vcell[m];
    int lo = 0, hi = nhalo - 1, pos = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const int hc = hcell[mid];
      if (hc == cell) {
        pos = mid;
        break;
      }
      if (hc < cell) lo = mid + 1; else hi = mid - 1;
    }
    if (pos < 0) {
      atomicOr(bad, 1);
      pos = 0;
    }
    vhalo[m] = pos;
  }
}

__global__ __launch_bounds__(256) void enrich_vec_of_tile_kernel(int ntiles, const int* __restrict__ flag,
                                                                 const int* __restrict__ voff, int* __restrict__ vec_of_tile,
                                                                 int* __restrict__ tile_of_vec) {
  for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) {
    vec_of_tile[t] = flag[t] ? voff[t] : -1;
    if (flag[t]) tile_of_vec[voff[t]] = t;
  }
}

// ---- set-up ---------------------------------------------------------------------------------------------------------
// Knobs (csgpu_opts.enrich = -1 switches the enrichment off, .enrich_tau, .enrich_steps; defaults 0.06 and 6).
inline double enrich_tau() { return knobs().enrich ? knobs().enrich_tau : 0.0; }

template <class U, class T>
inline void enrich_setup(Enrich& E, const U* rows, int R, int C, int Rc, int Cc, const long long* size0, const int* agg,
                         const unsigned long long* size_c, hipStream_t st) {
  E = Enrich();
  const double tau = enrich_tau();
  if (!(tau > 0.0)) return;
  const int psteps = knobs().enrich_steps;
  const int64_t n = (int64_t)R * C;
  const int ntiles = Rc * Cc;
  DBuf phi((size_t)n * sizeof(T)), flag = dalloc<int>((size_t)ntiles + 1), mcount = dalloc<int>((size_t)ntiles + 1);
  CS_HIP(hipMemsetAsync(phi.p, 0, phi.bytes, st));
  CS_HIP(hipMemsetAsync(flag.p, 0, flag.bytes, st));
  CS_HIP(hipMemsetAsync(mcount.p, 0, mcount.bytes, st));
  const int gt = std::min(ceil_div(ntiles, 64), 65536);
  hipLaunchKernelGGL((enrich_phi_kernel<U, T>), dim3(gt), dim3(64), 0, st, R, C, Rc, Cc, rows, size0, agg, size_c, tau, psteps,
                     dptr<T>(phi), dptr<int>(flag), dptr<int>(mcount));
  check_launch("enrichment vectors");
  DBuf voff = dalloc<int>((size_t)ntiles + 1), tot = dalloc<int>(4);
  CS_HIP(hipMemcpyAsync(voff.p, flag.p, ((size_t)ntiles + 1) * sizeof(int), hipMemcpyDeviceToDevice, st));
  exclusive_scan_i32(dptr<int>(voff), (int64_t)ntiles + 1, st, dptr<int>(tot));
  const int nvec = read_int(dptr<int>(tot), st);
  if (nvec <= 0) return;
  exclusive_scan_i32(dptr<int>(mcount), (int64_t)ntiles + 1, st, dptr<int>(tot) + 1);
  const int nmem = read_int(dptr<int>(tot) + 1, st);
  DBuf vec_of_tile = dalloc<int>((size_t)ntiles), tile_of_vec = dalloc<int>((size_t)nvec);
  hipLaunchKernelGGL(enrich_vec_of_tile_kernel, dim3(grid_for(ntiles)), dim3(256), 0, st, ntiles, (const int*)dptr<int>(flag),
                     (const int*)dptr<int>(voff), dptr<int>(vec_of_tile), dptr<int>(tile_of_vec));
  const int gv = std::min(ceil_div(nvec, 64), 65536);
  E.vptr = dalloc<int>((size_t)nvec + 1);
  E.vcell = dalloc<int>((size_t)std::max(nmem, 1));
  E.vphi.alloc((size_t)std::max(nmem, 1) * sizeof(T));
  E.binv = dalloc<double>((size_t)nvec);
  E.aptr = dalloc<int>((size_t)nvec + 1);
  CS_HIP(hipMemsetAsync(E.aptr.p, 0, E.aptr.bytes, st));
  hipLaunchKernelGGL((enrich_lists_kernel<U, T>), dim3(gv), dim3(64), 0, st, R, C, Rc, Cc, rows, agg, (const T*)dptr<T>(phi),
                     (const int*)dptr<int>(tile_of_vec), nvec, (const int*)dptr<int>(mcount), dptr<int>(E.vptr), dptr<int>(E.vcell),
                     dptr<T>(E.vphi), dptr<double>(E.binv), dptr<int>(E.aptr));
  CS_HIP(hipMemcpyAsync(dptr<int>(E.vptr) + nvec, &nmem, sizeof(int), hipMemcpyHostToDevice, st));
  exclusive_scan_i32(dptr<int>(E.aptr), (int64_t)nvec + 1, st, dptr<int>(tot) + 2);
  const int nae = read_int(dptr<int>(tot) + 2, st);
  E.acell = dalloc<int>((size_t)std::max(nae, 1));
  E.acoef = dalloc<double>((size_t)std::max(nae, 1));
  hipLaunchKernelGGL((enrich_ae_fill_kernel<U, T>), dim3(gv), dim3(64), 0, st, R, C, Rc, Cc, rows, agg, (const T*)dptr<T>(phi),
                     (const int*)dptr<int>(tile_of_vec), nvec, (const int*)dptr<int>(E.aptr), dptr<int>(E.acell), dptr<double>(E.acoef));
  // halo list: the cells whose residual the pre-correction changes, and their rows of A E
  DBuf hoff = dalloc<int>((size_t)n + 1), ovf = dalloc<int>(2);
  CS_HIP(hipMemsetAsync(ovf.p, 0, ovf.bytes, st));
  hipLaunchKernelGGL((enrich_halo_kernel<U, T, 0>), dim3(grid_for(n + 1)), dim3(256), 0, st, n, R, rows, agg, (const T*)dptr<T>(phi),
                     (const int*)dptr<int>(vec_of_tile), dptr<int>(hoff), (const int*)nullptr, 0, (int*)nullptr, (const int*)nullptr,
                     (int*)nullptr, (double*)nullptr, dptr<int>(ovf));
  exclusive_scan_i32(dptr<int>(hoff), n + 1, st, dptr<int>(tot));
  const int nhalo = read_int(dptr<int>(tot), st);
  if (nhalo <= 0 || read_int(dptr<int>(ovf), st) != 0) {  // (a row of A E with more than kEnrichRowMax vectors: no enrichment)
    E = Enrich();
    return;
  }
  E.hcell = dalloc<int>((size_t)nhalo);
  hipLaunchKernelGGL(enrich_halo_fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, (const int*)dptr<int>(hoff), dptr<int>(E.hcell));
  hoff.release();
  E.hptr = dalloc<int>((size_t)nhalo + 1);
  CS_HIP(hipMemsetAsync(E.hptr.p, 0, E.hptr.bytes, st));
  hipLaunchKernelGGL((enrich_halo_kernel<U, T, 1>), dim3(grid_for(nhalo)), dim3(256), 0, st, n, R, rows, agg, (const T*)dptr<T>(phi),
                     (const int*)dptr<int>(vec_of_tile), (int*)nullptr, (const int*)dptr<int>(E.hcell), nhalo, dptr<int>(E.hptr),
                     (const int*)nullptr, (int*)nullptr, (double*)nullptr, dptr<int>(ovf));
  exclusive_scan_i32(dptr<int>(E.hptr), (int64_t)nhalo + 1, st, dptr<int>(tot) + 3);
  const int nhe = read_int(dptr<int>(tot) + 3, st);
  E.hvec = dalloc<int>((size_t)std::max(nhe, 1));
  E.hcoef = dalloc<double>((size_t)std::max(nhe, 1));
  hipLaunchKernelGGL((enrich_halo_kernel<U, T, 2>), dim3(grid_for(nhalo)), dim3(256), 0, st, n, R, rows, agg, (const T*)dptr<T>(phi),
                     (const int*)dptr<int>(vec_of_tile), (int*)nullptr, (const int*)dptr<int>(E.hcell), nhalo, (int*)nullptr,
                     (const int*)dptr<int>(E.hptr), dptr<int>(E.hvec), dptr<double>(E.hcoef), dptr<int>(ovf));
  E.vhalo = dalloc<int>((size_t)std::max(nmem, 1));
  hipLaunchKernelGGL(enrich_vhalo_kernel, dim3(grid_for(nmem)), dim3(256), 0, st, nmem, (const int*)dptr<int>(E.vcell),
                     (const int*)dptr<int>(E.hcell), nhalo, dptr<int>(E.vhalo), dptr<int>(ovf) + 1);
  check_launch("enrichment lists");
  if (read_int(dptr<int>(ovf) + 1, st) != 0) {  // (cannot happen: every member with phi != 0 has a non-empty row of A E ...
    E = Enrich();                               //  unless its row cancels exactly; then the correction is simply not used)
    return;
  }
  E.phi_bytes = (int)sizeof(T);
  E.nvec = nvec;
  E.nmem = nmem;
  E.nhalo = nhalo;
  E.R = R;
  E.n = n;
  E.ntiles = ntiles;
  if (knobs().verbose)
    fprintf(stderr, "csgpu: coarse-space enrichment: %d of %d aggregates get a second function (tau %.3g), %d members, %d halo cells, %d + %d entries of A E\n",
            nvec, ntiles, tau, nmem, nhalo, nae, nhe);
}

// ---- application (K interleaved columns) ------------------------------------------------------------------------------
// t[v][c] = sum_members phi_i r_i ; cc[v][c] = binv_v t   (one thread per (vector, column))
template <class T, int K>
__global__ __launch_bounds__(256) void enrich_gather_kernel(int nvec, const int* __restrict__ vptr, const int* __restrict__ vcell,
                                                            const T* __restrict__ vphi, const double* __restrict__ binv,
                                                            const T* __restrict__ r, double* __restrict__ t,
                                                            double* __restrict__ cc, const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < (int64_t)nvec * K; id += (int64_t)gridDim.x * 256) {
    const int v = (int)(id / K), c = (int)(id % K);
    double s = 0.0;
    for (int m = vptr[v]; m < vptr[v + 1]; ++m) s += (double)vphi[m] * (double)r[(int64_t)vcell[m] * K + c];
    t[id] = s;
    cc[id] = binv[v] * s;
  }
}

// halo cells: s = (A E c)_i kept for the post pass, r saved (SAVE: when r is the CG residual itself) and r -= s
template <class T, int K, bool SAVE>
__global__ __launch_bounds__(256) void enrich_pre_kernel(int nhalo, const int* __restrict__ hcell, const int* __restrict__ hptr,
                                                         const int* __restrict__ hvec, const double* __restrict__ hcoef,
                                                         const double* __restrict__ cc, T* __restrict__ r,
                                                         double* __restrict__ sbuf, T* __restrict__ save,
                                                         const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < (int64_t)nhalo * K; id += (int64_t)gridDim.x * 256) {
    const int h = (int)(id / K), c = (int)(id % K);
    double s = 0.0;
    for (int e = hptr[h]; e < hptr[h + 1]; ++e) s += hcoef[e] * cc[(int64_t)hvec[e] * K + c];
    sbuf[id] = s;
    const int64_t at = (int64_t)hcell[h] * K + c;
    const T old = r[at];
    if (SAVE) save[id] = old;
    r[at] = (T)((double)old - s);
  }
}

// c2 = binv (t - E'A(E c) - E'A z) ; cnew = c + c2 ; corr[v][c] = c s_z + t (c + c2) -> block partial rows of the r'z correction
template <class T, int K>
__global__ __launch_bounds__(256) void enrich_post_kernel(int nvec, const int* __restrict__ vptr, const T* __restrict__ vphi,
                                                          const int* __restrict__ vhalo, const int* __restrict__ aptr,
                                                          const int* __restrict__ acell, const double* __restrict__ acoef,
                                                          const double* __restrict__ binv, const T* __restrict__ z,
                                                          const double* __restrict__ sbuf, const double* __restrict__ t,
                                                          const double* __restrict__ cc, double* __restrict__ cnew,
                                                          double* __restrict__ part, const int* __restrict__ skip) {
  __shared__ double sm[256];
  if (skip && *skip) return;
  double acc = 0.0;   // this thread's column is (threadIdx.x % K) for every id it visits (256 % K == 0, grid stride % K == 0)
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < (int64_t)nvec * K; id += (int64_t)gridDim.x * 256) {
    const int v = (int)(id / K), c = (int)(id % K);
    double sz = 0.0, sy = 0.0;
    for (int e = aptr[v]; e < aptr[v + 1]; ++e) sz += acoef[e] * (double)z[(int64_t)acell[e] * K + c];
    for (int m = vptr[v]; m < vptr[v + 1]; ++m) sy += (double)vphi[m] * sbuf[(int64_t)vhalo[m] * K + c];
    const double c1 = cc[id], tv = t[id];
    const double c2 = binv[v] * (tv - sy - sz);
    cnew[id] = c1 + c2;
    acc += c1 * sz + tv * (c1 + c2);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < K) {
    double s = 0.0;
    for (int q = threadIdx.x; q < 256; q += K) s += sm[q];
    part[(int64_t)blockIdx.x * K + threadIdx.x] = s;
  }
}

// z += E cnew at the members ; r restored at the halo cells (SAVE)
template <class T, int K, bool SAVE>
__global__ __launch_bounds__(256) void enrich_finish_kernel(int nvec, const int* __restrict__ vptr, const int* __restrict__ vcell,
                                                            const T* __restrict__ vphi, const double* __restrict__ cnew,
                                                            T* __restrict__ z, int nhalo, const int* __restrict__ hcell,
                                                            const T* __restrict__ save, T* __restrict__ r,
                                                            const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int64_t nv = (int64_t)nvec * K, nh = SAVE ? (int64_t)nhalo * K : 0;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < nv + nh; id += (int64_t)gridDim.x * 256) {
    if (id < nv) {
      const int v = (int)(id / K), c = (int)(id % K);
      const double x = cnew[id];
      for (int m = vptr[v]; m < vptr[v + 1]; ++m) {
        const int64_t at = (int64_t)vcell[m] * K + c;
        z[at] = (T)((double)z[at] + (double)vphi[m] * x);
      }
    } else {
      const int64_t q = id - nv;
      r[(int64_t)hcell[q / K] * K + (q % K)] = save[q];
    }
  }
}

// block partials of the post pass -> kEnrichParts rows (the chunked sum of collapse_partials_kernel, blas1.h)
template <int K>
__global__ __launch_bounds__(256) void enrich_collapse_kernel(const double* __restrict__ in, int nparts, double* __restrict__ out,
                                                              const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 256 * K) return;
  const int r = t / K, c = t % K;
  const int chunk = (nparts + 255) / 256;
  const int lo = r * chunk, hi = min(nparts, lo + chunk);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) s += in[(size_t)i * K + c];
  out[(size_t)r * K + c] = s;
}

static const int kEnrichParts = 256;  // rows of r'z correction partials (appended to the V-cycle's own rows; pcg.h: kEnrichRows)

template <class T, int K>
inline void enrich_ensure_work(Enrich& E) {
  if (E.work_k == K && E.work_bytes == (int)sizeof(T)) return;
  E.t = dalloc<double>((size_t)E.nvec * K);
  E.c = dalloc<double>((size_t)E.nvec * K);
  E.c2 = dalloc<double>((size_t)E.nvec * K);
  E.sbuf = dalloc<double>((size_t)E.nhalo * K);
  E.ppart = dalloc<double>((size_t)kMaxGrid * K);
  E.save.alloc((size_t)std::max(E.nhalo, 1) * K * sizeof(T));
  E.work_k = K;
  E.work_bytes = (int)sizeof(T);
}

// before the V-cycle: r = the V-cycle's input, modified in place at the halo cells. SAVE: r is the CG residual itself (the
// hierarchy computes in the CG precision) and enrich_post puts the saved entries back, bit for bit; otherwise r is the
// preconditioner-precision copy the next residual update rewrites anyway.
template <class T, int K, bool SAVE>
inline void enrich_pre(Enrich& E, T* r, const int* skip, hipStream_t st) {
  enrich_ensure_work<T, K>(E);
  hipLaunchKernelGGL((enrich_gather_kernel<T, K>), dim3(grid_for((int64_t)E.nvec * K)), dim3(256), 0, st, E.nvec,
                     (const int*)dptr<int>(E.vptr), (const int*)dptr<int>(E.vcell), (const T*)dptr<T>(E.vphi),
                     (const double*)dptr<double>(E.binv), (const T*)r, dptr<double>(E.t), dptr<double>(E.c), skip);
  hipLaunchKernelGGL((enrich_pre_kernel<T, K, SAVE>), dim3(grid_for((int64_t)E.nhalo * K)), dim3(256), 0, st, E.nhalo,
                     (const int*)dptr<int>(E.hcell), (const int*)dptr<int>(E.hptr), (const int*)dptr<int>(E.hvec),
                     (const double*)dptr<double>(E.hcoef), (const double*)dptr<double>(E.c), r, dptr<double>(E.sbuf),
                     dptr<T>(E.save), skip);
}

// after the V-cycle: z corrected, r restored, kEnrichParts rows of r'z correction partials written to `part`
template <class T, int K, bool SAVE>
inline void enrich_post(Enrich& E, T* r, T* z, double* part, const int* skip, hipStream_t st) {
  // one workgroup per 256 (vector, column) items up to kMaxGrid: the block partials go to a scratch array and are collapsed
  // into the kEnrichParts rows behind the V-cycle's own (a grid of kEnrichParts workgroups ran 5 ms per pass at 10000^2)
  const int g = grid_for((int64_t)E.nvec * K);
  hipLaunchKernelGGL((enrich_post_kernel<T, K>), dim3(g), dim3(256), 0, st, E.nvec, (const int*)dptr<int>(E.vptr),
                     (const T*)dptr<T>(E.vphi), (const int*)dptr<int>(E.vhalo), (const int*)dptr<int>(E.aptr),
                     (const int*)dptr<int>(E.acell), (const double*)dptr<double>(E.acoef), (const double*)dptr<double>(E.binv),
                     (const T*)z, (const double*)dptr<double>(E.sbuf), (const double*)dptr<double>(E.t),
                     (const double*)dptr<double>(E.c), dptr<double>(E.c2), dptr<double>(E.ppart), skip);
  hipLaunchKernelGGL((enrich_collapse_kernel<K>), dim3(ceil_div(kEnrichParts * K, 256)), dim3(256), 0, st,
                     (const double*)dptr<double>(E.ppart), g, part, skip);
  hipLaunchKernelGGL((enrich_finish_kernel<T, K, SAVE>), dim3(grid_for((int64_t)(E.nvec + (SAVE ? E.nhalo : 0)) * K)), dim3(256), 0,
                     st, E.nvec, (const int*)dptr<int>(E.vptr), (const int*)dptr<int>(E.vcell), (const T*)dptr<T>(E.vphi),
                     (const double*)dptr<double>(E.c2), z, E.nhalo, (const int*)dptr<int>(E.hcell), (const T*)dptr<T>(E.save), r,
                     skip);
}

// ---- the restriction's share of the pre-pass (fused residual update + restriction) ------------------------------------------
// The fused pass of lattice.h leaves b_c = Q^T r; the V-cycle of an enriched level wants Q^T (r - s) with s = A E c on the halo
// cells. Q^T s = (Q^T A E) c =: W c, and W is tiny: a coarse node meets the vectors of the 5 x 5 block of tiles around it at
// most (2.2 M entries at 10000^2 / 15 % NODATA against 25 M entries of Q^T on the halo cells -- the first version gathered
// sbuf through those and cost 1.3 ms per K = 32 iteration, more than any pass of the enrichment itself; c is 46 MB and stays
// in the last-level cache). Built once per set-up: Q^T on the halo cells by coarse rows (counts and slots by atomics, every
// row then sorted by halo position), multiplied with the rows of A E in that order -- fixed summation order, deterministic.
// One thread per (coarse node, column) applies it.
__device__ __forceinline__ int enr_tile(int i, int nc) {  // (lat_tile of lattice.h)
  const int t = i / 3;
  return t < nc ? t : nc - 1;
}

// PASS 0: entries per coarse node; PASS 1: (halo position, weight) at cptr[a] + a running slot
template <class T, int PASS>
__global__ __launch_bounds__(256) void enrich_coarse_lists_kernel(int nhalo, const int* __restrict__ hcell, int R, int Rc, int Cc,
                                                                  const T* __restrict__ q, int* __restrict__ cnt,
                                                                  const int* __restrict__ cptr, int* __restrict__ th,
                                                                  double* __restrict__ tw) {
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < (int64_t)nhalo * 9; id += (int64_t)gridDim.x * 256) {
    const int h = (int)(id / 9), s = (int)(id % 9);
    const int64_t cell = hcell[h];
    const T w = q[cell * 9 + s];
    if (w == T(0)) continue;
    const int Ic = enr_tile((int)(cell % R), Rc) + s % 3 - 1, Jc = enr_tile((int)(cell / R), Cc) + s / 3 - 1;
    if (Ic < 0 || Ic >= Rc || Jc < 0 || Jc >= Cc) continue;
    const int a = Jc * Rc + Ic;
    const int k = atomicAdd(&cnt[a], 1);
    if (PASS == 1) {
      th[cptr[a] + k] = h;
      tw[cptr[a] + k] = (double)w;
    }
  }
}

__global__ __launch_bounds__(256) void enrich_coarse_flag_kernel(int nc, const int* __restrict__ cnt, int* __restrict__ flag) {
  for (int a = blockIdx.x * 256 + threadIdx.x; a <= nc; a += gridDim.x * 256) flag[a] = (a < nc && cnt[a] > 0) ? 1 : 0;
}

// compact list of the touched coarse nodes; each sorts its entries by halo position (insertion sort: rows of <= 81 entries)
__global__ __launch_bounds__(256) void enrich_coarse_rows_kernel(int nc, const int* __restrict__ cptr, const int* __restrict__ tpos,
                                                                 int* __restrict__ tcell, int* __restrict__ tptr,
                                                                 int* __restrict__ th, double* __restrict__ tw) {
  for (int a = blockIdx.x * 256 + threadIdx.x; a < nc; a += gridDim.x * 256) {
    const int b = cptr[a], e = cptr[a + 1];
    if (e <= b) continue;
    const int u = tpos[a];
    tcell[u] = a;
    tptr[u] = b;
    for (int i = b + 1; i < e; ++i) {
      const int hk = th[i];
      const double wk = tw[i];
      int j = i - 1;
      while (j >= b && th[j] > hk) {
        th[j + 1] = th[j];
        tw[j + 1] = tw[j];
        --j;
      }
      th[j + 1] = hk;
      tw[j + 1] = wk;
    }
  }
}

// row u of W = Q^T A E from row u of Q^T (entries sorted by halo position) and the rows of A E; vectors in the order they
// first appear. PASS 0: number of distinct vectors (ovf when more than kEnrichWMax); PASS 1: (vector, value) from wptr[u]
static const int kEnrichWMax = 40;
template <int PASS>
__global__ __launch_bounds__(64) void enrich_coarse_w_kernel(int ntouch, const int* __restrict__ tptr, const int* __restrict__ th,
                                                            const double* __restrict__ tw, const int* __restrict__ hptr,
                                                            const int* __restrict__ hvec, const double* __restrict__ hcoef,
                                                            int* __restrict__ wcnt, const int* __restrict__ wptr,
                                                            int* __restrict__ wv, double* __restrict__ ww, int* __restrict__ ovf) {
  for (int u = blockIdx.x * 64 + threadIdx.x; u < ntouch; u += gridDim.x * 64) {
    int vid[kEnrichWMax];
    double val[kEnrichWMax];
    int m = 0;
    bool over = false;
    for (int e = tptr[u]; e < tptr[u + 1]; ++e) {
      const int h = th[e];
      const double w = tw[e];
      for (int f = hptr[h]; f < hptr[h + 1]; ++f) {
        const int v = hvec[f];
        int k = 0;
        while (k < m && vid[k] != v) ++k;
        if (k == m) {
          if (m == kEnrichWMax) {
            over = true;
            continue;
          }
          vid[m] = v;
          val[m] = 0.0;
          ++m;
        }
        val[k] += w * hcoef[f];
      }
    }
    if (over) atomicOr(ovf, 1);
    if (PASS == 0) {
      wcnt[u] = m;
    } else {
      const int b = wptr[u];
      for (int k = 0; k < m; ++k) {
        wv[b + k] = vid[k];
        ww[b + k] = val[k];
      }
    }
  }
}

template <class T>
inline void enrich_coarse_setup(Enrich& E, const T* q, int R, int Rc, int Cc, hipStream_t st) {
  E.ntouch = 0;
  if (E.nvec <= 0 || E.nhalo <= 0) return;
  const int nc = Rc * Cc;
  DBuf cnt = dalloc<int>((size_t)nc + 1), flag = dalloc<int>((size_t)nc + 1), tot = dalloc<int>(2);
  CS_HIP(hipMemsetAsync(cnt.p, 0, cnt.bytes, st));
  const int g = grid_for((int64_t)E.nhalo * 9);
  hipLaunchKernelGGL((enrich_coarse_lists_kernel<T, 0>), dim3(g), dim3(256), 0, st, E.nhalo, (const int*)dptr<int>(E.hcell), R, Rc, Cc,
                     q, dptr<int>(cnt), (const int*)nullptr, (int*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(enrich_coarse_flag_kernel, dim3(grid_for((int64_t)nc + 1)), dim3(256), 0, st, nc, (const int*)dptr<int>(cnt),
                     dptr<int>(flag));
  exclusive_scan_i32(dptr<int>(flag), (int64_t)nc + 1, st, dptr<int>(tot));      // flag -> position in the compact list
  exclusive_scan_i32(dptr<int>(cnt), (int64_t)nc + 1, st, dptr<int>(tot) + 1);   // cnt -> first entry
  const int ntouch = read_int(dptr<int>(tot), st), nent = read_int(dptr<int>(tot) + 1, st);
  if (ntouch <= 0 || nent <= 0) return;
  DBuf cursor = dalloc<int>((size_t)nc + 1);
  CS_HIP(hipMemsetAsync(cursor.p, 0, cursor.bytes, st));
  E.tcell = dalloc<int>((size_t)ntouch);
  DBuf qptr = dalloc<int>((size_t)ntouch + 1), qh = dalloc<int>((size_t)nent), qw = dalloc<double>((size_t)nent);   // Q^T on the halo cells
  hipLaunchKernelGGL((enrich_coarse_lists_kernel<T, 1>), dim3(g), dim3(256), 0, st, E.nhalo, (const int*)dptr<int>(E.hcell), R, Rc, Cc,
                     q, dptr<int>(cursor), (const int*)dptr<int>(cnt), dptr<int>(qh), dptr<double>(qw));
  hipLaunchKernelGGL(enrich_coarse_rows_kernel, dim3(grid_for(nc)), dim3(256), 0, st, nc, (const int*)dptr<int>(cnt),
                     (const int*)dptr<int>(flag), dptr<int>(E.tcell), dptr<int>(qptr), dptr<int>(qh), dptr<double>(qw));
  CS_HIP(hipMemcpyAsync(dptr<int>(qptr) + ntouch, &nent, sizeof(int), hipMemcpyHostToDevice, st));
  // W = Q^T (A E), row by row
  E.tptr = dalloc<int>((size_t)ntouch + 1);
  CS_HIP(hipMemsetAsync(E.tptr.p, 0, E.tptr.bytes, st));
  CS_HIP(hipMemsetAsync(tot.p, 0, tot.bytes, st));
  const int gw = std::min(ceil_div(ntouch, 64), 65536);
  hipLaunchKernelGGL((enrich_coarse_w_kernel<0>), dim3(gw), dim3(64), 0, st, ntouch, (const int*)dptr<int>(qptr), (const int*)dptr<int>(qh),
                     (const double*)dptr<double>(qw), (const int*)dptr<int>(E.hptr), (const int*)dptr<int>(E.hvec),
                     (const double*)dptr<double>(E.hcoef), dptr<int>(E.tptr), (const int*)nullptr, (int*)nullptr, (double*)nullptr,
                     dptr<int>(tot) + 1);
  exclusive_scan_i32(dptr<int>(E.tptr), (int64_t)ntouch + 1, st, dptr<int>(tot));
  const int nw = read_int(dptr<int>(tot), st);
  if (nw <= 0 || read_int(dptr<int>(tot) + 1, st) != 0) {  // (a coarse node that meets more than kEnrichWMax vectors: two passes)
    E.tcell.release();
    E.tptr.release();
    return;
  }
  E.th = dalloc<int>((size_t)nw);
  E.tw = dalloc<double>((size_t)nw);
  hipLaunchKernelGGL((enrich_coarse_w_kernel<1>), dim3(gw), dim3(64), 0, st, ntouch, (const int*)dptr<int>(qptr), (const int*)dptr<int>(qh),
                     (const double*)dptr<double>(qw), (const int*)dptr<int>(E.hptr), (const int*)dptr<int>(E.hvec),
                     (const double*)dptr<double>(E.hcoef), (int*)nullptr, (const int*)dptr<int>(E.tptr), dptr<int>(E.th),
                     dptr<double>(E.tw), dptr<int>(tot) + 1);
  check_launch("enrichment coarse lists");
  CS_HIP(hipStreamSynchronize(st));  // (nent is a stack variable; the temporaries go back to the pool)
  E.ntouch = ntouch;
  if (knobs().verbose)
    fprintf(stderr, "csgpu: coarse-space enrichment: %d coarse nodes take the restriction's share of the pre-pass (%d entries of W = Q'AE from %d of Q')\n",
            ntouch, nw, nent);
}

// bc -= W c on the touched coarse nodes (c = cc of the gather pass that has just run)
template <class T, int K>
__global__ __launch_bounds__(256) void enrich_coarse_fix_kernel(int ntouch, const int* __restrict__ tcell, const int* __restrict__ tptr,
                                                                const int* __restrict__ th, const double* __restrict__ tw,
                                                                const double* __restrict__ sbuf, T* __restrict__ bc,
                                                                const int* __restrict__ skip) {
  if (skip && *skip) return;
  // (a vector's row of c is read by the coarse nodes of up to five coarse columns: every XCD -- blockIdx % 8 -- walks ONE
  // contiguous eighth of the ascending list, so that the re-reads find the row in that XCD's L2)
  const int64_t items = (int64_t)ntouch * K;
  int64_t lo = 0, hi = items, first = blockIdx.x, nblk = gridDim.x;
  if (gridDim.x >= 8) {
    const int64_t per = ((items + 8 * 256 - 1) / (8 * 256)) * 256;
    const int xcd = blockIdx.x % 8;
    lo = (int64_t)xcd * per;
    hi = lo + per < items ? lo + per : items;
    first = blockIdx.x / 8;
    nblk = (gridDim.x + 7 - xcd) / 8;
  }
  for (int64_t id = lo + first * 256 + threadIdx.x; id < hi; id += nblk * 256) {
    const int u = (int)(id / K), c = (int)(id % K);
    double s = 0.0;
    for (int e = tptr[u]; e < tptr[u + 1]; ++e) s += tw[e] * sbuf[(int64_t)th[e] * K + c];
    const int64_t at = (int64_t)tcell[u] * K + c;
    bc[at] = (T)((double)bc[at] - s);
  }
}

template <class T, int K>
inline void enrich_coarse_fix(Enrich& E, T* bc, const int* skip, hipStream_t st) {
  hipLaunchKernelGGL((enrich_coarse_fix_kernel<T, K>), dim3(grid_for((int64_t)E.ntouch * K)), dim3(256), 0, st, E.ntouch,
                     (const int*)dptr<int>(E.tcell), (const int*)dptr<int>(E.tptr), (const int*)dptr<int>(E.th),
                     (const double*)dptr<double>(E.tw), (const double*)dptr<double>(E.c), bc, skip);
}

}  // namespace csgpu
