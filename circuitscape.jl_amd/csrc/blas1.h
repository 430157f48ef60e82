// blas1.h -- K2: fused vector kernels of the PCG loop, with device-resident scalars (no host sync per iteration).
//
// GPU counterpart of the vector algebra inside Krylov.cg (reference call site src/core.jl:639):
//   alpha = gamma / p'Ap ; x += alpha p ; r -= alpha Ap ; gamma' = r'z ; beta = gamma'/gamma ; p = z + beta p
// and of its stopping rule  sqrt(r'z) <= atol + rtol*sqrt(r0'z0)   (or the true-residual variant).
//
// All vectors are interleaved [n][K] (see spmv.h). Reductions are two-stage and atomic-free: every workgroup
// writes one partial per column (wave64 shuffle -> LDS), a single-workgroup "scalar" kernel sums the partials in
// a fixed order and updates the per-column CG scalars in device memory. Results are bit-reproducible run to run.
#pragma once
#include "prims.h"

namespace csgpu {

// Per-batch CG scalars living in device memory (one struct per handle, arrays indexed by column c < K).
static const int kMaxK = 32;  // widest batch (32 since round 4: the matrix values of a marching pass are amortised over twice as many columns)
struct CgScalars {
  double gamma[kMaxK];   // r'z
  double pAp[kMaxK];
  double alpha[kMaxK];
  double beta[kMaxK];
  double rnorm[kMaxK];   // current value of the monitored norm
  double rnorm0[kMaxK];
  double eps[kMaxK];     // stopping threshold atol + rtol*rnorm0
  double eps2[kMaxK];    // criterion 2 only: threshold atol + rtol*||r0||_2 of the second (true-residual) test
  double bnorm[kMaxK];   // ||b||_2 (for the final relative residual)
  double relres[kMaxK];  // ||A x - b|| / ||b|| from the explicit post-check
  int done[kMaxK];       // 1 converged, 2 breakdown (p'Ap <= 0 or non-finite), 4 itmax reached (streaming solves only)
  int iters[kMaxK];
  int all_done;
  int pad;
  // ---- streaming pair solves (pcg_stream_pairs, pcg.h): every column is a slot that takes the next pair of the call's
  // list as soon as its own pair has converged. `ctl` is written by the host between two iterations (the device is idle
  // then); restart[c] is consumed -- and cleared -- by the kernels of the next iteration.
  struct Ctl {
    int restart[kMaxK];  // 1: column c starts a new pair in the next iteration (r := e_dst - e_src, x := 0, p := z)
    int src[kMaxK];      // row ids of the new pair
    int dst[kMaxK];
    int active[kMaxK];   // 0: the slot is idle (pair list exhausted)
  } ctl;
  int polish[kMaxK];     // the column met the configured rule with ||r|| / ||b|| >= 1e-4 and now runs to 2.5e-5 on the true
                         // residual (what cg_reopen_kernel does for a whole batch)
};

// ---- dot products: partials[block][c] = sum_i a[i,c]*b[i,c]  (second pair optional: a2.b2 -> partials2)
template <class T, int K, bool TWO>
__global__ __launch_bounds__(256) void dot_kernel(int64_t n, const T* __restrict__ a, const T* __restrict__ b,
                                                  double* __restrict__ partials, const T* __restrict__ a2,
                                                  const T* __restrict__ b2, double* __restrict__ partials2) {
  __shared__ double s_red[4 * K];
  __shared__ double s_red2[4 * K];
  // each thread keeps a fixed column: stride over elements is a multiple of K because 256 % K == 0
  double s = 0.0, s2 = 0.0;
  const int64_t total = n * K;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    s += (double)a[e] * (double)b[e];
    if (TWO) s2 += (double)a2[e] * (double)b2[e];
  }
#pragma unroll
  for (int o = 32; o >= K; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    if (TWO) s2 += __shfl_xor(s2, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane < K) {
    s_red[w * K + lane] = s;
    if (TWO) s_red2[w * K + lane] = s2;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int t = threadIdx.x;
    partials[(size_t)blockIdx.x * K + t] = s_red[t] + s_red[K + t] + s_red[2 * K + t] + s_red[3 * K + t];
    if (TWO) partials2[(size_t)blockIdx.x * K + t] = s_red2[t] + s_red2[K + t] + s_red2[2 * K + t] + s_red2[3 * K + t];
  }
}

// Sum `nparts` partial rows for column c in a fixed order (called by one workgroup of 256 threads).
template <int K>
__device__ __forceinline__ double reduce_partials(const double* partials, int nparts, int c, double* sm) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[(size_t)i * K + c];
  return block_sum_256(s, sm);
}

// Collapse `nparts` partial rows to kCollapsedParts rows (row r = sum of a contiguous chunk of input rows, in order:
// deterministic), so that the single-workgroup scalar kernels below never walk more than a few hundred rows however
// many workgroups the producing SpMM launch had. One thread per (output row, column).
static const int kCollapsedParts = 256;
template <int K>
__global__ __launch_bounds__(256) void collapse_partials_kernel(const double* __restrict__ in, int nparts,
                                                                double* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= kCollapsedParts * K) return;
  const int r = t / K, c = t % K;
  const int chunk = (nparts + kCollapsedParts - 1) / kCollapsedParts;
  const int lo = r * chunk, hi = min(nparts, lo + chunk);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) s += in[(size_t)i * K + c];
  out[(size_t)r * K + c] = s;
}

// ---- scalar kernel 1: pAp -> alpha
template <int K>
__global__ __launch_bounds__(256) void cg_alpha_kernel(CgScalars* S, const double* partials, int nparts) {
  __shared__ double sm[4];
  if (S->all_done) return;  // a surplus iteration enqueued before the host saw the flag: every kernel of it is a no-op
  for (int c = 0; c < K; ++c) {
    const double pAp = reduce_partials<K>(partials, nparts, c, sm);
    if (threadIdx.x == 0) {
      S->pAp[c] = pAp;
      double alpha = 0.0;
      if (!S->done[c]) {
        if (pAp > 0.0 && pAp == pAp && S->gamma[c] == S->gamma[c]) {
          alpha = S->gamma[c] / pAp;
        } else {
          S->done[c] = 2;  // breakdown (zero curvature / non-finite), Krylov.jl exits likewise
        }
      }
      S->alpha[c] = alpha;
    }
    __syncthreads();
  }
}

// ---- y = (V) x   (precision conversion between the CG vectors and the preconditioner's vectors)
template <class U, class V>
__global__ __launch_bounds__(256) void convert_kernel(int64_t total, const U* __restrict__ x, V* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) y[e] = (V)x[e];
}

// ---- r -= alpha Ap, fused with: the preconditioner-precision copy of r (rp, when TP != T), the first damped-Jacobi
//      sweep of the next preconditioner application from a zero guess (xa = omega * dinv .* r; skipped when xa is
//      null) and, optionally, the partials of r'r.
// N adjacent vector elements moved with one memory instruction (16 bytes for the T-typed vectors)
template <class V, int N>
struct alignas(sizeof(V) * N) VecN {
  V e[N];
};

// The two streaming kernels below move 28 bytes per vector element and nothing else: every lane handles VEC adjacent
// columns of a node (one 16-byte access to the T vectors, one 8/16-byte access to the TP vectors) and two such groups
// per loop trip (independent loads in flight), grid-stride over the node-column pairs.
template <class T, int K>
struct CgVec {
  static constexpr int RAW = 16 / (int)sizeof(T);
  static constexpr int VEC = K < RAW ? K : RAW;  // columns per lane
  static constexpr int LPR = K / VEC;            // lanes covering one node's K columns
};

// XUP: also x += alpha p in the same pass (solves whose caller needs the whole solution vector; resistance-only pair
// solves accumulate x at their focal nodes only, cg_focal_x_kernel).
template <class T, class TP, int K, bool RR, bool XUP>
__global__ __launch_bounds__(256) void cg_update_r_kernel(int64_t n, const CgScalars* S, T* __restrict__ r,
                                                          const T* __restrict__ Ap, TP* __restrict__ rp,
                                                          TP* __restrict__ xa, const TP* __restrict__ dinv, TP omega,
                                                          double* __restrict__ partials, T* __restrict__ x,
                                                          const TP* __restrict__ p) {
  constexpr int VEC = CgVec<T, K>::VEC, LPR = CgVec<T, K>::LPR;
  typedef VecN<T, VEC> VT;
  typedef VecN<TP, VEC> VP;
  __shared__ double s_red[4 * K];
  if (S->all_done) return;
  const int c0 = (threadIdx.x % LPR) * VEC;
  T alpha[VEC];
  double s[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    alpha[q] = (T)S->alpha[c0 + q];
    s[q] = 0.0;
  }
  const int64_t nv = n * K / VEC;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto body = [&](int64_t v, const VT& rv, const VT& av) {
    VT rn;
    VP rq, xq;
    const TP wd = xa ? omega * dinv[v / LPR] : TP(0);
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      rn.e[q] = rv.e[q] - alpha[q] * av.e[q];
      rq.e[q] = (TP)rn.e[q];
      xq.e[q] = wd * (TP)rn.e[q];
      if (RR) s[q] += (double)rn.e[q] * (double)rn.e[q];
    }
    reinterpret_cast<VT*>(r)[v] = rn;
    if (rp) reinterpret_cast<VP*>(rp)[v] = rq;
    if (xa) reinterpret_cast<VP*>(xa)[v] = xq;
  };
  auto xbody = [&](int64_t v, const VT& xv, const VP& pv) {
    VT xn;
#pragma unroll
    for (int q = 0; q < VEC; ++q) xn.e[q] = fma(alpha[q], (T)pv.e[q], xv.e[q]);
    reinterpret_cast<VT*>(x)[v] = xn;
  };
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += 2 * stride) {
    const int64_t v2 = v + stride;
    const bool two = v2 < nv;
    const VT r1 = reinterpret_cast<const VT*>(r)[v], a1 = reinterpret_cast<const VT*>(Ap)[v];
    VT r2 = r1, a2 = a1;
    VT x1, x2;
    VP p1, p2;
    if (XUP) {
      x1 = reinterpret_cast<const VT*>(x)[v];
      p1 = reinterpret_cast<const VP*>(p)[v];
    }
    if (two) {
      r2 = reinterpret_cast<const VT*>(r)[v2];
      a2 = reinterpret_cast<const VT*>(Ap)[v2];
      if (XUP) {
        x2 = reinterpret_cast<const VT*>(x)[v2];
        p2 = reinterpret_cast<const VP*>(p)[v2];
      }
    }
    body(v, r1, a1);
    if (XUP) xbody(v, x1, p1);
    if (two) {
      body(v2, r2, a2);
      if (XUP) xbody(v2, x2, p2);
    }
  }
  if (RR) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      double t = s[q];
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane < LPR) s_red[w * K + lane * VEC + q] = t;
    }
    __syncthreads();
    if (threadIdx.x < K) {
      const int t = threadIdx.x;
      partials[(size_t)blockIdx.x * K + t] = s_red[t] + s_red[K + t] + s_red[2 * K + t] + s_red[3 * K + t];
    }
  }
}

// ---- p = z + beta p   (beta of the iteration that just finished; the first iteration runs with beta = 0 on a zeroed
//      p). p and z live in the preconditioner's precision TP; the combination is evaluated in T. pin / pout may be the
//      same buffer (in-place update, CSR path) or the two halves of a ping-pong pair.
template <class T, class TP, int K>
__global__ __launch_bounds__(256) void cg_update_p_kernel(int64_t n, const CgScalars* S, const TP* pin,
                                                          TP* pout, const TP* __restrict__ z) {
  constexpr int VEC = CgVec<T, K>::VEC, LPR = CgVec<T, K>::LPR;
  typedef VecN<TP, VEC> VP;
  if (S->all_done) return;
  const int c0 = (threadIdx.x % LPR) * VEC;
  T beta[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) beta[q] = (T)S->beta[c0 + q];
  const int64_t nv = n * K / VEC;
  const int64_t stride = (int64_t)gridDim.x * 256;
  auto body = [&](int64_t v, const VP& pv, const VP& zv) {
    VP pn;
#pragma unroll
    for (int q = 0; q < VEC; ++q) pn.e[q] = (TP)fma(beta[q], (T)pv.e[q], (T)zv.e[q]);
    reinterpret_cast<VP*>(pout)[v] = pn;
  };
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += 2 * stride) {
    const int64_t v2 = v + stride;
    const bool two = v2 < nv;
    const VP p1 = reinterpret_cast<const VP*>(pin)[v], z1 = reinterpret_cast<const VP*>(z)[v];
    VP p2 = p1, z2 = z1;
    if (two) {
      p2 = reinterpret_cast<const VP*>(pin)[v2];
      z2 = reinterpret_cast<const VP*>(z)[v2];
    }
    body(v, p1, z1);
    if (two) body(v2, p2, z2);
  }
}

// ---- focal-node solution: xf[m][c] += alpha_c * p[fnode[m]][c] for the nf nodes whose solution values the caller
//      consumes (resistance-only pair solves read x at the pair's two nodes and at the gathered focal nodes only:
//      src/core.jl:231-232, 685-703). Same arithmetic as the fused x-update of cg_update_r_kernel (one fma per
//      iteration), so the values are bit-identical to the corresponding entries of a full solution vector.
template <class T, class TP, int K>
__global__ __launch_bounds__(256) void cg_focal_x_kernel(const CgScalars* S, const int* __restrict__ fnode, int nf,
                                                         const TP* __restrict__ p, T* __restrict__ xf) {
  if (S->all_done) return;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nf * K; e += gridDim.x * 256) {
    const int m = e / K, c = e % K;
    const T al = (T)S->alpha[c];
    // (alpha == 0: a finished column, or a slot that takes a new pair -- its p may be stale; fma(0, p, x) == x otherwise)
    if (al != T(0)) xf[e] = fma(al, (T)p[(size_t)fnode[m] * K + c], xf[e]);
  }
}

// ---- streaming pair solves: a slot that takes a new pair gets its right-hand side b = e_dst - e_src as the residual
//      (the residual update of the same iteration has just zeroed the column: dia_cg_kernel<DIA_RUPD>, `restart`), its
//      preconditioner-precision copy, and a zero solution at the focal nodes. One workgroup.
template <class T, class TP, int K>
__global__ __launch_bounds__(256) void stream_restart_kernel(const CgScalars* S, T* __restrict__ r, TP* __restrict__ rp,
                                                             int nf, T* __restrict__ xf) {
  for (int c = 0; c < K; ++c) {
    if (!S->ctl.restart[c]) continue;
    for (int m = threadIdx.x; m < nf; m += 256) xf[(size_t)m * K + c] = T(0);
    if (threadIdx.x == 0 && S->ctl.src[c] != S->ctl.dst[c]) {
      const size_t a = (size_t)S->ctl.src[c] * K + c, b = (size_t)S->ctl.dst[c] * K + c;
      r[a] = T(-1);
      r[b] = T(1);
      if (rp) {
        rp[a] = TP(-1);
        rp[b] = TP(1);
      }
    }
  }
}

// ---- scalar kernel 2 of a streaming solve: cg_beta_kernel per SLOT. A slot with ctl.restart set is initialised (what
//      cg_beta_kernel's init call does for a batch: ||b||^2 = 2 is known); a running slot is advanced, and when it meets
//      the configured rule the reference's post-check ||r|| / ||b|| < 1e-4 (core.jl:640, on the fp64 recurrence residual
//      whose ||r||^2 partials the residual update wrote) decides between "done" and polishing to 2.5e-5 on the true
//      residual (cg_reopen_kernel's rule, without the restart). itmax is per pair.
template <int K>
__global__ __launch_bounds__(256) void cg_stream_beta_kernel(CgScalars* S, const double* partials_rz, int nparts_rz,
                                                             const double* partials_rr, int nparts_rr, int criterion,
                                                             double rtol, double atol, int itmax) {
  __shared__ double sm[4];
  __shared__ int s_all;
  if (threadIdx.x == 0) s_all = 1;
  __syncthreads();
  for (int c = 0; c < K; ++c) {
    const double rz = reduce_partials<K>(partials_rz, nparts_rz, c, sm);
    const double rr_now = reduce_partials<K>(partials_rr, nparts_rr, c, sm);
    if (threadIdx.x == 0) {
      if (!S->ctl.active[c]) {
        S->done[c] = 1;
        S->beta[c] = 0.0;
        S->alpha[c] = 0.0;
      } else if (S->ctl.restart[c]) {
        const double rr = S->ctl.src[c] != S->ctl.dst[c] ? 2.0 : 0.0;
        const double mon = criterion == 1 ? sqrt(rr) : sqrt(fabs(rz));
        const double mon2 = sqrt(rr);
        S->bnorm[c] = sqrt(rr);
        S->rnorm0[c] = mon;
        S->eps[c] = atol + rtol * mon;
        S->eps2[c] = atol + rtol * mon2;
        S->rnorm[c] = mon;
        S->gamma[c] = rz;
        S->beta[c] = 0.0;
        S->iters[c] = 0;
        S->relres[c] = rr > 0.0 ? 1.0 : 0.0;
        S->polish[c] = 0;
        S->done[c] = ((mon <= S->eps[c] && (criterion != 2 || mon2 <= S->eps2[c])) || !(rz == rz)) ? 1 : 0;
        S->ctl.restart[c] = 0;
      } else if (!S->done[c]) {
        S->iters[c] += 1;
        const double bn = S->bnorm[c];
        const double relres = bn > 0.0 ? sqrt(rr_now) / bn : sqrt(rr_now);
        S->relres[c] = relres;
        const double mon = (criterion == 1 || S->polish[c]) ? sqrt(rr_now) : sqrt(fabs(rz));
        const double mon2 = sqrt(rr_now);
        S->rnorm[c] = mon;
        const double g = S->gamma[c];
        bool met = (mon <= S->eps[c] && (criterion != 2 || S->polish[c] || mon2 <= S->eps2[c])) || mon + 1.0 <= 1.0;
        if (met && !S->polish[c] && !(relres < 1e-4)) {
          // the configured rule is met but the reference's own check would throw: keep iterating on the true residual
          S->polish[c] = 1;
          S->eps[c] = 2.5e-5 * bn;
          met = sqrt(rr_now) <= S->eps[c];
        }
        if (met) {
          S->done[c] = 1;
          S->beta[c] = 0.0;
        } else if (!(rz == rz) || g == 0.0) {
          S->done[c] = 2;
          S->beta[c] = 0.0;
        } else if (S->iters[c] >= itmax) {
          S->done[c] = 4;
          S->beta[c] = 0.0;
        } else {
          S->beta[c] = rz / g;
        }
        S->gamma[c] = rz;
      } else {
        S->beta[c] = 0.0;
      }
      if (!S->done[c]) s_all = 0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) S->all_done = s_all;
}

// ---- Dirichlet mask: column c of the batch is tied to ground at the nodes gidx[gptr[c] .. gptr[c+1]) -- their entries of
//      the vectors a (and b, optional, may have another type) are set to zero. One-to-all / all-to-one solves differ
//      only in WHICH nodes are grounded (src/raster/onetoall.jl:106-151 -> advanced.jl:282-288 deletes those rows and
//      columns and factorises again); keeping r, z (hence p, x) zero there solves the same reduced system with the
//      hierarchy of the ungrounded matrix.
template <class A, class B, int K>
__global__ __launch_bounds__(256) void mask_grounds_kernel(const int* __restrict__ gptr, const int* __restrict__ gidx,
                                                           A* __restrict__ a, B* __restrict__ b, const int* skip) {
  if (skip && *skip) return;
  const int total = gptr[K];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int c = 0;
    while (c + 1 < K && e >= gptr[c + 1]) ++c;
    const size_t at = (size_t)gidx[e] * K + c;
    if (a) a[at] = A(0);
    if (b) b[at] = B(0);
  }
}

// ---- scalar kernel 2: gamma' = r'z -> convergence test, beta.   criterion 0: monitored norm = sqrt(|r'z|),
// criterion 1: sqrt(r'r) from `partials_rr`, criterion 2: both tests must hold (Dirichlet-masked solves on a shared
// hierarchy, pcg.h).   `init` != 0: first evaluation (sets rnorm0 / eps, no beta).
template <int K>
__global__ __launch_bounds__(256) void cg_beta_kernel(CgScalars* S, const double* partials_rz, int nparts_rz,
                                                      const double* partials_rr, int nparts_rr, int criterion,
                                                      double rtol, double atol, int init, int ncols_active) {
  __shared__ double sm[4];
  __shared__ int s_all;
  if (!init && S->all_done) return;
  if (threadIdx.x == 0) s_all = 1;
  __syncthreads();
  for (int c = 0; c < K; ++c) {
    const double rz = reduce_partials<K>(partials_rz, nparts_rz, c, sm);
    double rr = 0.0;
    if (criterion != 0 || init) rr = reduce_partials<K>(partials_rr, nparts_rr, c, sm);
    if (threadIdx.x == 0) {
      const double mon = criterion == 1 ? sqrt(rr) : sqrt(fabs(rz));
      const double mon2 = sqrt(rr);  // second test of criterion 2
      if (init) {
        S->bnorm[c] = sqrt(rr);  // r0 = b: ||b||_2 for the relative-residual post-check
        S->rnorm0[c] = mon;
        S->eps[c] = atol + rtol * mon;
        S->eps2[c] = atol + rtol * mon2;
        S->rnorm[c] = mon;
        S->gamma[c] = rz;
        S->beta[c] = 0.0;
        S->iters[c] = 0;
        S->done[c] = (c >= ncols_active || (mon <= S->eps[c] && (criterion != 2 || mon2 <= S->eps2[c])) || !(rz == rz)) ? 1 : 0;
      } else {
        if (!S->done[c]) {
          S->iters[c] += 1;
          S->rnorm[c] = mon;
          const double g = S->gamma[c];
          if ((mon <= S->eps[c] && (criterion != 2 || mon2 <= S->eps2[c])) || mon + 1.0 <= 1.0) {
            S->done[c] = 1;
            S->beta[c] = 0.0;
          } else if (!(rz == rz) || g == 0.0) {
            S->done[c] = 2;
            S->beta[c] = 0.0;
          } else {
            S->beta[c] = rz / g;
          }
          S->gamma[c] = rz;
        } else {
          S->beta[c] = 0.0;
        }
      }
      if (!S->done[c]) s_all = 0;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) S->all_done = s_all;
}

// ---- y = s * dinv .* b   (first damped-Jacobi sweep from a zero initial guess)
template <class T, int K>
__global__ __launch_bounds__(256) void scale_dinv_kernel(int64_t n, T* __restrict__ y, const T* __restrict__ b,
                                                         const T* __restrict__ dinv, T s, const int* skip) {
  if (skip && *skip) return;
  const int64_t total = n * K;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    y[e] = s * dinv[e / K] * b[e];
}

// ---- dense coarse solve: y[i,c] = sum_j M[i,j] * b[j,c]   (M = pseudo-inverse of the coarsest operator)
template <class T, int K>
__global__ __launch_bounds__(256) void dense_apply_kernel(int n, const T* __restrict__ M, const T* __restrict__ b,
                                                          T* __restrict__ y, const int* skip) {
  if (skip && *skip) return;
  const int64_t total = (int64_t)n * K;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / K), c = (int)(e % K);
    T s = T(0);
    for (int j = 0; j < n; ++j) s += M[(size_t)i * n + j] * b[(size_t)j * K + c];
    y[e] = s;
  }
}

// ---- the coarsest-level correction of a Dirichlet-masked solve when the coarsest level is not inside the tail kernel
//      (pcg.h, DirichletCoarse; same arithmetic as tail_dirichlet in tail.h). One workgroup per column.
template <class T, int K>
__global__ __launch_bounds__(256) void dense_dirichlet_kernel(int n, const T* __restrict__ v, const int* __restrict__ comp,
                                                              int ncomp, double* __restrict__ coef, int mode,
                                                              const T* __restrict__ b, T* __restrict__ y, const int* skip) {
  if (skip && *skip) return;
  __shared__ double s_red[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  if (tid < ncomp) {
    double s = 0;
    for (int i = 0; i < n; ++i)
      if (comp[i] == tid) s += (double)v[i] * (double)b[(size_t)i * K + c];
    s_red[tid] = s;
  }
  __syncthreads();
  if (mode == 2) {
    if (tid == 0) {
      double smax = 0;
      for (int k = 0; k < ncomp; ++k) smax = fmax(smax, s_red[k]);
      for (int k = 0; k < ncomp; ++k) coef[(size_t)k * kMaxK + c] = (s_red[k] > 1e-9 * smax && s_red[k] > 0) ? 1.0 / s_red[k] : 0.0;
    }
    return;
  }
  for (int i = tid; i < n; i += 256) {
    const int k = comp[i];
    if (k >= 0) y[(size_t)i * K + c] += (T)(s_red[k] * coef[(size_t)k * kMaxK + c]) * v[i];
  }
}

// ---- Dirichlet sets as a marker vector: m[gidx[e], c] = 1 for the entries of column c's set (m zeroed by the caller)
template <class M, int K>
__global__ __launch_bounds__(256) void mark_grounds_kernel(const int* __restrict__ gptr, const int* __restrict__ gidx,
                                                           M* __restrict__ m) {
  const int total = gptr[K];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int c = 0;
    while (c + 1 < K && e >= gptr[c + 1]) ++c;
    m[(size_t)gidx[e] * K + c] = M(1);
  }
}

// ---- penalty vector of the Dirichlet sets: d[j, c] = sum of |a_ij| over the FREE neighbours i of node j of column c's
//      set (the conductance that ties the rest of the graph to the set through j), zero elsewhere (d zeroed by the caller;
//      `mark` from mark_grounds_kernel). Its sum over a connected component is G = 1_f' A_g 1_f of that component.
template <class T, class M, int K>
__global__ __launch_bounds__(256) void dirichlet_penalty_kernel(const int* __restrict__ rp, const int* __restrict__ ci,
                                                                const T* __restrict__ va, const int* __restrict__ gptr,
                                                                const int* __restrict__ gidx, const M* __restrict__ mark,
                                                                M* __restrict__ d) {
  const int total = gptr[K];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    int c = 0;
    while (c + 1 < K && e >= gptr[c + 1]) ++c;
    const int j = gidx[e];
    double s = 0;
    for (int k = rp[j]; k < rp[j + 1]; ++k) {
      const int i = ci[k];
      if (i != j && mark[(size_t)i * K + c] == M(0)) s += fabs((double)va[k]);
    }
    d[(size_t)j * K + c] = (M)s;
  }
}

// ---- relative residual post-check: partials of ||b - A x||^2 come from a DOT-fused SpMV; this finishes it
// (partials_bb null: ||b|| as recorded by the init call of cg_beta_kernel)
template <int K>
__global__ __launch_bounds__(256) void relres_kernel(CgScalars* S, const double* partials_rr, int nparts_rr,
                                                     const double* partials_bb, int nparts_bb) {
  __shared__ double sm[4];
  for (int c = 0; c < K; ++c) {
    const double rr = reduce_partials<K>(partials_rr, nparts_rr, c, sm);
    double bb = S->bnorm[c] * S->bnorm[c];
    if (partials_bb) bb = reduce_partials<K>(partials_bb, nparts_bb, c, sm);
    if (threadIdx.x == 0) {
      if (partials_bb) S->bnorm[c] = sqrt(bb);
      S->relres[c] = bb > 0.0 ? sqrt(rr / bb) : sqrt(rr);
    }
    __syncthreads();
  }
}

// Re-open columns that stopped on the configured rule but whose explicit residual ||Ax-b||/||b|| is not below `limit`:
// they continue on the true-residual criterion until ||r|| <= target * ||b||. The CG state of a finished column is
// r = b - A x (invariant of the updates), z = M^-1 r, p = z, gamma = r'z, beta = 0 -- exactly a restart from x.
template <int K>
__global__ void cg_reopen_kernel(CgScalars* S, double limit, double target, int ncols_active) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int all = 1;
  for (int c = 0; c < K; ++c) {
    if (c < ncols_active && S->done[c] == 1 && !(S->relres[c] < limit)) {
      S->done[c] = 0;
      S->eps[c] = target * S->bnorm[c];
    }
    if (!S->done[c]) all = 0;
  }
  S->all_done = all;
}

}  // namespace csgpu
