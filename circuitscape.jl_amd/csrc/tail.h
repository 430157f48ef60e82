// tail.h -- the coarse tail of the V-cycle in ONE launch.
//
// Below a few thousand rows a level's products are pure launch latency: ~8 dependent launches of 5-10 us per level and
// iteration, which is most of a PCG iteration on rasters up to ~2000^2 (BASELINE configs[1], 1000 x 1000: 0.73 ms per
// iteration of which the fine level's HBM traffic explains 0.2 ms) and still ~0.3 ms per iteration at 10000^2. The
// columns of a batch are independent right-hand sides, so no grid-wide synchronisation is needed to run the whole
// recursion below level `first` inside one kernel: workgroup c owns column c of the batch, keeps that column's vectors of
// every tail level contiguously in a (cache-resident) scratch area and walks the levels down and up with workgroup
// barriers only. Same arithmetic as the launch-per-product path of pcg.h (first pre-sweep x = w D^-1 b, damped-Jacobi
// sweeps, residual, restriction with R, prolongation fused with the first post-sweep through Q = P - w D^-1 A P, dense
// pseudo-inverse on the coarsest level), rows summed in CSR order.
// GPU counterpart of the coarse part of AlgebraicMultigrid.jl's __solve! (reference call site
// src/core.jl:164-167, 178) -- the reference recurses level by level on the host.
#pragma once
#include "amg_setup.h"
#include "blas1.h"

namespace csgpu {

static const int kTailMaxLevels = 10;
static const int kTailThreads = 1024;
static const int kTailMaxSweeps = 8;

template <class T>
struct TailLevel {
  int n, nu, has_q;
  T omega;               // weight of the coarsest level's Jacobi sweeps (no dense inverse)
  T w[kTailMaxSweeps];   // weight of sweep s (damped Jacobi: all equal; Chebyshev levels: amg_setup.h)
  const int *arp, *aci;
  const T* ava;
  const T* dinv;
  const int *rrp, *rci;  // restriction R (rows of the NEXT level)
  const T* rva;
  const int *qrp, *qci;  // Q (has_q) or P
  const T* qva;
  const T* cand;         // candidate vector of the level (null: no projection)
  int64_t off;           // offset of this level's four vectors inside a column's scratch area
};

template <class T>
struct TailArgs {
  int nlev;              // tail levels (the last one is the hierarchy's coarsest level)
  int dense;             // coarsest level solved with the dense pseudo-inverse (else 8 Jacobi sweeps)
  const T* inv;          // [n_last][n_last]
  TailLevel<T> lev[kTailMaxLevels];
  T* scratch;            // [K][stride]
  int64_t stride;
  const T* bin;          // [n_first][K] right-hand side of the first tail level (interleaved batch layout)
  T* xout;               // [n_first][K] its solution
  T cand_inv_norm2;      // 1 / |candidate|^2 (the same on every level); 0: no projection
  const int* skip;
  // Dirichlet-masked solves (pcg.h, DirichletCoarse): x_coarsest += sum_k v_k (v_k'b) coef[k][c], v_k = the coarsest level's
  // candidate on component k (dir_mode 1); dir_mode 2 writes coef from b instead (probe), 0: none
  const T* dir_cand;
  const int* dir_comp;
  double* dir_coef;  // [ncomp][kMaxK]
  int dir_ncomp, dir_mode;
};

// b <- b - v (v'b) / (v'v): in exact arithmetic the restricted right-hand sides of a near-singular Laplacian system have no
// component along the candidate (v_c'(R r) = (P v_c)'r = 1'r = 0 for every right-hand side the reference solves on such a
// system); in an fp32 hierarchy the restriction chain leaves one, and the deeper the level the larger it is relative to
// the level's own scale (amg_setup.h, component_candidates).
template <class T>
__device__ __forceinline__ void tail_project(T* b, const T* __restrict__ v, int n, T inv_norm2, double* s_red, int tid) {
  double s = 0;
  for (int i = tid; i < n; i += kTailThreads) s += (double)v[i] * (double)b[i];
  s_red[tid] = s;
  __syncthreads();
  for (int h = kTailThreads / 2; h > 0; h >>= 1) {
    if (tid < h) s_red[tid] += s_red[tid + h];
    __syncthreads();
  }
  const T c = (T)(s_red[0] * (double)inv_norm2);
  __syncthreads();
  for (int i = tid; i < n; i += kTailThreads) b[i] -= c * v[i];
  __syncthreads();
}

// The coarsest-level correction of a Dirichlet-masked solve (pcg.h, DirichletCoarse), column c of the batch.
// s_k = sum over the coarse nodes of component k of v_i b_i (one thread per component, fixed order: bit-reproducible).
// mode 1: x_i += v_i s_k coef[k][c].  mode 2 (probe: b is the restricted penalty vector, s_k = G_k): coef[k][c] = 1 / s_k for
// the components that hold a share of the column's Dirichlet set, 0 for the others.
template <class T>
__device__ __forceinline__ void tail_dirichlet(T* x, const T* b, const T* __restrict__ v, const int* __restrict__ comp, int n,
                                               int ncomp, double* coef, int c, int mode, double* s_red, int tid) {
  if (tid < ncomp) {
    double s = 0;
    for (int i = 0; i < n; ++i)
      if (comp[i] == tid) s += (double)v[i] * (double)b[i];
    s_red[tid] = s;
  }
  __syncthreads();
  if (mode == 2) {
    if (tid == 0) {
      double smax = 0;
      for (int k = 0; k < ncomp; ++k) smax = fmax(smax, s_red[k]);
      for (int k = 0; k < ncomp; ++k) coef[(size_t)k * kMaxK + c] = (s_red[k] > 1e-9 * smax && s_red[k] > 0) ? 1.0 / s_red[k] : 0.0;
    }
  } else {
    for (int i = tid; i < n; i += kTailThreads) {
      const int k = comp[i];
      if (k >= 0) x[i] += (T)(s_red[k] * coef[(size_t)k * kMaxK + c]) * v[i];
    }
  }
  __syncthreads();
}

template <class T>
__device__ __forceinline__ T tail_row(const int* __restrict__ rp, const int* __restrict__ ci, const T* __restrict__ va,
                                      const T* x, int i) {
  // four independent gathers in flight: a single workgroup has no other way to hide the latency of the chain
  T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
  int k = rp[i];
  const int e = rp[i + 1];
  for (; k + 4 <= e; k += 4) {
    const int c0 = ci[k], c1 = ci[k + 1], c2 = ci[k + 2], c3 = ci[k + 3];
    const T v0 = va[k], v1 = va[k + 1], v2 = va[k + 2], v3 = va[k + 3];
    s0 += v0 * x[c0];
    s1 += v1 * x[c1];
    s2 += v2 * x[c2];
    s3 += v3 * x[c3];
  }
  for (; k < e; ++k) s0 += va[k] * x[ci[k]];
  return (s0 + s1) + (s2 + s3);
}

template <class T, int K>
__global__ __launch_bounds__(kTailThreads) void coarse_tail_kernel(TailArgs<T> a) {
  if (a.skip && *a.skip) return;
  const int c = blockIdx.x, tid = threadIdx.x;
  T* ws = a.scratch + (size_t)c * a.stride;
  __shared__ T* s_x[kTailMaxLevels];  // which of a level's two solution buffers holds x after the way down
  __shared__ T s_part[kTailThreads];
  __shared__ double s_red[kTailThreads];
  {
    const TailLevel<T>& L = a.lev[0];
    T* b = ws + L.off;
    for (int i = tid; i < L.n; i += kTailThreads) b[i] = a.bin[(size_t)i * K + c];
    __syncthreads();
    if (a.cand_inv_norm2 != T(0) && L.cand) tail_project(b, L.cand, L.n, a.cand_inv_norm2, s_red, tid);
  }
  __syncthreads();
  // ---- down
  for (int l = 0; l < a.nlev; ++l) {
    const TailLevel<T>& L = a.lev[l];
    const int n = L.n;
    T* b = ws + L.off;
    T* x = b + n;
    T* y = x + n;
    T* r = y + n;
    if (l + 1 == a.nlev) {
      if (a.dense && n <= kTailThreads) {
        // x = M b with the column range split over kTailThreads / n_pad groups of threads (partial sums through LDS);
        // M is the pseudo-inverse of a symmetric matrix, so M[j][i] (contiguous over the threads of a group) stands for
        // M[i][j]
        const int npad = (n + 63) & ~63, parts = kTailThreads / npad, chunk = (n + parts - 1) / parts;
        const int i = tid % npad, part = tid / npad;
        T s = T(0);
        if (part < parts && i < n) {
          const int j1 = min(n, (part + 1) * chunk);
          for (int j = part * chunk; j < j1; ++j) s += a.inv[(size_t)j * n + i] * b[j];
        }
        if (part < parts) s_part[part * npad + i] = s;
        __syncthreads();
        if (tid < n) {
          T t = T(0);
          for (int q = 0; q < parts; ++q) t += s_part[q * npad + tid];
          x[tid] = t;
        }
      } else if (a.dense) {
        for (int i = tid; i < n; i += kTailThreads) {
          T s = T(0);
          for (int j = 0; j < n; ++j) s += a.inv[(size_t)j * n + i] * b[j];
          x[i] = s;
        }
      } else {
        for (int i = tid; i < n; i += kTailThreads) x[i] = L.omega * L.dinv[i] * b[i];
        __syncthreads();
        for (int s = 0; s < 8; ++s) {
          for (int i = tid; i < n; i += kTailThreads)
            y[i] = x[i] + L.omega * L.dinv[i] * (b[i] - tail_row(L.arp, L.aci, L.ava, x, i));
          __syncthreads();
          T* t = x;
          x = y;
          y = t;
        }
      }
      if (a.dense && a.dir_mode) {
        __syncthreads();
        tail_dirichlet(x, b, a.dir_cand, a.dir_comp, n, a.dir_ncomp, a.dir_coef, c, a.dir_mode, s_red, tid);
      }
      if (tid == 0) s_x[l] = x;
      __syncthreads();
      break;
    }
    for (int i = tid; i < n; i += kTailThreads) x[i] = L.w[0] * L.dinv[i] * b[i];
    __syncthreads();
    for (int s = 1; s < L.nu; ++s) {
      const T ws = L.w[s];
      for (int i = tid; i < n; i += kTailThreads)
        y[i] = x[i] + ws * L.dinv[i] * (b[i] - tail_row(L.arp, L.aci, L.ava, x, i));
      __syncthreads();
      T* t = x;
      x = y;
      y = t;
    }
    for (int i = tid; i < n; i += kTailThreads) r[i] = b[i] - tail_row(L.arp, L.aci, L.ava, x, i);
    if (tid == 0) s_x[l] = x;
    __syncthreads();
    {
      const TailLevel<T>& Lc = a.lev[l + 1];
      T* bc = ws + Lc.off;
      for (int i = tid; i < Lc.n; i += kTailThreads) bc[i] = tail_row(L.rrp, L.rci, L.rva, r, i);
      __syncthreads();
      if (a.cand_inv_norm2 != T(0) && Lc.cand) tail_project(bc, Lc.cand, Lc.n, a.cand_inv_norm2, s_red, tid);
    }
    __syncthreads();
  }
  // ---- up
  for (int l = a.nlev - 2; l >= 0; --l) {
    const TailLevel<T>& L = a.lev[l];
    const int n = L.n;
    T* b = ws + L.off;
    T* x = s_x[l];
    T* y = (x == b + n) ? b + 2 * n : b + n;
    T* r = b + 3 * n;
    const T* xc = s_x[l + 1];
    int first = 0;
    if (L.has_q) {  // x' = x + w_0 D^-1 r + Q x_c : prolongation and first post-sweep in one product
      for (int i = tid; i < n; i += kTailThreads)
        y[i] = x[i] + L.w[0] * L.dinv[i] * r[i] + tail_row(L.qrp, L.qci, L.qva, xc, i);
      first = 1;
    } else {
      for (int i = tid; i < n; i += kTailThreads) y[i] = x[i] + tail_row(L.qrp, L.qci, L.qva, xc, i);
    }
    __syncthreads();
    {
      T* t = x;
      x = y;
      y = t;
    }
    for (int s = first; s < L.nu; ++s) {
      const T ws = L.w[s];
      for (int i = tid; i < n; i += kTailThreads)
        y[i] = x[i] + ws * L.dinv[i] * (b[i] - tail_row(L.arp, L.aci, L.ava, x, i));
      __syncthreads();
      T* t = x;
      x = y;
      y = t;
    }
    if (tid == 0) s_x[l] = x;
    __syncthreads();
  }
  {
    const TailLevel<T>& L = a.lev[0];
    const T* x = s_x[0];
    for (int i = tid; i < L.n; i += kTailThreads) a.xout[(size_t)i * K + c] = x[i];
  }
}

}  // namespace csgpu
