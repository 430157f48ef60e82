// =============================================================================
// cs_oracle.cpp -- TEST INFRASTRUCTURE ONLY (CPU oracle / CPU baseline).
//
// A CPU restatement of the arithmetic the reference executes on its CG+AMG path:
//
//   Circuitscape.jl  src/core.jl:164-167   aspreconditioner(smoothed_aggregation(matrix;
//                                            coarse_solver=Pinv, presmoother=GaussSeidel(),
//                                            postsmoother=GaussSeidel()))
//   Circuitscape.jl  src/core.jl:636-643   Krylov.cg(G, curr, M=M, ldiv=true, rtol=T(1e-6),
//                                            itmax=100_000) + true-residual check < 1e-4
//   Circuitscape.jl  src/raster/advanced.jl:307-312  multiple_solve(::AMGSolver, ...)
//
// The arithmetic itself lives in two un-vendored registry dependencies of the reference
// (Project.toml:6,12,29,35): AlgebraicMultigrid.jl (compat "1.2") and Krylov.jl (compat "0.10").
// Their sources are NOT in /root/reference and Julia is not installed, so this file restates
// their published algorithms:
//
//   smoothed_aggregation: SymmetricStrength(theta=0) -> StandardAggregation (PyAMG 3-pass
//   greedy) -> improve_candidates (4 symmetric Gauss-Seidel sweeps on A*B=0) -> fit_candidates
//   -> JacobiProlongation(omega=4/3, local row-abs-sum weighting) -> R = P' -> A_c = R*A*P,
//   while levels+1 < max_levels(10) and n > max_coarse(10); coarse solver = dense pinv;
//   one V(1,1) cycle with symmetric Gauss-Seidel as the preconditioner application.
//
//   Krylov.cg: PCG from x0 = 0, stop when sqrt(r'z) <= atol + rtol*sqrt(r0'z0),
//   atol = sqrt(eps(T)), residual measured in the M^-1 norm.
//
// Parity status: pinned at SOLUTION level against the reference's golden resistance files
// (tests/golden, generated from test/output_verify/*_resistances.out); hierarchy and
// iteration trajectory of the reference are unpinned (no reference test pins them).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product (libcsgpu.so) never links, loads or calls it.
// =============================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

typedef int64_t i64;
typedef int32_t i32;

template <class T>
struct Csr {
  i64 nrows = 0, ncols = 0;
  std::vector<i64> ptr;  // nrows+1
  std::vector<i32> col;
  std::vector<T> val;
  i64 nnz() const { return (i64)col.size(); }
};

// ---------------------------------------------------------------- sparse helpers
template <class T>
static void spmv(const Csr<T>& A, const T* x, T* y) {
  for (i64 i = 0; i < A.nrows; ++i) {
    T s = 0;
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) s += A.val[k] * x[A.col[k]];
    y[i] = s;
  }
}

template <class T>
static Csr<T> transpose(const Csr<T>& A) {
  Csr<T> B;
  B.nrows = A.ncols;
  B.ncols = A.nrows;
  B.ptr.assign(B.nrows + 1, 0);
  for (i64 k = 0; k < A.nnz(); ++k) B.ptr[A.col[k] + 1]++;
  for (i64 i = 0; i < B.nrows; ++i) B.ptr[i + 1] += B.ptr[i];
  B.col.resize(A.nnz());
  B.val.resize(A.nnz());
  std::vector<i64> pos(B.ptr.begin(), B.ptr.end() - 1);
  for (i64 i = 0; i < A.nrows; ++i)
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      i64 p = pos[A.col[k]]++;
      B.col[p] = (i32)i;
      B.val[p] = A.val[k];
    }
  return B;
}

// Gustavson row-wise product with a dense accumulator; output columns sorted; exact zeros dropped
// (the reference calls dropzeros! on the Galerkin product).
template <class T>
static Csr<T> spgemm(const Csr<T>& A, const Csr<T>& B, bool drop_zeros) {
  Csr<T> C;
  C.nrows = A.nrows;
  C.ncols = B.ncols;
  C.ptr.assign(C.nrows + 1, 0);
  std::vector<T> acc(B.ncols, T(0));
  std::vector<char> mark(B.ncols, 0);
  std::vector<i32> touched;
  for (i64 i = 0; i < A.nrows; ++i) {
    touched.clear();
    for (i64 ka = A.ptr[i]; ka < A.ptr[i + 1]; ++ka) {
      const i32 k = A.col[ka];
      const T a = A.val[ka];
      for (i64 kb = B.ptr[k]; kb < B.ptr[k + 1]; ++kb) {
        const i32 j = B.col[kb];
        if (!mark[j]) {
          mark[j] = 1;
          touched.push_back(j);
        }
        acc[j] += a * B.val[kb];
      }
    }
    std::sort(touched.begin(), touched.end());
    for (i32 j : touched) {
      if (!drop_zeros || acc[j] != T(0)) {
        C.col.push_back(j);
        C.val.push_back(acc[j]);
      }
      acc[j] = 0;
      mark[j] = 0;
    }
    C.ptr[i + 1] = (i64)C.col.size();
  }
  return C;
}

// ---------------------------------------------------------------- AMG pieces
// AlgebraicMultigrid.jl SymmetricStrength(theta): keep a_ij iff a_ij^2 >= theta^2 |a_ii||a_jj|,
// diagonal always kept, explicit zeros dropped. Only the PATTERN of S is consumed downstream
// (StandardAggregation looks at structure only), so values are not materialised.
template <class T>
static Csr<T> symmetric_strength(const Csr<T>& A, double theta) {
  const i64 n = A.nrows;
  std::vector<T> diag(n, T(0));
  for (i64 i = 0; i < n; ++i)
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k)
      if (A.col[k] == i) diag[i] += A.val[k];
  for (i64 i = 0; i < n; ++i) diag[i] = std::abs(diag[i]);
  Csr<T> S;
  S.nrows = S.ncols = n;
  S.ptr.assign(n + 1, 0);
  for (i64 i = 0; i < n; ++i) {
    const double epsAii = theta * theta * (double)diag[i];
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const i32 j = A.col[k];
      const T v = A.val[k];
      bool keep = true;
      if (j != i && (double)v * (double)v < epsAii * (double)diag[j]) keep = false;
      if (v == T(0)) keep = false;  // dropzeros!
      if (keep) {
        S.col.push_back(j);
        S.val.push_back(std::abs(v));
      }
    }
    S.ptr[i + 1] = (i64)S.col.size();
  }
  return S;
}

// AlgebraicMultigrid.jl StandardAggregation (port of PyAMG standard_aggregation, 3 passes).
// Returns agg[i] in [0, nagg) or -1 for nodes left unaggregated (isolated).
template <class T>
static i64 standard_aggregation(const Csr<T>& S, std::vector<i64>& x) {
  const i64 n = S.nrows;
  x.assign(n, 0);
  i64 next = 1;
  // pass 1
  for (i64 i = 0; i < n; ++i) {
    if (x[i] != 0) continue;
    bool has_agg_nb = false, has_nb = false;
    for (i64 k = S.ptr[i]; k < S.ptr[i + 1]; ++k) {
      const i32 j = S.col[k];
      if (j != i) {
        has_nb = true;
        if (x[j] != 0) {
          has_agg_nb = true;
          break;
        }
      }
    }
    if (!has_nb) {
      x[i] = -n;
    } else if (!has_agg_nb) {
      x[i] = next;
      for (i64 k = S.ptr[i]; k < S.ptr[i + 1]; ++k) x[S.col[k]] = next;
      ++next;
    }
  }
  // pass 2
  for (i64 i = 0; i < n; ++i) {
    if (x[i] != 0) continue;
    for (i64 k = S.ptr[i]; k < S.ptr[i + 1]; ++k) {
      const i64 xj = x[S.col[k]];
      if (xj > 0) {
        x[i] = -xj;
        break;
      }
    }
  }
  --next;
  // pass 3
  for (i64 i = 0; i < n; ++i) {
    const i64 xi = x[i];
    if (xi != 0) {
      if (xi > 0)
        x[i] = xi - 1;
      else if (xi == -n)
        x[i] = -1;
      else
        x[i] = -xi - 1;
      continue;
    }
    x[i] = next;
    for (i64 k = S.ptr[i]; k < S.ptr[i + 1]; ++k) {
      const i32 j = S.col[k];
      if (x[j] == 0) x[j] = next;
    }
    ++next;
  }
  return next;
}

// One symmetric Gauss-Seidel sweep (forward then backward), AlgebraicMultigrid.jl gs!.
template <class T>
static void gs_sweep(const Csr<T>& A, const T* b, T* x, i64 start, i64 step, i64 stop) {
  for (i64 i = start; i != stop + step; i += step) {
    T rsum = 0, d = 0;
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const i32 j = A.col[k];
      const T v = A.val[k];
      if (j == i)
        d = v;
      else
        rsum += v * x[j];
    }
    if (d != T(0)) x[i] = (b[i] - rsum) / d;
  }
}
template <class T>
static void gs_symmetric(const Csr<T>& A, const T* b, T* x, int iters) {
  if (A.nrows == 0) return;
  for (int it = 0; it < iters; ++it) {
    gs_sweep(A, b, x, 0, 1, A.nrows - 1);
    gs_sweep(A, b, x, A.nrows - 1, -1, 0);
  }
}

// Dense symmetric pseudo-inverse through a cyclic Jacobi eigen-decomposition
// (AlgebraicMultigrid.jl Pinv = pinv(Matrix(A)); Julia's default rtol = eps*min(size)).
template <class T>
static std::vector<T> dense_pinv(const Csr<T>& A) {
  const i64 n = A.nrows;
  std::vector<double> M(n * n, 0.0), V(n * n, 0.0);
  for (i64 i = 0; i < n; ++i)
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) M[i * n + A.col[k]] += (double)A.val[k];
  for (i64 i = 0; i < n; ++i)
    for (i64 j = i + 1; j < n; ++j) {
      double s = 0.5 * (M[i * n + j] + M[j * n + i]);
      M[i * n + j] = M[j * n + i] = s;
    }
  for (i64 i = 0; i < n; ++i) V[i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, dg = 0;
    for (i64 i = 0; i < n; ++i) {
      dg += M[i * n + i] * M[i * n + i];
      for (i64 j = i + 1; j < n; ++j) off += M[i * n + j] * M[i * n + j];
    }
    if (off <= 1e-32 * dg || off == 0) break;
    for (i64 p = 0; p < n; ++p)
      for (i64 q = p + 1; q < n; ++q) {
        const double apq = M[p * n + q];
        if (apq == 0) continue;
        const double app = M[p * n + p], aqq = M[q * n + q];
        const double tau = (aqq - app) / (2 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::abs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), s = t * c;
        for (i64 k = 0; k < n; ++k) {
          const double mkp = M[k * n + p], mkq = M[k * n + q];
          M[k * n + p] = c * mkp - s * mkq;
          M[k * n + q] = s * mkp + c * mkq;
        }
        for (i64 k = 0; k < n; ++k) {
          const double mpk = M[p * n + k], mqk = M[q * n + k];
          M[p * n + k] = c * mpk - s * mqk;
          M[q * n + k] = s * mpk + c * mqk;
        }
        for (i64 k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  double smax = 0;
  for (i64 i = 0; i < n; ++i) smax = std::max(smax, std::abs(M[i * n + i]));
  const double rtol = (double)std::numeric_limits<T>::epsilon() * (double)n;
  std::vector<T> Pinv(n * n, T(0));
  std::vector<double> acc(n * n, 0.0);
  for (i64 e = 0; e < n; ++e) {
    const double lam = M[e * n + e];
    if (std::abs(lam) <= rtol * smax) continue;
    const double inv = 1.0 / lam;
    for (i64 i = 0; i < n; ++i) {
      const double vi = V[i * n + e] * inv;
      for (i64 j = 0; j < n; ++j) acc[i * n + j] += vi * V[j * n + e];
    }
  }
  for (i64 i = 0; i < n * n; ++i) Pinv[i] = (T)acc[i];
  return Pinv;
}

template <class T>
struct Level {
  Csr<T> A, P, R;
};

template <class T>
struct Workspace {
  std::vector<std::vector<T>> res, cx, cb;  // per level
};

template <class T>
struct Hierarchy {
  std::vector<Level<T>> levels;
  Csr<T> finalA;
  std::vector<T> pinv;
  double setup_seconds = 0;
  Workspace<T> make_ws() const {
    Workspace<T> w;
    for (size_t l = 0; l < levels.size(); ++l) {
      w.res.emplace_back(levels[l].A.nrows);
      const i64 nc = levels[l].P.ncols;
      w.cx.emplace_back(nc);
      w.cb.emplace_back(nc);
    }
    return w;
  }
};

struct Opts {
  double theta = 0.0;
  double omega = 4.0 / 3.0;
  int max_levels = 10;
  int max_coarse = 10;
  int improve_iters = 4;
};

template <class T>
static void extend_hierarchy(Hierarchy<T>& H, Csr<T>& A, std::vector<T>& B, const Opts& o) {
  const i64 n = A.nrows;
  Csr<T> S = symmetric_strength(A, o.theta);
  std::vector<i64> agg;
  const i64 nagg = standard_aggregation(S, agg);
  // improve_candidates = GaussSeidel(iter=4) on A*B = 0
  {
    std::vector<T> zero(n, T(0));
    gs_symmetric(A, zero.data(), B.data(), o.improve_iters);
  }
  // fit_candidates: T = B restricted to aggregates, columns normalised; coarse B = column norms
  std::vector<double> nrm2(nagg, 0.0);
  for (i64 i = 0; i < n; ++i)
    if (agg[i] >= 0) nrm2[agg[i]] += (double)B[i] * (double)B[i];
  std::vector<T> Bc(nagg), tval(n, T(0));
  std::vector<double> scale(nagg);
  for (i64 j = 0; j < nagg; ++j) {
    const double nr = std::sqrt(nrm2[j]);
    if (nr > 0.0) {  // (an aggregate whose candidate entries are all zero gets no column)
      scale[j] = 1.0 / nr;
      Bc[j] = (T)nr;
    } else {
      scale[j] = 0;
      Bc[j] = 0;
    }
  }
  Csr<T> Tm;
  Tm.nrows = n;
  Tm.ncols = nagg;
  Tm.ptr.assign(n + 1, 0);
  for (i64 i = 0; i < n; ++i) {
    if (agg[i] >= 0) {
      Tm.col.push_back((i32)agg[i]);
      Tm.val.push_back((T)((double)B[i] * scale[agg[i]]));
    }
    Tm.ptr[i + 1] = (i64)Tm.col.size();
  }
  // JacobiProlongation(4/3), LocalWeighting: P = T - omega * Dloc^-1 * A * T, Dloc_i = sum_j |a_ij|
  Csr<T> DA = A;
  for (i64 i = 0; i < n; ++i) {
    double d = 0;
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) d += std::abs((double)A.val[k]);
    const double w = d != 0 ? o.omega / d : 0.0;
    for (i64 k = A.ptr[i]; k < A.ptr[i + 1]; ++k) DA.val[k] = (T)((double)A.val[k] * w);
  }
  Csr<T> DAT = spgemm(DA, Tm, false);
  // P = T - DAT (pattern union; T's entry (i, agg i) is inside DAT's row pattern when a_ii != 0)
  Csr<T> P;
  P.nrows = n;
  P.ncols = nagg;
  P.ptr.assign(n + 1, 0);
  for (i64 i = 0; i < n; ++i) {
    bool placed = (Tm.ptr[i] == Tm.ptr[i + 1]);
    const i32 tc = placed ? -1 : Tm.col[Tm.ptr[i]];
    const T tv = placed ? T(0) : Tm.val[Tm.ptr[i]];
    for (i64 k = DAT.ptr[i]; k < DAT.ptr[i + 1]; ++k) {
      const i32 j = DAT.col[k];
      if (!placed && tc < j) {
        P.col.push_back(tc);
        P.val.push_back(tv);
        placed = true;
      }
      if (!placed && tc == j) {
        P.col.push_back(j);
        P.val.push_back(tv - DAT.val[k]);
        placed = true;
      } else {
        P.col.push_back(j);
        P.val.push_back(-DAT.val[k]);
      }
    }
    if (!placed) {
      P.col.push_back(tc);
      P.val.push_back(tv);
    }
    P.ptr[i + 1] = (i64)P.col.size();
  }
  Csr<T> R = transpose(P);
  Csr<T> RA = spgemm(R, A, false);
  Csr<T> Ac = spgemm(RA, P, true);
  Level<T> L;
  L.A = std::move(A);
  L.P = std::move(P);
  L.R = std::move(R);
  H.levels.push_back(std::move(L));
  A = std::move(Ac);
  B = std::move(Bc);
}

template <class T>
static Hierarchy<T>* amg_setup(Csr<T> A, const Opts& o) {
  auto t0 = std::chrono::steady_clock::now();
  auto* H = new Hierarchy<T>();
  std::vector<T> B(A.nrows, T(1));
  while ((int)H->levels.size() + 1 < o.max_levels && A.nrows > o.max_coarse) extend_hierarchy(*H, A, B, o);
  H->finalA = std::move(A);
  H->pinv = dense_pinv(H->finalA);
  H->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return H;
}

// One V-cycle from the caller-provided x (AlgebraicMultigrid.jl __solve!).
template <class T>
static void vcycle(const Hierarchy<T>& H, Workspace<T>& w, size_t lvl, T* x, const T* b) {
  if (H.levels.empty()) {  // hierarchy is the coarse solver alone
    const i64 n = H.finalA.nrows;
    for (i64 i = 0; i < n; ++i) {
      double s = 0;
      for (i64 j = 0; j < n; ++j) s += (double)H.pinv[i * n + j] * (double)b[j];
      x[i] = (T)s;
    }
    return;
  }
  const Level<T>& L = H.levels[lvl];
  const i64 n = L.A.nrows;
  gs_symmetric(L.A, b, x, 1);
  T* res = w.res[lvl].data();
  spmv(L.A, x, res);
  for (i64 i = 0; i < n; ++i) res[i] = b[i] - res[i];
  T* cb = w.cb[lvl].data();
  T* cx = w.cx[lvl].data();
  spmv(L.R, res, cb);
  const i64 nc = L.P.ncols;
  std::fill(cx, cx + nc, T(0));
  if (lvl + 1 == H.levels.size()) {
    for (i64 i = 0; i < nc; ++i) {
      T s = 0;
      for (i64 j = 0; j < nc; ++j) s += H.pinv[i * nc + j] * cb[j];
      cx[i] = s;
    }
  } else {
    vcycle(H, w, lvl + 1, cx, cb);
  }
  spmv(L.P, cx, res);
  for (i64 i = 0; i < n; ++i) x[i] += res[i];
  gs_symmetric(L.A, b, x, 1);
}

template <class T>
static void precond(const Hierarchy<T>& H, Workspace<T>& w, T* z, const T* r, i64 n) {
  std::fill(z, z + n, T(0));
  vcycle(H, w, 0, z, r);
}

struct PcgResult {
  int iters;
  double final_mnorm;   // sqrt(r'z) at exit
  double true_relres;   // ||A x - b|| / ||b||
  double seconds;
  int status;           // 0 converged, 1 itmax, 2 breakdown
};

// Krylov.jl cg (v0.10) with a left preconditioner applied through ldiv!.
//   criterion 0: Krylov.jl rule   sqrt(r'z) <= atol + rtol*sqrt(r0'z0)
//   criterion 1: tight/true rule  ||r||_2   <= atol + rtol*||b||_2      (oracle "tight mode")
template <class T>
static PcgResult pcg(const Hierarchy<T>& H, const Csr<T>& A, const T* b, T* x, double rtol, double atol,
                     int itmax, int criterion) {
  auto t0 = std::chrono::steady_clock::now();
  const i64 n = A.nrows;
  Workspace<T> w = H.make_ws();
  std::vector<T> r(b, b + n), z(n), p(n), Ap(n);
  std::fill(x, x + n, T(0));
  auto dot = [&](const T* a, const T* c) {
    T s = 0;
    for (i64 i = 0; i < n; ++i) s += a[i] * c[i];
    return s;
  };
  precond(H, w, z.data(), r.data(), n);
  std::copy(z.begin(), z.end(), p.begin());
  T gamma = dot(r.data(), z.data());
  double rnorm = criterion == 0 ? std::sqrt((double)gamma) : std::sqrt((double)dot(r.data(), r.data()));
  const double eps_stop = atol + rtol * rnorm;
  double pnorm2 = (double)gamma;
  PcgResult res{0, rnorm, 0, 0, 0};
  bool solved = rnorm <= eps_stop;
  int iter = 0;
  while (!solved && iter < itmax) {
    spmv(A, p.data(), Ap.data());
    const T pAp = dot(p.data(), Ap.data());
    if ((double)pAp <= (double)std::numeric_limits<T>::epsilon() * pnorm2) {
      res.status = 2;
      break;
    }
    const T alpha = gamma / pAp;
    for (i64 i = 0; i < n; ++i) x[i] += alpha * p[i];
    for (i64 i = 0; i < n; ++i) r[i] -= alpha * Ap[i];
    precond(H, w, z.data(), r.data(), n);
    const T gamma_next = dot(r.data(), z.data());
    rnorm = criterion == 0 ? std::sqrt(std::abs((double)gamma_next)) : std::sqrt((double)dot(r.data(), r.data()));
    const T beta = gamma_next / gamma;
    pnorm2 = (double)gamma_next + (double)beta * (double)beta * pnorm2;
    gamma = gamma_next;
    for (i64 i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    ++iter;
    solved = rnorm <= eps_stop || (rnorm + 1.0 <= 1.0);
  }
  if (!solved && res.status == 0) res.status = 1;
  res.iters = iter;
  res.final_mnorm = rnorm;
  // reference post-check (core.jl:640): ||G v - curr|| / ||curr||
  spmv(A, x, Ap.data());
  double rr = 0, bb = 0;
  for (i64 i = 0; i < n; ++i) {
    const double d = (double)Ap[i] - (double)b[i];
    rr += d * d;
    bb += (double)b[i] * (double)b[i];
  }
  res.true_relres = bb > 0 ? std::sqrt(rr / bb) : std::sqrt(rr);
  res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return res;
}

template <class T>
struct Handle {
  Csr<T> A;                     // the matrix CG sees (level-0 copy kept separately from the hierarchy)
  std::unique_ptr<Hierarchy<T>> H;
};

template <class T>
static Csr<T> make_csr(i64 n, const i64* rowptr, const i64* colidx, const double* vals, int index_base) {
  Csr<T> A;
  A.nrows = A.ncols = n;
  A.ptr.resize(n + 1);
  for (i64 i = 0; i <= n; ++i) A.ptr[i] = rowptr[i] - index_base;
  const i64 nnz = A.ptr[n];
  A.col.resize(nnz);
  A.val.resize(nnz);
  for (i64 k = 0; k < nnz; ++k) {
    A.col[k] = (i32)(colidx[k] - index_base);
    A.val[k] = (T)vals[k];
  }
  return A;
}

}  // namespace

// ---------------------------------------------------------------- C interface (ctypes)
extern "C" {

struct cso_opts {
  double theta;
  double omega;
  int max_levels;
  int max_coarse;
  int improve_iters;
  int reserved;
};

struct cso_result {
  int iters;
  int status;
  double final_mnorm;
  double true_relres;
  double seconds;
};

// val_bytes: 8 -> fp64 arithmetic, 4 -> fp32 arithmetic (input arrays are always double/int64).
void* cso_setup(int64_t n, const int64_t* rowptr, const int64_t* colidx, const double* vals, int index_base,
                int val_bytes, const cso_opts* o) {
  Opts opts;
  if (o) {
    opts.theta = o->theta;
    opts.omega = o->omega;
    opts.max_levels = o->max_levels;
    opts.max_coarse = o->max_coarse;
    opts.improve_iters = o->improve_iters;
  }
  if (val_bytes == 8) {
    auto* h = new Handle<double>();
    h->A = make_csr<double>(n, rowptr, colidx, vals, index_base);
    h->H.reset(amg_setup<double>(h->A, opts));
    return h;
  } else {
    auto* h = new Handle<float>();
    h->A = make_csr<float>(n, rowptr, colidx, vals, index_base);
    h->H.reset(amg_setup<float>(h->A, opts));
    return h;
  }
}

void cso_free(void* handle, int val_bytes) {
  if (val_bytes == 8)
    delete (Handle<double>*)handle;
  else
    delete (Handle<float>*)handle;
}

// info[0]=levels (incl. coarsest), info[1]=setup seconds, info[2]=operator complexity,
// then (n_l, nnz_l) for each level, up to cap doubles.
int cso_info(void* handle, int val_bytes, double* info, int cap) {
  auto fill = [&](auto* h) {
    const auto& H = *h->H;
    int nl = (int)H.levels.size() + 1;
    double nnz0 = H.levels.empty() ? (double)H.finalA.nnz() : (double)H.levels[0].A.nnz();
    double tot = 0;
    for (auto& L : H.levels) tot += (double)L.A.nnz();
    tot += (double)H.finalA.nnz();
    if (cap > 0) info[0] = nl;
    if (cap > 1) info[1] = H.setup_seconds;
    if (cap > 2) info[2] = tot / nnz0;
    int p = 3;
    for (int l = 0; l < nl; ++l) {
      const auto& A = l < (int)H.levels.size() ? H.levels[l].A : H.finalA;
      if (p + 1 < cap) {
        info[p] = (double)A.nrows;
        info[p + 1] = (double)A.nnz();
      }
      p += 2;
    }
    return nl;
  };
  return val_bytes == 8 ? fill((Handle<double>*)handle) : fill((Handle<float>*)handle);
}

// Solve A x = b for nrhs right-hand sides (column-major n x nrhs, double in/out), nthreads workers
// (one right-hand side per task, mirroring Threads.@spawn per source point, core.jl:269).
int cso_solve(void* handle, int val_bytes, const double* b, double* x, int64_t nrhs, double rtol, double atol,
              int itmax, int criterion, int nthreads, cso_result* results) {
  auto run = [&](auto* h, auto tag) {
    typedef decltype(tag) T;
    const i64 n = h->A.nrows;
    std::atomic<i64> next(0);
    auto worker = [&]() {
      std::vector<T> bb(n), xx(n);
      for (;;) {
        const i64 c = next.fetch_add(1);
        if (c >= nrhs) break;
        for (i64 i = 0; i < n; ++i) bb[i] = (T)b[c * n + i];
        const double at = atol < 0 ? std::sqrt((double)std::numeric_limits<T>::epsilon()) : atol;
        PcgResult r = pcg<T>(*h->H, h->A, bb.data(), xx.data(), rtol, at, itmax, criterion);
        for (i64 i = 0; i < n; ++i) x[c * n + i] = (double)xx[i];
        if (results) {
          results[c].iters = r.iters;
          results[c].status = r.status;
          results[c].final_mnorm = r.final_mnorm;
          results[c].true_relres = r.true_relres;
          results[c].seconds = r.seconds;
        }
      }
    };
    int nt = std::max(1, std::min<int>(nthreads, (int)nrhs));
    if (nt == 1) {
      worker();
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back(worker);
      for (auto& t : th) t.join();
    }
    return 0;
  };
  return val_bytes == 8 ? run((Handle<double>*)handle, double()) : run((Handle<float>*)handle, float());
}

// Pair solves without materialising n x npairs on the caller side: rhs = e_dst - e_src (core.jl:224-226),
// v -= v[src], R = v[dst] - v[src] (core.jl:231-232); optionally gathers v at `gather` nodes per pair.
int cso_solve_pairs(void* handle, int val_bytes, const int64_t* src, const int64_t* dst, int64_t npairs,
                    const int64_t* gather, int64_t ngather, double* gathered, double* resist, double rtol,
                    double atol, int itmax, int criterion, int nthreads, cso_result* results) {
  auto run = [&](auto* h, auto tag) {
    typedef decltype(tag) T;
    const i64 n = h->A.nrows;
    std::atomic<i64> next(0);
    auto worker = [&]() {
      std::vector<T> bb(n), xx(n);
      for (;;) {
        const i64 c = next.fetch_add(1);
        if (c >= npairs) break;
        std::fill(bb.begin(), bb.end(), T(0));
        bb[src[c]] = T(-1);
        bb[dst[c]] = T(1);
        const double at = atol < 0 ? std::sqrt((double)std::numeric_limits<T>::epsilon()) : atol;
        PcgResult r = pcg<T>(*h->H, h->A, bb.data(), xx.data(), rtol, at, itmax, criterion);
        const T vs = xx[src[c]];
        resist[c] = (double)(xx[dst[c]] - vs);
        for (i64 g = 0; g < ngather; ++g) gathered[c * ngather + g] = (double)(xx[gather[g]] - vs);
        if (results) {
          results[c].iters = r.iters;
          results[c].status = r.status;
          results[c].final_mnorm = r.final_mnorm;
          results[c].true_relres = r.true_relres;
          results[c].seconds = r.seconds;
        }
      }
    };
    int nt = std::max(1, std::min<int>(nthreads, (int)npairs));
    if (nt == 1) {
      worker();
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back(worker);
      for (auto& t : th) t.join();
    }
    return 0;
  };
  return val_bytes == 8 ? run((Handle<double>*)handle, double()) : run((Handle<float>*)handle, float());
}

// Plain CSR SpMV timing helper for the CPU side of the SpMV GB/s comparison.
double cso_spmv_seconds(void* handle, int val_bytes, int reps) {
  auto run = [&](auto* h, auto tag) {
    typedef decltype(tag) T;
    const i64 n = h->A.nrows;
    std::vector<T> x(n, T(1)), y(n);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      spmv(h->A, x.data(), y.data());
      x[0] += y[0] * T(1e-30);
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
  };
  return val_bytes == 8 ? run((Handle<double>*)handle, double()) : run((Handle<float>*)handle, float());
}

}  // extern "C"
