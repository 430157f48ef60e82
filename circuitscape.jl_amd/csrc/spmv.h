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
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

#include "prims.h"

namespace csgpu {

// The matrix (val / col) is read exactly once per launch by exactly one workgroup. Loading it non-temporally
// (-DCSGPU_NT=1) so that it does not evict the gathered x rows from L2 was measured on MI355X (10000^2, K=16): L2 fetch
// of the [S Q] product -6 %, batch time 625 vs 614 ms on the same box -- no gain, so plain loads are the default.
#ifndef CSGPU_NT
#define CSGPU_NT 0
#endif
template <class V>
__device__ __forceinline__ V stream_load(const V* p) {
#if CSGPU_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

enum SpmvEpi {
  EPI_PLAIN = 0,   // y = A x
  EPI_RESID = 1,   // y = b - A x
  EPI_JACOBI = 2,  // y = x + omega * dinv .* (b - A x)
  EPI_ADD = 3,     // y = xadd + A x          (prolongation: xadd may alias y)
  EPI_QADD = 4     // y = xadd + omega * dinv .* b + A x   (fused prolongation + first post-smoothing sweep, A = Q)
};

inline int spmv_grid_cap();  // workgroups per launch (defined with spmv_grid below)

static const int kSpmvRows = 256;   // rows per workgroup pass (row-block granularity of the traversal order)
static const int kSpmvTile = 2560;  // nonzeros staged in LDS per tile

// T: matrix values, accumulation and output; XT: type of the gathered input vector (XT == T except for the CG
// product of the mixed-precision path, where the search direction p is stored in the preconditioner's precision).
template <class T, class XT = T>
struct SpmvArgs {
  int nrows;
  long long nnz;    // number of stored entries (selects the long-row kernel for plain products)
  const int* rowptr;
  const int* col;
  const T* val;
  const XT* x;      // input, interleaved [ncols][K]
  T* y;             // output, interleaved [nrows][K]
  const T* b;       // EPI_RESID / EPI_JACOBI
  const T* xadd;    // EPI_ADD (may alias y); EPI_JACOBI reads x itself for the row's own value
  const T* dinv;    // EPI_JACOBI: 1 / a_ii
  T omega;          // EPI_JACOBI
  const T* dotw;    // DOT: partial[c] += dotw[row*K+c] * y[row*K+c]; null => dot with x itself (square A)
  double* partials; // DOT: [gridDim.x][K]
  const int* order; // optional traversal order of the row blocks (band-aware, see spmv_block_order); may be null
  const int* skip;  // optional device flag: when *skip != 0 the launch returns immediately (all columns converged)
  const int* order_lr;  // traversal order of the long-row kernel's row blocks (spmv_block_order_rect); may be null
};

// N adjacent values moved with one (up to 16-byte) memory instruction
template <class T, int N>
struct alignas(sizeof(T) * N) SpmvVec {
  T e[N];
};

// y is written once and (at these sizes) not re-read before it has left the caches: -DCSGPU_NT_STORE=1 stores it
// non-temporally so the write stream does not allocate in L2 next to the x rows being re-used.
#ifndef CSGPU_NT_STORE
#define CSGPU_NT_STORE 0
#endif
template <class T, int N>
__device__ __forceinline__ void stream_store(SpmvVec<T, N>* p, const SpmvVec<T, N>& v) {
#if CSGPU_NT_STORE && defined(__HIPCC__)
  typedef T VT __attribute__((ext_vector_type(N)));
  VT t;
#pragma unroll
  for (int q = 0; q < N; ++q) t[q] = v.e[q];
  __builtin_nontemporal_store(t, reinterpret_cast<VT*>(p));
#else
  *p = v;
#endif
}

template <class T, int K, int EPI, bool DOT, class XT, int TPL = kSpmvTile / 256>
__global__ __launch_bounds__(256) void spmv_kernel(SpmvArgs<T, XT> a) {
  // (K == 1: products staged in LDS; K > 1: matrix staged in LDS, 4 gathers in flight per lane)
  // K = 16: half-size row blocks (128 rows / 1280 nonzeros per tile) keep the per-lane state (rows owned by a lane) the
  // same as at K = 8, so the kernel stays at 4 waves/SIMD; the traversal order (built for 256-row blocks) is followed
  // at half-block granularity.
  // K = 32 with 8-byte x: the columns are processed as TWO halves of 16 (KH = 16), one after the other, each exactly like a
  // K = 16 launch on a vector of stride 32 -- with all 32 columns at once a lane would own 8 rows and the kernel ran
  // 1.75x slower per column (measured: the CSR level 1 of a 10000^2 raster with 15 % NODATA, profiles/
  // r4_nodata_kernel_stats_k32.csv). The matrix tile of the second half comes out of L2.
  constexpr int KH = (K == 32 && sizeof(XT) == 8) ? 16 : K;
  constexpr int NH = K / KH;
  constexpr int SPLIT = KH >= 16 ? 2 : 1;
  constexpr int ROWS = kSpmvRows / SPLIT;
  constexpr int TILE = 256 * TPL / SPLIT;  // TPL = nonzeros per lane and 256-row tile (10; 16 for the [S Q] product)
  static_assert(TILE % 256 == 0, "every lane streams TILE / 256 nonzeros per tile");
  __shared__ int s_rp[ROWS + 1];
  __shared__ T s_val[TILE];
  __shared__ int s_col[K > 1 ? TILE : 1];
  __shared__ double s_red[4 * (K > 1 ? K : 1)];

  if (a.skip && *a.skip) return;  // wave-uniform: the PCG loop already converged, this launch is a no-op
  const int tid = threadIdx.x;
  // K > 1: a row's K-wide x segment is gathered as 16-byte vectors: CPL adjacent columns per lane, LPR lanes per row.
  constexpr int VEC = 16 / (int)sizeof(XT);
  constexpr int CPL = KH < VEC ? KH : VEC;   // columns per lane
  constexpr int LPR = KH / CPL;             // lanes per row
  constexpr int RPP = 256 / LPR;           // rows per pass (256 lanes / LPR)
  constexpr int NPASS = ROWS / RPP;        // rows owned by one lane
  typedef SpmvVec<XT, CPL> XV;  // gathered x segment
  typedef SpmvVec<T, CPL> YV;   // epilogue vectors (b, xadd, dotw, y)
  const int c0b = K > 1 ? (tid % LPR) * CPL : 0;  // first column owned by the lane (within a half)
  double dot_acc[NH][CPL];
#pragma unroll
  for (int hh = 0; hh < NH; ++hh)
#pragma unroll
    for (int q = 0; q < CPL; ++q) dot_acc[hh][q] = 0.0;

  // Row-block -> workgroup mapping. Workgroup b is dispatched to XCD b % 8 (observed, used for speed only): give each
  // XCD one CONTIGUOUS eighth of the row blocks and let its workgroups march through it in order, so the x rows a
  // raster row block shares with its neighbours +-nrows_of_raster away are re-used out of that XCD's private 4 MiB L2
  // instead of being fetched by three different XCDs.
  const int nblocks = SPLIT * ((a.nrows + kSpmvRows - 1) / kSpmvRows);  // halves of a partial last block may be empty
  int rb_first = blockIdx.x, rb_last = nblocks, rb_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, chunk = (nblocks + 7) >> 3;
    rb_first = xcd * chunk + (blockIdx.x >> 3);
    rb_last = min(nblocks, (xcd + 1) * chunk);
    rb_step = gridDim.x >> 3;
  }
  for (int pos = rb_first; pos < rb_last; pos += rb_step) {
    const int rb = a.order ? a.order[pos / SPLIT] * SPLIT + (pos % SPLIT) : pos;
    const int row0 = rb * ROWS;
    if (row0 >= a.nrows) continue;  // second half of a final, partial 256-row block
    const int nr = min(ROWS, a.nrows - row0);
    __syncthreads();  // previous pass finished with s_rp / tiles
    for (int t = tid; t <= nr; t += 256) s_rp[t] = a.rowptr[row0 + t];
    __syncthreads();
    const int kbeg = s_rp[0], kend = s_rp[nr];
#pragma unroll 1
    for (int hh = 0; hh < NH; ++hh) {
    const int c0 = c0b + hh * KH;
    if (hh > 0) __syncthreads();  // the first half finished with the staged tile

    T acc[NPASS][CPL];
    // the row's own x values, captured when the diagonal entry is gathered (saves re-reading x in the epilogue)
    XV xself[NPASS];
    bool have_self[NPASS];
    constexpr bool WANT_SELF = (K > 1) && (EPI == EPI_JACOBI || DOT);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      have_self[p] = false;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        acc[p][q] = T(0);
        xself[p].e[q] = XT(0);
      }
    }

    for (int ts = kbeg; ts < kend; ts += TILE) {
      const int te = min(kend, ts + TILE);
      if (ts != kbeg) __syncthreads();
      // ---- phase 2: coalesced stream of the tile
      {
        constexpr int U = TILE / 256;  // all of a lane's loads are issued before the first dependent use
        T vv[U];
        int cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = ts + tid + u * 256;
          vv[u] = k < te ? stream_load(a.val + k) : T(0);
          cc[u] = k < te ? stream_load(a.col + k) : 0;
        }
        if (K == 1) {
          T xx[U];
#pragma unroll
          for (int u = 0; u < U; ++u) xx[u] = (ts + tid + u * 256 < te) ? (T)a.x[cc[u]] : T(0);
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (ts + tid + u * 256 < te) s_val[tid + u * 256] = vv[u] * xx[u];
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (ts + tid + u * 256 < te) {
              s_val[tid + u * 256] = vv[u];
              s_col[tid + u * 256] = cc[u];
            }
        }
      }
      __syncthreads();
      // ---- phase 3: per-row reduction out of LDS
      if (K == 1) {
        if (tid < nr) {
          const int lo = max(s_rp[tid], ts), hi = min(s_rp[tid + 1], te);
          T s = T(0);
          for (int k = lo; k < hi; ++k) s += s_val[k - ts];
          acc[0][0] += s;
        }
      } else {
        // the lane owns CPL columns of NPASS rows (r = tid/LPR + p*RPP); walk the rows in lock-step, JU nonzeros at a
        // time, so NPASS*JU independent 16-byte gathers are in flight. Per-row summation order (ascending k) is unchanged.
        int lo[NPASS], len[NPASS];
        int maxlen = 0;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          const int r = tid / LPR + p * RPP;
          lo[p] = 0;
          len[p] = 0;
          if (r < nr) {
            const int l = max(s_rp[r], ts), h = min(s_rp[r + 1], te);
            lo[p] = l - ts;
            len[p] = h - l;
          }
          maxlen = max(maxlen, len[p]);
        }
        constexpr int JU = NPASS >= 8 ? 1 : (8 / NPASS);
        for (int j = 0; j < maxlen; j += JU) {
          XV xv[JU][NPASS];
          T vv[JU][NPASS];
#pragma unroll
          for (int u = 0; u < JU; ++u) {
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
              const bool on = j + u < len[p];
              const int i = on ? lo[p] + j + u : 0;
              vv[u][p] = on ? s_val[i] : T(0);
              if (on) {
                const int col = s_col[i];
                xv[u][p] = *reinterpret_cast<const XV*>(a.x + (size_t)col * K + c0);
                if (WANT_SELF && col == row0 + tid / LPR + p * RPP) {
                  xself[p] = xv[u][p];
                  have_self[p] = true;
                }
              } else {
#pragma unroll
                for (int q = 0; q < CPL; ++q) xv[u][p].e[q] = XT(0);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < JU; ++u) {
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
              if (j + u < len[p]) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) acc[p][q] += vv[u][p] * (T)xv[u][p].e[q];
              }
          }
        }
      }
    }
    // ---- phase 4: epilogue (vectorised: CPL adjacent columns per lane)
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int r = K == 1 ? tid : tid / LPR + p * RPP;
      if (K == 1 && p > 0) break;
      if (r < nr) {
        const size_t row = (size_t)(row0 + r);
        const size_t e0 = row * K + c0;
        YV bv, xo, out, dw;
        if (EPI == EPI_RESID || EPI == EPI_JACOBI || EPI == EPI_QADD) bv = *reinterpret_cast<const YV*>(a.b + e0);
        if (EPI == EPI_JACOBI) {
          XV xs;
          if (WANT_SELF && have_self[p])
            xs = xself[p];
          else
            xs = *reinterpret_cast<const XV*>(a.x + e0);
#pragma unroll
          for (int q = 0; q < CPL; ++q) xo.e[q] = (T)xs.e[q];
        }
        if (EPI == EPI_ADD || EPI == EPI_QADD) xo = *reinterpret_cast<const YV*>(a.xadd + e0);
        if (DOT) {
          if (a.dotw) {
            dw = *reinterpret_cast<const YV*>(a.dotw + e0);
          } else {  // dot with x itself
            XV xs;
            if (WANT_SELF && have_self[p])
              xs = xself[p];
            else
              xs = *reinterpret_cast<const XV*>(a.x + e0);
#pragma unroll
            for (int q = 0; q < CPL; ++q) dw.e[q] = (T)xs.e[q];
          }
        }
        const T sc = (EPI == EPI_JACOBI || EPI == EPI_QADD) ? a.omega * a.dinv[row] : T(0);
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          T v = acc[p][q];
          if (EPI == EPI_RESID) v = bv.e[q] - v;
          if (EPI == EPI_JACOBI) v = xo.e[q] + sc * (bv.e[q] - v);
          if (EPI == EPI_ADD) v = xo.e[q] + v;
          if (EPI == EPI_QADD) v = xo.e[q] + sc * bv.e[q] + v;
          out.e[q] = v;
          if (DOT) dot_acc[hh][q] += (double)dw.e[q] * (double)v;
        }
        stream_store(reinterpret_cast<YV*>(a.y + e0), out);
      }
    }
    }  // halves
  }

  if (DOT) {
    // lanes owning the same columns sit LPR apart: reduce over lane bits >= log2(LPR), then across the 4 waves via LDS
    const int lane = tid & 63, w = tid >> 6;
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        double v = dot_acc[hh][q];
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane < LPR) s_red[w * K + hh * KH + lane * CPL + q] = v;
      }
    __syncthreads();
    if (tid < K) a.partials[(size_t)blockIdx.x * K + tid] = s_red[tid] + s_red[K + tid] + s_red[2 * K + tid] + s_red[3 * K + tid];
  }
}
// (A single-wave variant of this kernel -- one 64-lane workgroup per 64 / 32 rows, no block-wide barriers -- was written
// at the end of round 1 and measured in round 2: 2-5 % slower at K = 8 and 16, profiles/r2_wave_spmm_ab.json. Removed.)

// ---- long rows (restriction: R = P^T has ~25 nonzeros per row on rasters, Q^T ~49) --------------------------------
// The kernel above gives every row one lane group for the whole 256-row block; with rows of 25-50 nonzeros a block
// spans several LDS tiles and in each tile only the rows whose nonzeros fall inside it have work (20-40 % of the
// lanes). Here a workgroup owns ROWS (64 / 32) rows -- about one tile of nonzeros -- and SL lane groups share a row:
// slice s takes the row's nonzeros s, s+SL, ... inside the tile, the SL partial sums are combined with wave shuffles
// (fixed order: deterministic). Plain product only (y = A x), no fused dot.
template <class T, int K, int ROWS>
__global__ __launch_bounds__(256) void spmm_longrow_kernel(SpmvArgs<T, T> a) {
  constexpr int TILE = kSpmvTile;
  __shared__ int s_rp[ROWS + 1];
  __shared__ T s_val[TILE];
  __shared__ int s_col[TILE];
  if (a.skip && *a.skip) return;
  const int tid = threadIdx.x;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPL = K < VEC ? K : VEC;
  constexpr int LPR = K / CPL;                                     // lanes covering the K columns
  constexpr int SL = (ROWS * LPR >= 256) ? 1 : 256 / (ROWS * LPR);  // slices per row
  constexpr int LR = LPR * SL;                                     // lanes per row (<= 64: inside one wave)
  constexpr int RPP = 256 / LR;
  constexpr int NPASS = ROWS / RPP;
  static_assert(LR <= 64 && RPP * NPASS == ROWS, "lane layout");
  typedef SpmvVec<T, CPL> XV;
  const int c0 = (tid % LPR) * CPL;
  const int sl = (tid / LPR) % SL;
  const int rloc = tid / LR;
  const int nblocks = (a.nrows + ROWS - 1) / ROWS;
  // same XCD-aware mapping as spmv_kernel: each XCD marches through one contiguous eighth of the row blocks, so the
  // x rows neighbouring aggregates share are re-used out of that XCD's L2
  int rb_first = blockIdx.x, rb_last = nblocks, rb_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, chunk = (nblocks + 7) >> 3;
    rb_first = xcd * chunk + (blockIdx.x >> 3);
    rb_last = min(nblocks, (xcd + 1) * chunk);
    rb_step = gridDim.x >> 3;
  }
  for (int pos = rb_first; pos < rb_last; pos += rb_step) {
    const int rb = a.order_lr ? a.order_lr[pos] : pos;
    const int row0 = rb * ROWS;
    const int nr = min(ROWS, a.nrows - row0);
    __syncthreads();
    for (int t = tid; t <= nr; t += 256) s_rp[t] = a.rowptr[row0 + t];
    __syncthreads();
    const int kbeg = s_rp[0], kend = s_rp[nr];
    T acc[NPASS][CPL];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
      for (int q = 0; q < CPL; ++q) acc[p][q] = T(0);
    for (int ts = kbeg; ts < kend; ts += TILE) {
      const int te = min(kend, ts + TILE);
      if (ts != kbeg) __syncthreads();
      {
        constexpr int U = TILE / 256;
        T vv[U];
        int cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = ts + tid + u * 256;
          vv[u] = k < te ? stream_load(a.val + k) : T(0);
          cc[u] = k < te ? stream_load(a.col + k) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ts + tid + u * 256 < te) {
            s_val[tid + u * 256] = vv[u];
            s_col[tid + u * 256] = cc[u];
          }
      }
      __syncthreads();
      int lo[NPASS], len[NPASS];
      int maxlen = 0;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int r = rloc + p * RPP;
        lo[p] = 0;
        len[p] = 0;
        if (r < nr) {
          const int l = max(s_rp[r], ts), h = min(s_rp[r + 1], te);
          lo[p] = l - ts + sl;
          len[p] = h - l > sl ? (h - l - sl + SL - 1) / SL : 0;  // entries l+sl, l+sl+SL, ... below h
        }
        maxlen = max(maxlen, len[p]);
      }
      constexpr int JU = NPASS >= 8 ? 1 : (8 / NPASS);
      for (int j = 0; j < maxlen; j += JU) {
        XV xv[JU][NPASS];
        T vv[JU][NPASS];
#pragma unroll
        for (int u = 0; u < JU; ++u) {
#pragma unroll
          for (int p = 0; p < NPASS; ++p) {
            const bool on = j + u < len[p];
            const int i = on ? lo[p] + (j + u) * SL : 0;
            vv[u][p] = on ? s_val[i] : T(0);
            if (on) {
              xv[u][p] = *reinterpret_cast<const XV*>(a.x + (size_t)s_col[i] * K + c0);
            } else {
#pragma unroll
              for (int q = 0; q < CPL; ++q) xv[u][p].e[q] = T(0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < JU; ++u)
#pragma unroll
          for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[p][q] += vv[u][p] * xv[u][p].e[q];
      }
    }
    // combine the SL slices of a row (lanes LPR apart inside the row's LR-lane group), then slice 0 writes
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        T v = acc[p][q];
#pragma unroll
        for (int o = LPR; o < LR; o <<= 1) v += __shfl_xor(v, o, 64);
        acc[p][q] = v;
      }
      const int r = rloc + p * RPP;
      if (sl == 0 && r < nr) {
        XV out;
#pragma unroll
        for (int q = 0; q < CPL; ++q) out.e[q] = acc[p][q];
        *reinterpret_cast<XV*>(a.y + (size_t)(row0 + r) * K + c0) = out;
      }
    }
  }
}

// rows per workgroup the long-row kernel uses for a matrix (0: the 256-row kernel handles it)
inline int longrow_rows(long long nnz, int nrows) {
  if (nnz < 16 * (long long)nrows) return 0;
  if (!knobs().longrow) return 0;
  return nnz < 40 * (long long)nrows ? 64 : 32;
}

template <class T, int K>
inline bool spmm_longrow_launch(const SpmvArgs<T, T>& a, hipStream_t st) {
  const int rows = longrow_rows(a.nnz, a.nrows);
  if (rows == 0) return false;
  auto grid = [&](int rows) {
    int g = std::max(1, std::min(std::max(1024, spmv_grid_cap()), ceil_div(a.nrows, rows)));
    if (g >= 64) g &= ~7;  // multiple of 8: XCD-aware mapping active
    return g;
  };
  if (rows == 64)
    hipLaunchKernelGGL((spmm_longrow_kernel<T, K, 64>), dim3(grid(64)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((spmm_longrow_kernel<T, K, 32>), dim3(grid(32)), dim3(256), 0, st, a);
  return true;
}

// grid (= number of dot partials per column) for an nrows-row product; a multiple of 8 once there is enough
// work, so that the kernel's XCD-aware row-block mapping applies.
// Workgroups per product: enough (64 per CU) that the hardware dispatcher, which starts workgroups in blockIdx
// order as slots free up, keeps every XCD's resident set on a contiguous window of its row-block range (that is what
// makes the band re-use hit in L2); capped so the dot-partial arrays stay small. Measured on MI355X, 10000^2 fp64:
// 4096 -> 16384 workgroups: K=1 3.22 -> 2.84 ms, K=8 5.74 -> 5.52 ms; 16384 -> 65536 at K=16: 8.12 -> 7.51 ms (262144:
// 7.79 ms). The partial rows are collapsed to 256 by collapse_partials_kernel before the scalar kernels read them.
inline int spmv_grid_cap() {  // tuning knob (Knobs::spmv_grid_cap): 0 = one workgroup per row block
  return knobs().spmv_grid_cap;
}

// upper bound of spmv_grid() over all K for an nrows-row product (sizes the dot-partial arrays)
inline size_t spmv_grid_upper(int64_t nrows) {
  const int cap = spmv_grid_cap();
  const size_t nb = (size_t)(2 * ((nrows + kSpmvRows - 1) / kSpmvRows) + 8);
  const size_t g = cap > 0 ? std::min<size_t>(nb, (size_t)cap) : nb;
  return g;
}

template <class T, int K>
inline int spmv_grid(int nrows) {
  int nb = ceil_div(nrows, kSpmvRows) * (K >= 16 ? 2 : 1);
  if (nb < 1) nb = 1;
  const int cap = spmv_grid_cap();
  if (cap > 0) {
    if (nb > cap) nb = cap;
    if (nb >= 64) nb &= ~7;
  } else if (nb >= 64) {
    nb = (nb + 7) & ~7;
  }
  return nb;
}

template <class T, int K, int EPI, bool DOT, class XT = T>
inline void spmv_launch_t(const SpmvArgs<T, XT>& a, hipStream_t st) {
  hipLaunchKernelGGL((spmv_kernel<T, K, EPI, DOT, XT>), dim3(spmv_grid<T, K>(a.nrows)), dim3(256), 0, st, a);
}

// CG product of the mixed-precision path: y (T) = A (T) * p (XT), partials of p'Ap
template <class T, int K, class XT>
inline void spmv_launch_cg(const SpmvArgs<T, XT>& a, hipStream_t st) {
  if (a.nrows <= 0) return;
  spmv_launch_t<T, K, EPI_PLAIN, true, XT>(a, st);
}

template <class T, int K>
inline void spmv_launch(const SpmvArgs<T>& a, int epi, bool dot, hipStream_t st) {
  if (a.nrows <= 0) return;
  switch (epi) {
    case EPI_PLAIN:
      if (dot)
        spmv_launch_t<T, K, EPI_PLAIN, true>(a, st);
      else if (!spmm_longrow_launch<T, K>(a, st))
        spmv_launch_t<T, K, EPI_PLAIN, false>(a, st);
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
    case EPI_QADD:
      dot ? spmv_launch_t<T, K, EPI_QADD, true>(a, st) : spmv_launch_t<T, K, EPI_QADD, false>(a, st);
      break;
  }
}

// Plain product with the 4096-nonzero tile (rows of ~14 nonzeros: the [S Q] matrix of the two-product V(1,1) level)
template <class T, int K>
inline void spmv_launch_wide(const SpmvArgs<T>& a, bool dot, hipStream_t st) {
  if (a.nrows <= 0) return;
  const bool wide = !knobs().narrow_tile && a.nnz > 11 * (long long)a.nrows;
  const dim3 g(spmv_grid<T, K>(a.nrows));
  if (wide) {
    if (dot)
      hipLaunchKernelGGL((spmv_kernel<T, K, EPI_PLAIN, true, T, 16>), g, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((spmv_kernel<T, K, EPI_PLAIN, false, T, 16>), g, dim3(256), 0, st, a);
  } else {
    if (dot)
      hipLaunchKernelGGL((spmv_kernel<T, K, EPI_PLAIN, true, T, 10>), g, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((spmv_kernel<T, K, EPI_PLAIN, false, T, 10>), g, dim3(256), 0, st, a);
  }
}

// ---- band-aware traversal order ------------------------------------------------------------------------------
// A raster Laplacian in column-major numbering couples row i with rows i +- 1 and i +- R (R = raster height), so a
// block of 256 consecutive rows reads three bands of x that lie R rows apart. Walking the row blocks in natural
// order re-reads every x row three times (once per band) because the re-use distance (R rows of x) outlives the
// 4 MiB L2. `spmv_block_order` detects the dominant band offset from the matrix itself and orders the row blocks
// "across raster columns": blocks whose first row has the same offset inside the period are consecutive, so blocks
// that share x rows run next to each other (same XCD, same time) and the re-use is served from L2.
// Matrices without a wide band (network graphs, tiny levels) keep the natural order (empty buffer).
__global__ __launch_bounds__(256) void block_mincol_kernel(int nrows, const int* __restrict__ rp,
                                                           const int* __restrict__ ci, int* __restrict__ mincol,
                                                           int rows_per_block) {
  const int nblocks = (nrows + rows_per_block - 1) / rows_per_block;
  for (int b = blockIdx.x * 256 + threadIdx.x; b < nblocks; b += gridDim.x * 256) {
    const int r0 = b * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    int m = 0x7fffffff;
    for (int r = r0; r < r1; ++r)
      if (rp[r] < rp[r + 1]) m = min(m, ci[rp[r]]);  // columns are sorted: first entry is the row minimum
    mincol[b] = m;
  }
}

template <class T>
inline void spmv_block_order(const Csr<T>& A, DBuf& order, hipStream_t st, long long* period_out = nullptr) {
  order.release();
  if (period_out) *period_out = 0;
  if (A.nrows != A.ncols) return;
  const int nblocks = ceil_div(A.nrows, kSpmvRows);
  if (nblocks < 64) return;
  DBuf dmin = dalloc<int>(nblocks);
  hipLaunchKernelGGL(block_mincol_kernel, dim3(grid_for(nblocks)), dim3(256), 0, st, A.nrows, A.rp(), A.ci(), dptr<int>(dmin),
                     kSpmvRows);
  std::vector<int> mc(nblocks);
  CS_HIP(hipMemcpyAsync(mc.data(), dmin.p, (size_t)nblocks * sizeof(int), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  std::vector<long long> off;
  off.reserve(nblocks);
  for (int b = 0; b < nblocks; ++b)
    if (mc[b] != 0x7fffffff) off.push_back((long long)b * kSpmvRows - mc[b]);
  if (off.empty()) return;
  std::nth_element(off.begin(), off.begin() + off.size() / 2, off.end());
  const long long period = off[off.size() / 2];  // ~ R (+1): distance to the far band
  if (period < 4 * kSpmvRows || period > A.nrows / 4) return;
  if (period_out) *period_out = period;
  std::vector<std::pair<int, int>> key(nblocks);
  for (int b = 0; b < nblocks; ++b) key[b] = std::make_pair((int)((((long long)b * kSpmvRows) % period) / kSpmvRows), b);
  std::sort(key.begin(), key.end());
  std::vector<int> ord(nblocks);
  for (int b = 0; b < nblocks; ++b) ord[b] = key[b].second;
  order.alloc((size_t)nblocks * sizeof(int));
  CS_HIP(hipMemcpyAsync(order.p, ord.data(), (size_t)nblocks * sizeof(int), hipMemcpyHostToDevice, st));
  CS_HIP(hipStreamSynchronize(st));
}

// Traversal order for the long-row kernel on a rectangular operator whose COLUMNS are the nodes of a banded (raster)
// level with band period `period` (restriction-type operators: every coarse row gathers a window of fine nodes that
// spans several raster columns). Row blocks whose windows start at the same position inside the period -- the same
// height in neighbouring raster columns -- become consecutive, so the fine rows they share are re-used out of L2
// instead of being fetched once per coarse column they belong to. Empty when the structure is not recognised.
template <class T>
inline void spmv_block_order_rect(const Csr<T>& A, long long period, DBuf& order, hipStream_t st) {
  order.release();
  const int rows = longrow_rows(A.nnz, A.nrows);
  if (rows == 0 || period <= 0) return;
  const int nblocks = ceil_div(A.nrows, rows);
  if (nblocks < 64) return;
  DBuf dmin = dalloc<int>(nblocks);
  hipLaunchKernelGGL(block_mincol_kernel, dim3(grid_for(nblocks)), dim3(256), 0, st, A.nrows, A.rp(), A.ci(), dptr<int>(dmin),
                     rows);
  std::vector<int> mc(nblocks);
  CS_HIP(hipMemcpyAsync(mc.data(), dmin.p, (size_t)nblocks * sizeof(int), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  // height advance between consecutive blocks of one raster column
  std::vector<long long> d;
  d.reserve(nblocks);
  for (int b = 0; b + 1 < nblocks; ++b) {
    if (mc[b] == 0x7fffffff || mc[b + 1] == 0x7fffffff) continue;
    const long long x = (long long)mc[b + 1] - mc[b];
    if (x > 0 && x < period / 4) d.push_back(x);
  }
  if (d.size() < (size_t)nblocks / 2) return;
  std::nth_element(d.begin(), d.begin() + d.size() / 2, d.end());
  const long long step = d[d.size() / 2];
  if (step <= 0 || period / step < 4) return;
  std::vector<std::pair<int, int>> key(nblocks);
  for (int b = 0; b < nblocks; ++b) {
    const long long m = mc[b] == 0x7fffffff ? 0 : mc[b];
    key[b] = std::make_pair((int)((m % period) / step), b);
  }
  std::sort(key.begin(), key.end());
  std::vector<int> ord(nblocks);
  for (int b = 0; b < nblocks; ++b) ord[b] = key[b].second;
  order.alloc((size_t)nblocks * sizeof(int));
  CS_HIP(hipMemcpyAsync(order.p, ord.data(), (size_t)nblocks * sizeof(int), hipMemcpyHostToDevice, st));
  CS_HIP(hipStreamSynchronize(st));
}

// Convenience: y = A x (+ epilogue) for a Csr<T>.
template <class T, class XT = T>
inline SpmvArgs<T, XT> spmv_args(const Csr<T>& A, const XT* x, T* y) {
  SpmvArgs<T, XT> a;
  a.nrows = A.nrows;
  a.nnz = A.nnz;
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
  a.order = nullptr;
  a.skip = nullptr;
  a.order_lr = nullptr;
  return a;
}

}  // namespace csgpu
