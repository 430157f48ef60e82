// currents.h -- scope row N1: node currents from solved voltages, on the device.
//
// GPU counterpart of the reference's per-pair current post-processing (src/out.jl):
//   _get_branch_currents_posneg   out.jl:250-290   b_e = |g_e| (v_row - v_col) per upper-triangular entry, entries with
//                                                  |b_e / max_e b_e| < 1e-8 zeroed (separately for both orientations)
//   _get_node_currents_posneg     out.jl:186-207   column sums of the positive part of B - B'
//   get_node_currents             out.jl:178-184   node current = max(current into the node, current out of the node)
//   cumulative / maximum maps     out.jl:96-107    cum += map, max = max(max, map)
// For a Laplacian this is SpMV-shaped: in_k = sum_a max(0, g_ak (v_a - v_k)), out_k = sum_a max(0, g_ak (v_k - v_a)),
// each with its own drop threshold 1e-8 * maxcur (maxcur_pos = max_e g_e (v_row - v_col), maxcur_neg = max_e g_e
// (v_col - v_row), e = (row < col)). Voltages are the interleaved solution vectors still resident after the solve.
#pragma once
#include "prims.h"

namespace csgpu {

// pass 1: per-block partial maxima of the signed branch currents in both orientations -> part[block][K][2]
template <class T, int K>
__global__ __launch_bounds__(256) void branch_max_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                         const T* __restrict__ va, const T* __restrict__ x,
                                                         double* __restrict__ part) {
  __shared__ double sm[2][4][K];
  const int c = threadIdx.x % K;
  double mpos = -1e300, mneg = -1e300;
  const int64_t total = (int64_t)n * K;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int row = (int)(it / K);
    const double vr = (double)x[(size_t)row * K + c];
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
      const int col = ci[k];
      if (col > row) {  // each undirected edge once, oriented (row < col) like the reference's upper triangle
        const double g = fabs((double)va[k]);
        const double b = g * (vr - (double)x[(size_t)col * K + c]);
        mpos = b > mpos ? b : mpos;
        mneg = -b > mneg ? -b : mneg;
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= K; o >>= 1) {
    const double a = __shfl_xor(mpos, o, 64), b2 = __shfl_xor(mneg, o, 64);
    mpos = a > mpos ? a : mpos;
    mneg = b2 > mneg ? b2 : mneg;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane < K) {
    sm[0][w][lane] = mpos;
    sm[1][w][lane] = mneg;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int t = threadIdx.x;
    double a = sm[0][0][t], b2 = sm[1][0][t];
    for (int ww = 1; ww < 4; ++ww) {
      a = sm[0][ww][t] > a ? sm[0][ww][t] : a;
      b2 = sm[1][ww][t] > b2 ? sm[1][ww][t] : b2;
    }
    part[((size_t)blockIdx.x * K + t) * 2 + 0] = a;
    part[((size_t)blockIdx.x * K + t) * 2 + 1] = b2;
  }
}

template <int K>
__global__ __launch_bounds__(256) void branch_max_final_kernel(const double* __restrict__ part, int nparts,
                                                               double* __restrict__ maxcur /* [K][2] */) {
  __shared__ double sm[256];
  for (int q = 0; q < 2 * K; ++q) {
    double m = -1e300;
    for (int i = threadIdx.x; i < nparts; i += 256) {
      const double v = part[(size_t)i * 2 * K + q];
      m = v > m ? v : m;
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x + s] > sm[threadIdx.x] ? sm[threadIdx.x + s] : sm[threadIdx.x];
      __syncthreads();
    }
    if (threadIdx.x == 0) maxcur[q] = sm[0];
    __syncthreads();
  }
}

// order-preserving map double -> uint64 (atomicMax on the keys = maximum of the doubles, in any order)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ordered_value(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// Block-diagonal solves (K = 1, many components in one system): the reference post-processes every component
// separately, so the 1e-8 drop threshold refers to the largest branch current OF THAT COMPONENT (out.jl:250-290 under
// advanced.jl:186-271). compmax[2*comp + 0/1] = ordered keys of the maxima in both orientations (pre-set to key(-1e300)).
template <class T>
__global__ __launch_bounds__(256) void branch_max_comp_kernel(int n, const int* __restrict__ rp,
                                                              const int* __restrict__ ci, const T* __restrict__ va,
                                                              const T* __restrict__ x, const int* __restrict__ comp,
                                                              unsigned long long* __restrict__ compmax) {
  for (int row = blockIdx.x * 256 + threadIdx.x; row < n; row += gridDim.x * 256) {
    const double vr = (double)x[row];
    double mpos = -1e300, mneg = -1e300;
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
      const int col = ci[k];
      if (col > row) {
        const double b = fabs((double)va[k]) * (vr - (double)x[col]);
        mpos = b > mpos ? b : mpos;
        mneg = -b > mneg ? -b : mneg;
      }
    }
    if (mpos > -1e300) {
      atomicMax(&compmax[2 * comp[row]], ordered_key(mpos));
      atomicMax(&compmax[2 * comp[row] + 1], ordered_key(mneg));
    }
  }
}

__global__ __launch_bounds__(256) void compmax_init_kernel(int64_t count, unsigned long long* __restrict__ compmax) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    compmax[i] = ordered_key(-1e300);
}

// pass 2: node currents, interleaved like the voltages. comp / compmax (K = 1 only, may be null): per-component maxima
// from branch_max_comp_kernel instead of the per-column maxima in `maxcur`.
template <class T, int K>
__global__ __launch_bounds__(256) void node_current_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                           const T* __restrict__ va, const T* __restrict__ x,
                                                           const double* __restrict__ maxcur, T* __restrict__ curr,
                                                           const T* __restrict__ ground,
                                                           const int* __restrict__ comp,
                                                           const unsigned long long* __restrict__ compmax) {
  const int c = threadIdx.x % K;
  double mp = maxcur ? maxcur[2 * c] : 0.0, mn = maxcur ? maxcur[2 * c + 1] : 0.0;
  const int64_t total = (int64_t)n * K;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int row = (int)(it / K);
    if (comp) {
      mp = ordered_value(compmax[2 * comp[row]]);
      mn = ordered_value(compmax[2 * comp[row] + 1]);
    }
    const double vr = (double)x[(size_t)row * K + c];
    double in = 0.0, out = 0.0;
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
      const int col = ci[k];
      if (col == row) continue;
      const double g = fabs((double)va[k]);
      // signed branch current of the edge in the reference's upper-triangular orientation (smaller index first)
      const double bpos = col > row ? g * (vr - (double)x[(size_t)col * K + c]) : g * ((double)x[(size_t)col * K + c] - vr);
      const double keep_pos = !(fabs(bpos / mp) < 1e-8) ? bpos : 0.0;     // entry of B  (pos orientation)
      const double keep_neg = !(fabs(-bpos / mn) < 1e-8) ? -bpos : 0.0;   // entry of B' (neg orientation)
      // flow from `col` into `row`:  pos orientation contributes g (v_col - v_row), neg orientation g (v_row - v_col)
      const double into = col > row ? -keep_pos : keep_pos;   // g (v_col - v_row), thresholded with maxcur_pos
      const double outof = col > row ? -keep_neg : keep_neg;  // g (v_row - v_col), thresholded with maxcur_neg
      if (into > 0.0) in += into;
      if (outof > 0.0) out += outof;
    }
    if (ground) {
      // advanced modes: the current through the node's own finite ground conductance (out.jl:192-201) -- towards
      // ground when the node sits above it (counts as outflow), from ground otherwise (inflow)
      const double gc = (double)ground[row] * vr;
      if (gc < 0.0) in -= gc;
      if (gc > 0.0) out += gc;
    }
    curr[it] = (T)(in > out ? in : out);
  }
}

// branch currents (network mode, out.jl:209-248 with pos = true, then abs): for every stored entry (row < col) the value
// |g (v_row - v_col)| with the 1e-8 * maxcur_pos drop threshold, 0 at every other position; layout [nnz][K]
template <class T, int K>
__global__ __launch_bounds__(256) void branch_current_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                             const T* __restrict__ va, const T* __restrict__ x,
                                                             const double* __restrict__ maxcur, T* __restrict__ out) {
  const int c = threadIdx.x % K;
  const double mp = maxcur[2 * c];
  const int64_t total = (int64_t)n * K;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int row = (int)(it / K);
    const double vr = (double)x[(size_t)row * K + c];
    for (int k = rp[row]; k < rp[row + 1]; ++k) {
      const int col = ci[k];
      double b = 0.0;
      if (col > row) {
        b = fabs((double)va[k]) * (vr - (double)x[(size_t)col * K + c]);
        if (fabs(b / mp) < 1e-8) b = 0.0;
      }
      out[(size_t)k * K + c] = (T)fabs(b);
    }
  }
}

// cum[row] += sum_c weight[c] * curr[row, c];  mx[row] = max(mx[row], max_c curr[row, c])   (columns c < ncols)
template <class T, int K>
__global__ __launch_bounds__(256) void current_accumulate_kernel(int n, const T* __restrict__ curr, int ncols,
                                                                 const int* __restrict__ weight, T* __restrict__ cum,
                                                                 T* __restrict__ mx) {
  for (int row = blockIdx.x * 256 + threadIdx.x; row < n; row += gridDim.x * 256) {
    T s = cum ? cum[row] : T(0);
    T m = mx ? mx[row] : T(0);
    for (int c = 0; c < ncols; ++c) {
      const T v = curr[(size_t)row * K + c];
      const int w = weight[c];
      for (int r = 0; r < w; ++r) s += v;  // repeated addition == the reference adding the same map w times
      if (w > 0 && v > m) m = v;
    }
    if (cum) cum[row] = s;
    if (mx) mx[row] = m;
  }
}

}  // namespace csgpu
