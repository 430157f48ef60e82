// prims.h -- wave64 / workgroup primitives and a device-wide exclusive scan (hand-written, no rocPRIM).
#pragma once
#include "common.h"

namespace csgpu {

// ---- wave64 reductions through cross-lane shuffles (DPP/permute on gfx950)
template <class V>
__device__ __forceinline__ V wave_sum(V v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <class V>
__device__ __forceinline__ V wave_max(V v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    V w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}

// Sum over the 256 threads of a workgroup; result valid in every thread. `sm` needs 4 slots.
template <class V>
__device__ __forceinline__ V block_sum_256(V v, V* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// Inclusive scan across a wave.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// ---------------------------------------------------------------- device-wide exclusive scan (int32)
static const int kScanItems = 8;
static const int kScanTile = kBlock * kScanItems;  // 2048 elements per workgroup

__global__ __launch_bounds__(256) void scan_tiles_kernel(int* __restrict__ data, int64_t n,
                                                         int* __restrict__ tile_sums) {
  __shared__ int s_wave[4];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int local = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    v[j] = i < n ? data[i] : 0;
    local += v[j];
  }
  const int incl = wave_inclusive_scan(local);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int k = 0; k < w; ++k) wave_off += s_wave[k];
  int run = wave_off + incl - local;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    if (i < n) data[i] = run;
    run += v[j];
  }
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = wave_off + incl;
}

__global__ __launch_bounds__(256) void scan_add_kernel(int* __restrict__ data, int64_t n,
                                                       const int* __restrict__ tile_offsets) {
  const int off = tile_offsets[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  for (int j = threadIdx.x; j < kScanTile; j += kBlock) {
    const int64_t i = base + j;
    if (i < n) data[i] += off;
  }
}

// In-place exclusive scan of data[0..n). If total_out != nullptr the grand total is written there (device int).
inline void exclusive_scan_i32(int* data, int64_t n, hipStream_t st, int* total_out_dev = nullptr) {
  if (n <= 0) {
    if (total_out_dev) CS_HIP(hipMemsetAsync(total_out_dev, 0, sizeof(int), st));
    return;
  }
  const int ntiles = ceil_div(n, kScanTile);
  DBuf sums = dalloc<int>((size_t)ntiles + 1);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(ntiles), dim3(kBlock), 0, st, data, n, dptr<int>(sums));
  if (ntiles > 1) {
    exclusive_scan_i32(dptr<int>(sums), ntiles, st, total_out_dev);
    hipLaunchKernelGGL(scan_add_kernel, dim3(ntiles), dim3(kBlock), 0, st, data, n, dptr<int>(sums));
  } else if (total_out_dev) {
    CS_HIP(hipMemcpyAsync(total_out_dev, dptr<int>(sums), sizeof(int), hipMemcpyDeviceToDevice, st));
  }
  check_launch("exclusive_scan_i32");
  CS_HIP(hipStreamSynchronize(st));  // `sums` is freed on return
}

// ---------------------------------------------------------------- small utility kernels
template <class V>
__global__ __launch_bounds__(256) void fill_kernel(V* __restrict__ p, int64_t n, V v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
template <class V>
inline void fill(V* p, int64_t n, V v, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL((fill_kernel<V>), dim3(grid_for(n)), dim3(kBlock), 0, st, p, n, v);
}

__global__ __launch_bounds__(256) void fill_ll_kernel(long long* __restrict__ p, int64_t n, long long v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void fill_int_kernel(int* __restrict__ p, int64_t n, int v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int)i;
}

// (row, column) of the column-major cell id of an R-row raster with ONE division, 32 bits wide when the id allows it (a 64-bit
// division is a ~100-instruction routine on the device; the set-up kernels used to run up to twenty of them per cell)
__device__ __forceinline__ void cell_rc(int64_t id, int R, int& r, int& c) {
  if (id <= 0x7fffffffLL) {
    const unsigned u = (unsigned)id, q = u / (unsigned)R;
    c = (int)q;
    r = (int)(u - q * (unsigned)R);
  } else {
    const int64_t q = id / R;
    c = (int)q;
    r = (int)(id - q * R);
  }
}

// out[j * R + i] = in[i * C + j]: the caller's row-major raster in the column-major order of the node numbering, through a
// 32 x 33 LDS tile (both sides coalesced). The raster kernels then read their cell and its neighbours along raster columns --
// read straight from the row-major array, neighbouring threads are a raster ROW apart (one cache line per value).
template <class T>
__global__ __launch_bounds__(256) void raster_transpose_kernel(int R, int C, const T* __restrict__ in, T* __restrict__ out) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
  const int tiles_j = (C + 31) / 32, tiles_i = (R + 31) / 32;
  for (int64_t t = blockIdx.x; t < (int64_t)tiles_i * tiles_j; t += gridDim.x) {
    const int i0 = (int)(t / tiles_j) * 32, j0 = (int)(t % tiles_j) * 32;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      const int i = i0 + ty + k, j = j0 + tx;
      if (i < R && j < C) tile[ty + k][tx] = in[(size_t)i * C + j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      const int j = j0 + ty + k, i = i0 + tx;
      if (i < R && j < C) out[(size_t)j * R + i] = tile[tx][ty + k];
    }
  }
}

inline int read_int(const int* dev, hipStream_t st) {
  int v = 0;
  CS_HIP(hipMemcpyAsync(&v, dev, sizeof(int), hipMemcpyDeviceToHost, st));
  CS_HIP(hipStreamSynchronize(st));
  return v;
}

}  // namespace csgpu
