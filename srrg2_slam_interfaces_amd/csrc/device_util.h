// device_util.h -- small device helpers shared by kernels_prep.hip (ingest, grid, sort) and kernels.hip (ICP passes).
#pragma once
#include <hip/hip_runtime.h>

namespace {

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// monotone float -> unsigned key (for atomicMin/atomicMax on floats of either sign)
__device__ __forceinline__ unsigned fkey(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h) {
  float u = (x - o) * inv_h;
  u       = fminf(fmaxf(u, -2048.f), 4096.f);
  return (int) floorf(u);
}

__device__ __forceinline__ float bound2_of(int r, float h) {
  float b = ((float) r - 0.01f) * h;
  return (b * b) * 0.9999f;
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

}  // namespace
