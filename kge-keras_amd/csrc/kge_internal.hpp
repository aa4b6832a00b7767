// Internal helpers shared by the HIP translation units of libkge_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "kge_host.hpp"

namespace kge {

#define KGE_HIP_TRY(expr)                                                                         \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) return ::kge::fail(KGE_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

// kernel timing hooks (api.hip); no-ops unless kge_prof_enable(1)
void prof_begin(int label, hipStream_t st);
void prof_end(int label, hipStream_t st);
struct ProfScope {
  int label; hipStream_t st;
  ProfScope(int l, hipStream_t s) : label(l), st(s) { prof_begin(l, s); }
  ~ProfScope() { prof_end(label, st); }
};

constexpr int kWave = 64;  // gfx950 wavefront

// number of CUs on MI355X; grid sizing only (no correctness dependence)
constexpr int kCUs = 256;

}  // namespace kge

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
namespace kge {

// Sum over the G (power of two, <= 64) consecutive lanes of a sub-group; every lane gets the total.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum of one float per thread (blockDim.x multiple of 64, <= 1024); result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
  v = group_sum<64>(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];  // fixed order: deterministic
  }
  __syncthreads();
  return r;
}

}  // namespace kge
