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

// pRotatE phase distance (models.py:479-489; TransE kernels with NORM == 3): per element |sin(x/2)| with x = s + p - o.
// d|sin(x/2)|/dx = sign(sin(x/2)) * cos(x/2) / 2 (sign(0) = 0 as in TF); the constant modulus/2 is folded into the
// per-triple coefficient by the forward kernels, phase_dterm() is the element factor.
// Both functions have period pi in y = x/2, so y is reduced to r = y - k*pi in [-pi/2, pi/2] (three-constant Cody-Waite: exact
// products for |k| < 2^13) and |sin r| / sign(r) cos r come from Taylor polynomials of degree 11 / 12 (truncation < 6e-8 at pi/2):
// about 13 VALU instructions instead of the ~40 of sinf / sincosf with their general range reduction.  Arguments beyond
// |y| = 8192 (never seen in an embedding table) take the library functions.
__device__ __forceinline__ float phase_reduce(float y) {
  const float k = rintf(y * 0.318309886183790672f);
  float r = __fmaf_rn(-k, 3.140625f, y);
  r = __fmaf_rn(-k, 9.67502593994140625e-4f, r);
  return __fmaf_rn(-k, 1.509957990978376432e-7f, r);
}
__device__ __forceinline__ float phase_abs(float x) {
  const float y = 0.5f * x;
  if (!(fabsf(y) < 8192.f)) return fabsf(sinf(y));
  const float r = phase_reduce(y), z = r * r;
  float p = __fmaf_rn(z, -2.50521083854417188e-8f, 2.75573192239858907e-6f);  // -1/11!, 1/9!
  p = __fmaf_rn(z, p, -1.98412698412698413e-4f);                              // -1/7!
  p = __fmaf_rn(z, p, 8.33333333333333333e-3f);                               // 1/5!
  p = __fmaf_rn(z, p, -1.66666666666666667e-1f);                              // -1/3!
  return fabsf(__fmaf_rn(r * z, p, r));
}
__device__ __forceinline__ float phase_dterm(float x) {
  const float y = 0.5f * x;
  if (!(fabsf(y) < 8192.f)) {
    float sn, cs;
    sincosf(y, &sn, &cs);
    return sn > 0.f ? cs : (sn < 0.f ? -cs : 0.f);
  }
  const float r = phase_reduce(y), z = r * r;
  float c = __fmaf_rn(z, 2.08767569878680990e-9f, -2.75573192239858907e-7f);  // 1/12!, -1/10!
  c = __fmaf_rn(z, c, 2.48015873015873016e-5f);                               // 1/8!
  c = __fmaf_rn(z, c, -1.38888888888888889e-3f);                              // -1/6!
  c = __fmaf_rn(z, c, 4.16666666666666667e-2f);                               // 1/4!
  c = __fmaf_rn(z, c, -0.5f);
  c = __fmaf_rn(z, c, 1.f);
  return r > 0.f ? c : (r < 0.f ? -c : 0.f);
}

// Device-resident optimizer step state (kge_opt_desc.step_dev, KGE_STEP_STATE_BYTES = 24): {int64 t = updates done, float lr_t of
// update t+1, float unused, double running sum of the step losses (an epoch of 1e5 steps keeps the digits Keras' running mean keeps)}.
__device__ __forceinline__ float* step_state_lr(int64_t* step) { return reinterpret_cast<float*>(step + 1); }
__device__ __forceinline__ double* step_state_loss(int64_t* step) { return reinterpret_cast<double*>(step + 2); }
__device__ __forceinline__ float lr_of_step(int kind, float lr, float b1, float b2, int64_t t) {
  if (kind != KGE_OPT_ADAM_TF) return lr;
  return (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));  // Keras Adam
}
// A whole step in one library call (kge_train_step*, no L3 term): the kernel that closes the gradient half also advances the step
// state -- t += 1, the learning rate of THIS update into *lr_slot (read by the optimizer sweep that follows on the stream), lr_t of
// the next one and the loss sum into the state -- so the step ends with the optimizer sweep, without a one-thread kernel behind it.
struct StepAdvance { int64_t* step; float* lr_slot; int kind; float lr, b1, b2; };
__device__ __forceinline__ void step_advance(const StepAdvance& a, float loss) {
  if (!a.step) return;
  const int64_t t = a.step[0] + 1;
  a.step[0] = t;
  if (a.lr_slot) a.lr_slot[0] = lr_of_step(a.kind, a.lr, a.b1, a.b2, t);  // (null: the launch's waves still read the slot -- step1.hip filled it)
  step_state_lr(a.step)[0] = lr_of_step(a.kind, a.lr, a.b1, a.b2, t + 1);
  step_state_loss(a.step)[0] += (double)loss;
}

// Chained steps (StepChain, pull.hpp): results that another workgroup of the SAME launch consumes are stored write-through (sc1: they
// reach the memory side once acknowledged) and read back with agent-scope loads (which bypass the caches); a flag word per producing
// workgroup says when.
__device__ __forceinline__ void store4_sc1(float* p, float4 v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 load4_agent(const float* p) {
  return make_float4(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                     __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// All 256 threads of a workgroup: wait until flags[b0 .. b1] all carry `tok`.  Bounded: the producers come earlier in the grid and wait
// for nothing, so the token always arrives -- if it does not (a fault), the workgroup carries on and the step reports it (bit 32 of
// the error word).  Returns false when it gave up.
__device__ __forceinline__ bool flags_wait(const uint32_t* flags, int b0, int b1, uint32_t tok) {
  for (int spins = 0;; ++spins) {
    int ok = 1;
    for (int b = b0 + (int)threadIdx.x; b <= b1; b += 256) ok &= __hip_atomic_load(flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tok;
    if (__syncthreads_and(ok)) return true;
    if (spins >= (1 << 16)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}

// One extra workgroup of the pull launch reduces the forward's loss partials (fixed order) and advances the step state, when the
// step has no finish kernel to do it (OptFinish, pull.hpp).
struct LossFinish { const float* partials; int n; float* loss_out; StepAdvance adv; };

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

// dL/dscore (g) and the loss term (l) of one row for the pointwise losses (KGEkeras/utils.py:288-298).
// `inv` = 1/len(true) or 1/len(false) (reduce_mean).  [TF] relu'(0) = 0; clip_by_value passes the gradient on [lo, hi].
constexpr float kLossEps = 1e-6f;  // KGEkeras/utils.py:12
constexpr float kBceEps = 1e-7f;   // keras.backend.epsilon(): binary_crossentropy clips to [eps, 1-eps] and adds eps in the log
template <int LOSS>
__device__ __forceinline__ void pointwise_term(float score, bool is_pos, float margin, float inv, float& g, float& l) {
  if (LOSS == KGE_POINTWIZE_HINGE) {
    const float a = is_pos ? margin - score : margin + score;
    const bool on = a > 0.f;
    l = on ? a * inv : 0.f;
    g = on ? (is_pos ? -inv : inv) : 0.f;
  } else if (LOSS == KGE_POINTWIZE_LOGISTIC) {
    const float e = expf(is_pos ? -score : score);
    const float den = kLossEps + 1.f + e;
    l = logf(den) * inv;
    g = (is_pos ? -e : e) / den * inv;
  } else if (LOSS == KGE_POINTWIZE_CROSS_ENTROPY) {  // utils.py:297-298: BCE(1, true) + BCE(0, false) on raw scores
    const float hi = 1.f - kBceEps;
    const bool in = score >= kBceEps && score <= hi;
    const float c = fminf(fmaxf(score, kBceEps), hi);
    const float a = is_pos ? c + kBceEps : 1.f - c + kBceEps;
    l = -logf(a) * inv;
    g = in ? (is_pos ? -inv : inv) / a : 0.f;
  } else {  // KGE_POINTWIZE_SQUARE
    const float a = is_pos ? margin - score : margin + score;
    l = a * a * inv;
    g = (is_pos ? -2.f : 2.f) * a * inv;
  }
}

}  // namespace kge
