// Inverted dropout on the three gathered rows (KGEkeras/models.py:122,144-148: `self.dp(Embedding(ids))`, active in
// fit, identity in predict).  TensorFlow's mask stream cannot be reproduced, so the mask is a counter-based hash of
// (step seed, row of the batch t, role in {s,p,o}, element index): the forward kernel and both gradient paths
// regenerate identical masks without storing them.  keep-probability = 1 - thr/65536, kept values are scaled by
// 1/(1-rate) exactly like keras.layers.Dropout.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kge {

struct DropKey { uint32_t lo, hi; };

__host__ __device__ __forceinline__ uint32_t drop_thr(float rate) {
  const float t = rate * 65536.f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}

__host__ __device__ __forceinline__ DropKey drop_key(uint64_t seed, int64_t t, int role) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(3 * t + role + 1);  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return DropKey{(uint32_t)z, (uint32_t)(z >> 32)};
}

// 32 mask bits for elements 2w and 2w+1 (16 bits each): a Weyl sequence whose start and (odd) stride are the two halves of
// the row's key -- both already well mixed by splitmix64 -- followed by one multiply-xorshift round.  Six integer operations
// per two elements: the masks are regenerated for every gathered row in the forward AND in the gradient kernels, where a
// murmur3-style finaliser (sixteen operations) made the dropout step ALU bound.
__host__ __device__ __forceinline__ uint32_t drop_word(DropKey k, uint32_t w) {
  uint32_t x = k.lo + w * (k.hi | 1u);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15;
  return x;
}

// multiplier (0 or scale) of element e
__host__ __device__ __forceinline__ float drop_mul(DropKey k, uint32_t e, uint32_t thr, float scale) {
  const uint32_t w = drop_word(k, e >> 1);
  const uint32_t bits = (e & 1u) ? (w >> 16) : (w & 0xFFFFu);
  return bits >= thr ? scale : 0.f;
}

#ifdef __HIPCC__
// multipliers of the float4 chunk c (elements 4c..4c+3)
__device__ __forceinline__ float4 drop_mul4(DropKey k, uint32_t c, uint32_t thr, float scale) {
  const uint32_t w0 = drop_word(k, 2 * c), w1 = drop_word(k, 2 * c + 1);
  return make_float4((w0 & 0xFFFFu) >= thr ? scale : 0.f, (w0 >> 16) >= thr ? scale : 0.f,
                     (w1 & 0xFFFFu) >= thr ? scale : 0.f, (w1 >> 16) >= thr ? scale : 0.f);
}
#endif

}  // namespace kge
