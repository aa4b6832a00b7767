// Host-side bit-exact restatement of the random streams the reference batch builder consumes.
//
// Reference call sites (relative to the reference repo):
//   np.random.randint(0, N, size=(ns*B, 1)) x2          KGEkeras/utils.py:118-119  (fs first, then fo)
//   np.random.shuffle(self.index)                         tests.py:50-53
//   random.choices(kg, k=m)                               KGEkeras/utils.py:136
// Both NumPy's legacy global RandomState and CPython's `random` are MT19937 (Matsumoto & Nishimura);
// they differ in seeding (init_genrand vs init_by_array) and in how outputs are turned into integers
// (masked rejection on one 32-bit word vs floor(53-bit double * n) on two words).
#include <cmath>
#include <cstring>
#include <vector>

#include "kge_host.hpp"

namespace {

struct MT {
  uint32_t mt[624];
  int32_t pos;
  int32_t pad;
};
static_assert(sizeof(MT) == KGE_MT_STATE_BYTES, "state blob size");

inline void init_genrand(MT* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->pos = 624;
  s->pad = 0;
}

inline void init_by_array(MT* s, const uint32_t* key, int klen) {
  init_genrand(s, 19650218u);
  int i = 1, j = 0;
  for (int k = (624 > klen ? 624 : klen); k; --k) {
    s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    if (++i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    if (++j >= klen) j = 0;
  }
  for (int k = 623; k; --k) {
    s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    if (++i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
  }
  s->mt[0] = 0x80000000u;
  s->pos = 624;
}

inline void regenerate(MT* s) {
  uint32_t* mt = s->mt;
  auto tw = [](uint32_t u, uint32_t v) -> uint32_t {
    uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  };
  int k = 0;
  for (; k < 624 - 397; ++k) mt[k] = mt[k + 397] ^ tw(mt[k], mt[k + 1]);
  for (; k < 623; ++k) mt[k] = mt[k + (397 - 624)] ^ tw(mt[k], mt[k + 1]);
  mt[623] = mt[396] ^ tw(mt[623], mt[0]);
  s->pos = 0;
}

inline uint32_t next32(MT* s) {
  if (s->pos >= 624) regenerate(s);
  uint32_t y = s->mt[s->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

inline uint64_t smear(uint64_t v) {
  v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32;
  return v;
}

// NumPy legacy bounded draw in [0, rng] (rng < 2^32): masked rejection on single 32-bit outputs.
inline uint32_t bounded32(MT* s, uint32_t rng, uint32_t mask) {
  uint32_t v;
  while ((v = (next32(s) & mask)) > rng) {}
  return v;
}

// CPython random.random(): 53-bit double from two outputs.
inline double py_random(MT* s) {
  uint32_t a = next32(s) >> 5, b = next32(s) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

}  // namespace

extern "C" {

int kge_mt_seed_numpy(void* st, uint32_t seed) {
  if (!st) return kge::fail(KGE_EINVAL, "kge_mt_seed_numpy: null state");
  init_genrand(static_cast<MT*>(st), seed);
  return KGE_OK;
}

int kge_mt_seed_python(void* st, const uint32_t* key, int n) {
  if (!st || !key || n < 1) return kge::fail(KGE_EINVAL, "kge_mt_seed_python: null state/key or empty key");
  init_by_array(static_cast<MT*>(st), key, n);
  return KGE_OK;
}

int kge_mt_raw(void* st, uint32_t* out, int64_t n) {
  if (!st || (!out && n) || n < 0) return kge::fail(KGE_EINVAL, "kge_mt_raw: bad arguments");
  MT* s = static_cast<MT*>(st);
  for (int64_t i = 0; i < n; ++i) out[i] = next32(s);
  return KGE_OK;
}

int kge_mt_randint(void* st, int64_t low, int64_t high, int64_t* out, int64_t n) {
  if (!st || (!out && n) || n < 0) return kge::fail(KGE_EINVAL, "kge_mt_randint: bad arguments");
  if (high <= low) return kge::fail(KGE_EINVAL, "kge_mt_randint: low >= high (numpy raises ValueError)");
  uint64_t rng = (uint64_t)(high - 1 - low);
  if (rng > 0xFFFFFFFFull) return kge::fail(KGE_EUNSUPPORTED, "kge_mt_randint: ranges above 2^32 are not on the path");
  MT* s = static_cast<MT*>(st);
  if (rng == 0) { for (int64_t i = 0; i < n; ++i) out[i] = low; return KGE_OK; }  // numpy draws nothing
  if (rng == 0xFFFFFFFFull) { for (int64_t i = 0; i < n; ++i) out[i] = low + (int64_t)next32(s); return KGE_OK; }
  uint32_t mask = (uint32_t)smear(rng);
  for (int64_t i = 0; i < n; ++i) out[i] = low + (int64_t)bounded32(s, (uint32_t)rng, mask);
  return KGE_OK;
}

int kge_mt_shuffle(void* st, int64_t* perm, int64_t n) {
  if (!st || (!perm && n) || n < 0) return kge::fail(KGE_EINVAL, "kge_mt_shuffle: bad arguments");
  if (n > 0xFFFFFFFFll) return kge::fail(KGE_EUNSUPPORTED, "kge_mt_shuffle: n above 2^32");
  MT* s = static_cast<MT*>(st);
  for (int64_t i = n - 1; i >= 1; --i) {  // Fisher-Yates, j = random_interval(i)
    uint32_t mask = (uint32_t)smear((uint64_t)i);
    int64_t j = bounded32(s, (uint32_t)i, mask);
    int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
  return KGE_OK;
}

int kge_mt_choices(void* st, int64_t population, int64_t* idx, int64_t k) {
  if (!st || (!idx && k) || k < 0) return kge::fail(KGE_EINVAL, "kge_mt_choices: bad arguments");
  if (population <= 0) return kge::fail(KGE_EINVAL, "kge_mt_choices: empty population (random.choices raises IndexError)");
  MT* s = static_cast<MT*>(st);
  const double n = (double)population;
  for (int64_t i = 0; i < k; ++i) idx[i] = (int64_t)std::floor(py_random(s) * n);
  return KGE_OK;
}

int kge_build_batch_host(void* np_state, void* py_state, const int32_t* batch, int64_t B, int64_t ns, int64_t num_e,
                         int32_t* pos_out, int32_t* neg_out) {
  if (!np_state || !py_state || !batch || !pos_out || !neg_out || B <= 0 || ns <= 0 || num_e <= 0)
    return kge::fail(KGE_EINVAL, "kge_build_batch_host: bad arguments");
  if (num_e > 0x7FFFFFFFll || B * ns > 0x7FFFFFFFll) return kge::fail(KGE_EUNSUPPORTED, "kge_build_batch_host: size above int32");
  MT* nps = static_cast<MT*>(np_state);
  MT* pys = static_cast<MT*>(py_state);
  const int64_t m = B * ns;
  std::vector<int32_t> fs(m), fo(m);
  const uint32_t rng = (uint32_t)(num_e - 1);
  const uint32_t mask = (uint32_t)smear(rng);
  // utils.py:118-119: all of fs, then all of fo, from one continuous stream
  for (int64_t j = 0; j < m; ++j) fs[j] = rng ? (int32_t)bounded32(nps, rng, mask) : 0;
  for (int64_t j = 0; j < m; ++j) fo[j] = rng ? (int32_t)bounded32(nps, rng, mask) : 0;
  // utils.py:136: choices over the positives (len B), then over the negatives (len m)
  const double nB = (double)B, nM = (double)m;
  for (int64_t i = 0; i < m; ++i) {
    int64_t a = (int64_t)std::floor(py_random(pys) * nB);
    pos_out[3 * i + 0] = batch[3 * a + 0];
    pos_out[3 * i + 1] = batch[3 * a + 1];
    pos_out[3 * i + 2] = batch[3 * a + 2];
  }
  for (int64_t i = 0; i < m; ++i) {
    int64_t j = (int64_t)std::floor(py_random(pys) * nM);
    neg_out[3 * i + 0] = fs[j];
    neg_out[3 * i + 1] = batch[3 * (j / ns) + 1];  // utils.py:117 np.repeat(p, ns): row j carries relation of positive j // ns
    neg_out[3 * i + 2] = fo[j];
  }
  return KGE_OK;
}

}  // extern "C"
