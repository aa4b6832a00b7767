// "pull" gradient path of the training batch: no float atomics.  (Generic-list variant; pull2.hip adds the relation-grouped
// forward for pointwise losses and reuses the scan / pull / finish kernels of this file for the entity rows.)
//
// Same contract as the push path in train.hip (forward x2, loss, sparse gradients with duplicates summed;
// reference: tests.py:63-89, KGEkeras/models.py:134-222, KGEkeras/utils.py:288-319), different dataflow:
//
//   hist_kernel     one returning int atomic per (triple, role): the contribution's rank inside its destination row's list
//                   (relation counters aggregated in LDS per block)
//   scan_kernel     exclusive scans -> list offsets, chunking of long lists, partial-slab slots        (1 workgroup)
//   emap_kernel     (row, chunk) of every extra chunk
//   fwd_kernel      gather 3 rows, score, dL/dscore in registers (pointwise losses), loss partials    [HBM gather]
//   fill_kernel     atomic-free: per (triple, role) one 16-byte list entry {other row a, other row b, coefficient, t}
//                   at  off[destination] + rank
//   pull_kernel     one wave per (destination row, chunk of <= C entries): re-gathers the two OTHER rows of every
//                   contribution, forms the gradient term in registers and sums; single-chunk rows add straight
//                   into the dense gradient table, chunks of long lists (relations, hub entities) write a partial
//                   row to a slab                                                                      [HBM gather]
//   pull_finish     sums the slabs of multi-chunk rows in chunk order; leaves the counters zero for the next step
//
// Every destination row has exactly one writer, so the only non-determinism left is the order of a row's list
// (rank atomics).  Gathers run at the cache/HBM rate (6-10 TB/s) while float atomics are capped near 1.3 TB/s chip-wide
// (MI355X_MICROARCH.md, Global float atomics).
#include "kge_internal.hpp"
#include "pull.hpp"
#include "dropout.hpp"
#include "pull_kernel.hpp"

namespace kge {






// ------------------------------------------------------------------------------------------------ forward
// LOSS >= 0: pointwise loss fused (coef, loss partials);  LOSS < 0: scores only (pairwise two-phase path).
template <int MODEL, int NORM, int LOSS, int G, int VEC, bool DROP>
__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ E, const float* __restrict__ Rel,
                                                   const int32_t* __restrict__ pos, int64_t m_pos,
                                                   const int32_t* __restrict__ neg, int64_t m_neg, int N, int R, int d_e,
                                                   int nchunk, float gamma, float modulus, float margin, float* __restrict__ coef,
                                                   float* __restrict__ scores, float* __restrict__ loss_partials,
                                                   uint32_t* __restrict__ err, uint64_t seed, uint32_t thr, float dscale) {
  using C = V<VEC>;
  using D = DM<VEC>;
  __shared__ float red[16];
  const int lane = threadIdx.x & (G - 1);
  const int64_t group = (int64_t)(blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int64_t ngroups = (int64_t)gridDim.x * blockDim.x / G;
  const int64_t n = m_pos + m_neg;
  const int h = d_e >> 1;
  const float inv_pos = 1.f / (float)m_pos, inv_neg = 1.f / (float)m_neg;
  float lsum = 0.f;
  for (int64_t t = group; t < n; t += ngroups) {
    const bool is_pos = t < m_pos;
    const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int s = tr[0], p = tr[1], o = tr[2];
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    float acc = 0.f;
    if (ok) {
      const float* es = E + (int64_t)s * d_e;
      const float* rp = Rel + (int64_t)p * d_e;
      const float* eo = E + (int64_t)o * d_e;
      DropKey ks{0, 0}, kp{0, 0}, ko{0, 0};
      if (DROP) { ks = drop_key(seed, t, 0); kp = drop_key(seed, t, 1); ko = drop_key(seed, t, 2); }
      const int hc = h / VEC;  // chunk index of the imaginary half
      if (MODEL == KGE_COMPLEX) {
#pragma unroll 2
        for (int c = lane; c < nchunk; c += G) {
          auto sr = C::ld(es + c * VEC), si = C::ld(es + h + c * VEC);
          auto pr = C::ld(rp + c * VEC), pi = C::ld(rp + h + c * VEC);
          auto orr = C::ld(eo + c * VEC), oi = C::ld(eo + h + c * VEC);
          if (DROP) {
            sr = sr * D::mul(ks, c, thr, dscale); si = si * D::mul(ks, hc + c, thr, dscale);
            pr = pr * D::mul(kp, c, thr, dscale); pi = pi * D::mul(kp, hc + c, thr, dscale);
            orr = orr * D::mul(ko, c, thr, dscale); oi = oi * D::mul(ko, hc + c, thr, dscale);
          }
          acc += hsum(sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr);
        }
      } else {
#pragma unroll 4
        for (int c = lane; c < nchunk; c += G) {
          auto sv = C::ld(es + c * VEC), pv = C::ld(rp + c * VEC), ov = C::ld(eo + c * VEC);
          if (DROP) { sv = sv * D::mul(ks, c, thr, dscale); pv = pv * D::mul(kp, c, thr, dscale); ov = ov * D::mul(ko, c, thr, dscale); }
          if (MODEL == KGE_DISTMULT) acc += hsum(sv * pv * ov);
          else { auto x = sv + pv - ov; acc += NORM == 1 ? hsum(vabs(x)) : (NORM == 3 ? hsum(vphase_abs(x)) : hsum(x * x)); }
        }
      }
    }
    acc = group_sum<G>(acc);
    if (lane == 0) {
      float score = acc, dist = acc;
      if (MODEL == KGE_TRANSE) {
        if (NORM == 2) dist = sqrtf(acc);
        if (NORM == 3) dist = modulus * acc;
        score = gamma > 0.f ? gamma - dist : dist;
      }
      if (!ok) { score = __builtin_nanf(""); atomicOr(err, 1u); }
      if (scores) scores[t] = score;
      if (LOSS >= 0) {
        float g = 0.f, l = 0.f;
        if (ok) pointwise_term<LOSS>(score, is_pos, margin, is_pos ? inv_pos : inv_neg, g, l);
        lsum += l;
        if (MODEL == KGE_TRANSE) {
          g = gamma > 0.f ? -g : g;                            // d score / d dist
          if (NORM == 2) g = dist > 0.f ? g / dist : 0.f;      // d dist / d x = x / dist (0 at the origin; TF: NaN)
          if (NORM == 3) g *= 0.5f * modulus;                  // d dist / d x = modulus/2 * phase_dterm(x)
        }
        coef[t] = g;
      }
    }
  }
  if (LOSS >= 0) {
    const float b = block_sum(lsum, red);
    if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
  }
}

// dL/dscore (pairwise path) -> list coefficient, in place
template <int MODEL, int NORM>
__global__ void coef_from_g_kernel(const float* __restrict__ scores, float* __restrict__ g, int64_t n, float gamma, float modulus) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float c = g[t];
  const float sc = scores[t];
  if (!(sc == sc)) c = 0.f;  // invalid row
  if (MODEL == KGE_TRANSE) {
    c = gamma > 0.f ? -c : c;
    if (NORM == 2) {
      const float dist = gamma > 0.f ? gamma - sc : sc;
      c = dist > 0.f ? c / dist : 0.f;
    }
    if (NORM == 3) c *= 0.5f * modulus;
  }
  g[t] = c;
}

// ------------------------------------------------------------------------------------------------ index build
// Relation keys are hot (R addresses receive n increments: ~1700 each at FB15k-237 sizes) and so are the hub entities of a
// real graph (heavy-tailed degrees): same-address returning atomics serialise.  Every workgroup therefore counts its 1024
// triples locally first -- relations in an LDS histogram, entities in an LDS hash table (key -> occurrences so far) -- and
// issues one global returning atomic per distinct key; the returned base plus the local rank is the contribution's rank inside
// its destination list, so the later list fill needs no atomics.
constexpr int kRelLds = 4096;           // relations counted in LDS when R <= kRelLds
constexpr int kIdxBlockTriples = 1024;  // triples per block of the index-build kernel
constexpr int kIdxSlots = 4096;         // LDS hash slots for the 2 * 1024 entity keys of a block

__global__ __launch_bounds__(256) void hist_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                    const int32_t* __restrict__ neg, int64_t m_neg, int N, int R,
                                                    int32_t* __restrict__ cnt, int32_t* __restrict__ ranks /* [3][n] */) {
  __shared__ int lrel[kRelLds];
  __shared__ int hkey[kIdxSlots];
  __shared__ int hcnt[kIdxSlots];
  const bool lds = R <= kRelLds;
  if (lds) for (int i = threadIdx.x; i < R; i += 256) lrel[i] = 0;
  for (int i = threadIdx.x; i < kIdxSlots; i += 256) { hkey[i] = -1; hcnt[i] = 0; }
  __syncthreads();
  const int64_t n = m_pos + m_neg;
  const int64_t t0 = (int64_t)blockIdx.x * kIdxBlockTriples;
  constexpr int PER = kIdxBlockTriples / 256;
  int rs[PER], ro[PER], rp[PER], pp[PER], hs[PER], ho[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;
    rs[i] = ro[i] = rp[i] = pp[i] = hs[i] = ho[i] = -1;
    if (t < n) {
      const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
      const int s = tr[0], p = tr[1], o = tr[2];
      if ((unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N) {
#pragma unroll
        for (int role = 0; role < 2; ++role) {
          const int key = role == 0 ? s : o;
          int h = (int)(((unsigned)key * 2654435761u) >> 20) & (kIdxSlots - 1);
          for (;;) {
            const int cur = atomicCAS(&hkey[h], -1, key);
            if (cur == -1 || cur == key) break;
            h = (h + 1) & (kIdxSlots - 1);
          }
          const int lr = atomicAdd(&hcnt[h], 1);
          if (role == 0) { hs[i] = h; rs[i] = lr; } else { ho[i] = h; ro[i] = lr; }
        }
        pp[i] = p;
        rp[i] = lds ? atomicAdd(&lrel[p], 1) : atomicAdd(&cnt[N + p], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kIdxSlots; i += 256) {
    const int key = hkey[i];
    if (key >= 0) hcnt[i] = atomicAdd(&cnt[key], hcnt[i]);  // base of this block's occurrences in the key's list
  }
  if (lds) for (int i = threadIdx.x; i < R; i += 256) { const int v = lrel[i]; lrel[i] = v ? atomicAdd(&cnt[N + i], v) : 0; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;
    if (t < n) {
      ranks[t] = hs[i] < 0 ? -1 : hcnt[hs[i]] + rs[i];
      ranks[n + t] = ho[i] < 0 ? -1 : hcnt[ho[i]] + ro[i];
      ranks[2 * n + t] = pp[i] < 0 ? -1 : (lds ? lrel[pp[i]] + rp[i] : rp[i]);
    }
  }
}

// Offsets of the per-key lists: one workgroup per tile of 1024 keys (scan_tile_body, pull.hpp); also writes the (key, chunk) map.
__global__ __launch_bounds__(256) void scan_mb_kernel(const int32_t* __restrict__ cnt, int NK, int C, int32_t* __restrict__ off,
                                                       int32_t* __restrict__ eoff, int32_t* __restrict__ poff,
                                                       int32_t* __restrict__ totals, int2* __restrict__ emap, int max_extra,
                                                       const Tri* __restrict__ tsum, const Tri* __restrict__ ssum) {
  __shared__ Tri red[8];
  scan_tile_body(cnt, NK, C, blockIdx.x, off, eoff, poff, totals, emap, max_extra, red, tsum, ssum);
}

// Large tables (more than kScanDirect tiles): the sum of every tile of 1024 counters, then of every group of 1024 tile sums.
__device__ __forceinline__ Tri tri_block_sum(Tri v, Tri* red) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = tri_add(v, Tri{__shfl_xor(v.a, d, 64), __shfl_xor(v.b, d, 64), __shfl_xor(v.c, d, 64)});
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return tri_add(tri_add(red[0], red[1]), tri_add(red[2], red[3]));
}
__global__ __launch_bounds__(256) void scan_tsum_kernel(const int32_t* __restrict__ cnt, int NK, int C, Tri* __restrict__ tsum) {
  __shared__ Tri red[4];
  const int k0 = blockIdx.x * kScanTile + threadIdx.x * 4;
  Tri v{0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k0 + j < NK) v = tri_add(v, tri_of(cnt[k0 + j], C));
  v = tri_block_sum(v, red);
  if (threadIdx.x == 0) tsum[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void scan_ssum_kernel(const Tri* __restrict__ tsum, int tiles, Tri* __restrict__ ssum) {
  __shared__ Tri red[4];
  Tri v{0, 0, 0};
  for (int i = blockIdx.x * kScanGroup + threadIdx.x; i < min(tiles, (int)(blockIdx.x + 1) * kScanGroup); i += 256) v = tri_add(v, tsum[i]);
  v = tri_block_sum(v, red);
  if (threadIdx.x == 0) ssum[blockIdx.x] = v;
}

// Atomic-free list fill: position = list start of the destination + the rank handed out by hist_kernel.
// Entry = {a | kind<<31, b, coefficient bits, t}.
__global__ __launch_bounds__(256) void fill_kernel(const int32_t* __restrict__ pos, int64_t m_pos,
                                                    const int32_t* __restrict__ neg, int64_t m_neg, int N,
                                                    const float* __restrict__ coef, const int32_t* __restrict__ off,
                                                    const int32_t* __restrict__ ranks, int4* __restrict__ entries) {
  const int64_t n = m_pos + m_neg;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int r0 = ranks[t];
  if (r0 < 0) return;  // invalid ids
  const int32_t* tr = t < m_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
  const int s = tr[0], p = tr[1], o = tr[2];
  const int cb = __float_as_int(coef[t]);
  entries[off[s] + r0] = make_int4(p, o, cb, (int)t);                                   // d/ds : needs p, o
  entries[off[o] + ranks[n + t]] = make_int4(p | (int)0x80000000, s, cb, (int)t);       // d/do : needs p, s
  entries[off[N + p] + ranks[2 * n + t]] = make_int4(s, o, cb, (int)t);                 // d/dp : needs s, o
}


__global__ __launch_bounds__(256) void pull_finish_kernel(float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d_e,
                                                           int32_t* __restrict__ cnt, const int32_t* __restrict__ poff,
                                                           const float* __restrict__ slab, int C) {
  pull_finish_wave(blockIdx.x * 4 + (threadIdx.x >> 6), gE, gR, N, NK, d_e, cnt, poff, slab, C);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int pull_chunk();       // api.hip (env KGE_PULL_CHUNK)
int fwd_group_width(int nchunk);

static int64_t a256(int64_t x) { return align_up(x, 256); }

int64_t scan_sums_bytes(int64_t NK) {
  const int64_t tiles = cdiv(NK, kScanTile);
  return a256(tiles * (int64_t)sizeof(Tri)) + a256(cdiv(tiles, kScanGroup) * (int64_t)sizeof(Tri));
}
bool scan_sums_launch(const int32_t* cnt, int NK, int C, void* tsum, void* ssum, hipStream_t st) {
  const int tiles = (int)cdiv(NK, kScanTile);
  if (tiles <= kScanDirect || !tsum || !ssum) return false;
  hipLaunchKernelGGL(scan_tsum_kernel, dim3(tiles), dim3(256), 0, st, cnt, NK, C, static_cast<Tri*>(tsum));
  hipLaunchKernelGGL(scan_ssum_kernel, dim3((int)cdiv(tiles, kScanGroup)), dim3(256), 0, st, static_cast<const Tri*>(tsum), tiles, static_cast<Tri*>(ssum));
  return true;
}

// The per-key counters live in a header at a FIXED place of the caller's workspace (independent of the batch size and
// shared by both pull variants): they are the only state that must be zero on entry, and every step leaves them zero.
int64_t pull_header_bytes(const kge_model_desc* md) { return a256((int64_t)md->num_entities * 4) + a256((int64_t)md->num_relations * 4) + 256; }

int64_t pull_workspace_bytes(const kge_model_desc* md, int64_t n) {
  const int64_t NK = (int64_t)md->num_entities + md->num_relations;
  const int C = pull_chunk();
  const int64_t max_extra = 3 * n / C + 1;
  return a256(NK * 4) * 3 + 256 + a256(max_extra * 8) + a256(3 * n * 16) + a256(3 * n * 4) + a256(n * 4) + a256(2 * max_extra * md->d_e * 4) + scan_sums_bytes(NK) + 1024;
}

static void carve_pull(void* hdr, void* ws, const kge_model_desc* md, int64_t n, PullPlan* p) {
  const int64_t NK = (int64_t)md->num_entities + md->num_relations;
  const int C = pull_chunk();
  const int64_t max_extra = 3 * n / C + 1;
  char* q = reinterpret_cast<char*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += a256(bytes); return r; };
  p->cnt = reinterpret_cast<int32_t*>(((uintptr_t)hdr + 255) & ~(uintptr_t)255);  // entity counters, then relation counters
  p->off = (int32_t*)take(NK * 4);
  p->eoff = (int32_t*)take(NK * 4);
  p->poff = (int32_t*)take(NK * 4);
  p->totals = (int32_t*)take(16);
  p->emap = (int2*)take(max_extra * 8);
  p->entries = (int4*)take(3 * n * 16);
  p->ranks = (int32_t*)take(3 * n * 4);
  p->coef = (float*)take(n * 4);
  p->slab = (float*)take(2 * max_extra * md->d_e * 4);
  p->tsum = take(cdiv(NK, kScanTile) * (int64_t)sizeof(Tri));
  p->ssum = take(cdiv(cdiv(NK, kScanTile), kScanGroup) * (int64_t)sizeof(Tri));
  p->max_extra = (int)max_extra;
  p->chunk = C;
}

bool pull_supported(const kge_model_desc* md) {
  if (md->model == KGE_HOLE) return md->d_e <= 1024;
  if (md->model != KGE_DISTMULT && md->model != KGE_TRANSE && md->model != KGE_COMPLEX) return false;
  if (md->model == KGE_TRANSE && md->norm == 5) return false;  // the general p-norm trains on the push path
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  const bool v4 = width % 4 == 0 && md->d_e % 4 == 0;
  const int nchunk = v4 ? width / 4 : width;
  return nchunk <= 256;
}

template <int MODEL, int NORM, int LOSS, int VEC>
static void launch_fwd(int G, int grid, hipStream_t st, const float* E, const float* Rel, const int32_t* pos, int64_t m_pos,
                       const int32_t* neg, int64_t m_neg, const kge_model_desc* md, int nchunk, float margin, float* coef,
                       float* scores, float* partials, uint32_t* err, uint64_t seed) {
  const bool drop = md->dropout > 0.f;
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
#define KGE_FWD(GW, DR) hipLaunchKernelGGL((fwd_kernel<MODEL, NORM, LOSS, GW, VEC, DR>), dim3(grid), dim3(256), 0, st, E, Rel, pos, m_pos, neg, m_neg, \
                                           md->num_entities, md->num_relations, md->d_e, nchunk, md->gamma, md->modulus, margin, coef, scores, partials, err, seed, thr, dscale)
#define KGE_FWD2(GW) do { if (drop) KGE_FWD(GW, true); else KGE_FWD(GW, false); } while (0)
  switch (G) {
    case 4: KGE_FWD2(4); break;
    case 8: KGE_FWD2(8); break;
    case 16: KGE_FWD2(16); break;
    case 32: KGE_FWD2(32); break;
    default: KGE_FWD2(64); break;
  }
#undef KGE_FWD2
#undef KGE_FWD
}

template <int MODEL, int NORM, int VEC>
static void launch_fwd_loss(int loss, int G, int grid, hipStream_t st, const float* E, const float* Rel, const int32_t* pos,
                            int64_t m_pos, const int32_t* neg, int64_t m_neg, const kge_model_desc* md, int nchunk, float margin,
                            float* coef, float* scores, float* partials, uint32_t* err, uint64_t seed) {
  switch (loss) {
    case KGE_POINTWIZE_HINGE: launch_fwd<MODEL, NORM, KGE_POINTWIZE_HINGE, VEC>(G, grid, st, E, Rel, pos, m_pos, neg, m_neg, md, nchunk, margin, coef, scores, partials, err, seed); break;
    case KGE_POINTWIZE_LOGISTIC: launch_fwd<MODEL, NORM, KGE_POINTWIZE_LOGISTIC, VEC>(G, grid, st, E, Rel, pos, m_pos, neg, m_neg, md, nchunk, margin, coef, scores, partials, err, seed); break;
    case KGE_POINTWIZE_SQUARE: launch_fwd<MODEL, NORM, KGE_POINTWIZE_SQUARE, VEC>(G, grid, st, E, Rel, pos, m_pos, neg, m_neg, md, nchunk, margin, coef, scores, partials, err, seed); break;
    case KGE_POINTWIZE_CROSS_ENTROPY: launch_fwd<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY, VEC>(G, grid, st, E, Rel, pos, m_pos, neg, m_neg, md, nchunk, margin, coef, scores, partials, err, seed); break;
    default: launch_fwd<MODEL, NORM, -1, VEC>(G, grid, st, E, Rel, pos, m_pos, neg, m_neg, md, nchunk, margin, coef, scores, partials, err, seed); break;
  }
}


int pull_prepare(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, void* pull_hdr,
                 void* pull_ws, hipStream_t st, PullPlan* plan) {
  const int64_t n = m_pos + m_neg;
  carve_pull(pull_hdr, pull_ws, md, n, plan);
  PullPlan& p = *plan;
  const int N = md->num_entities, NK = N + md->num_relations;
  const int igrid = (int)cdiv(n, kIdxBlockTriples);
  ProfScope prof(KGE_PROF_INDEX, st);  // index build (ids only)
  hipLaunchKernelGGL(hist_kernel, dim3(igrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, N, md->num_relations, p.cnt, p.ranks);
  const bool big = scan_sums_launch(p.cnt, NK, p.chunk, p.tsum, p.ssum, st);
  hipLaunchKernelGGL(scan_mb_kernel, dim3((int)cdiv(NK, kScanTile)), dim3(256), 0, st, p.cnt, NK, p.chunk, p.off, p.eoff, p.poff, p.totals, p.emap, p.max_extra,
                     big ? static_cast<const Tri*>(p.tsum) : nullptr, big ? static_cast<const Tri*>(p.ssum) : nullptr);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int pull_forward(const kge_model_desc* md, int loss, float margin, const float* E, const float* Rel, const int32_t* pos,
                 int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& p, float* scores, float* loss_partials,
                 int* n_partials, uint32_t* err, uint64_t seed, hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  if (md->model == KGE_HOLE)
    return hole_forward(md, loss, margin, E, Rel, pos, m_pos, neg, m_neg, p.coef, scores, loss_partials, n_partials, err, seed, true, st);
  const Shape sh = shape_of(md, E, Rel, nullptr, nullptr);
  const int G = fwd_group_width(sh.nchunk);
  const int fgrid = (int)std::min<int64_t>(cdiv(n * G, 256), 4096);
  *n_partials = fgrid;
  ProfScope prof(KGE_PROF_FWD, st);
#define KGE_GO(MODEL, NORM)                                                                                                                         \
  do {                                                                                                                                              \
    if (sh.v4) launch_fwd_loss<MODEL, NORM, 4>(loss, G, fgrid, st, E, Rel, pos, m_pos, neg, m_neg, md, sh.nchunk, margin, p.coef, scores, loss_partials, err, seed); \
    else launch_fwd_loss<MODEL, NORM, 1>(loss, G, fgrid, st, E, Rel, pos, m_pos, neg, m_neg, md, sh.nchunk, margin, p.coef, scores, loss_partials, err, seed);       \
  } while (0)
  if (md->model == KGE_DISTMULT) KGE_GO(KGE_DISTMULT, 1);
  else if (md->model == KGE_COMPLEX) KGE_GO(KGE_COMPLEX, 1);
  else if (md->norm == 1) KGE_GO(KGE_TRANSE, 1);
  else if (md->norm == 3) KGE_GO(KGE_TRANSE, 3);
  else KGE_GO(KGE_TRANSE, 2);
#undef KGE_GO
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

int pull_backward(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const int32_t* pos,
                  int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& p, bool g_is_dscore, const float* scores,
                  uint64_t seed, hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  const int N = md->num_entities, NK = N + md->num_relations;
  const Shape sh = shape_of(md, E, Rel, gE, gR);
  const int tgrid = (int)cdiv(n, 256);
  if (g_is_dscore) {
    if (md->model == KGE_TRANSE) {
      if (md->norm == 1) hipLaunchKernelGGL((coef_from_g_kernel<KGE_TRANSE, 1>), dim3(tgrid), dim3(256), 0, st, scores, p.coef, n, md->gamma, md->modulus);
      else if (md->norm == 3) hipLaunchKernelGGL((coef_from_g_kernel<KGE_TRANSE, 3>), dim3(tgrid), dim3(256), 0, st, scores, p.coef, n, md->gamma, md->modulus);
      else hipLaunchKernelGGL((coef_from_g_kernel<KGE_TRANSE, 2>), dim3(tgrid), dim3(256), 0, st, scores, p.coef, n, md->gamma, md->modulus);
    } else {
      hipLaunchKernelGGL((coef_from_g_kernel<KGE_DISTMULT, 1>), dim3(tgrid), dim3(256), 0, st, scores, p.coef, n, md->gamma, md->modulus);
    }
  }
  hipLaunchKernelGGL(fill_kernel, dim3(tgrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, N, p.coef, p.off, p.ranks, p.entries);
  if (md->model == KGE_HOLE) {
    if (int rc = hole_backward(md, E, Rel, gE, gR, p, seed, st)) return rc;
  } else {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    const int units = NK + p.max_extra;
    const int pgrid = (int)cdiv(units, 4);
#define KGE_GO(MODEL, NORM)                                                                \
  do {                                                                                     \
    if (sh.v4) launch_pull<MODEL, NORM, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, p, seed);   \
    else launch_pull<MODEL, NORM, 1>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, p, seed);         \
  } while (0)
    if (md->model == KGE_DISTMULT) KGE_GO(KGE_DISTMULT, 1);
    else if (md->model == KGE_COMPLEX) KGE_GO(KGE_COMPLEX, 1);
    else if (md->norm == 1) KGE_GO(KGE_TRANSE, 1);
    else if (md->norm == 3) KGE_GO(KGE_TRANSE, 3);
    else KGE_GO(KGE_TRANSE, 2);
#undef KGE_GO
  }
  hipLaunchKernelGGL(pull_finish_kernel, dim3((int)cdiv(NK, 4)), dim3(256), 0, st, gE, gR, N, NK, md->d_e, p.cnt, p.poff, p.slab, p.chunk);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// ---- entry points for conve.hip: list entries without coefficients (.w = t, role in bit 31 of .x) and the closing pass
int pull2_fill_entries(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& p,
                       hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  KGE_HIP_TRY(hipMemsetAsync(p.coef, 0, (size_t)n * 4, st));
  hipLaunchKernelGGL(fill_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, st, pos, m_pos, neg, m_neg, md->num_entities, p.coef, p.off, p.ranks, p.entries);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}
int pull_finish_launch(const kge_model_desc* md, float* gE, float* gR, const PullPlan& p, hipStream_t st) {
  const int N = md->num_entities, NK = N + md->num_relations;
  hipLaunchKernelGGL(pull_finish_kernel, dim3((int)cdiv(NK, 4)), dim3(256), 0, st, gE, gR, N, NK, md->d_e, p.cnt, p.poff, p.slab, p.chunk);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// ---- entry points for pull2.hip: the same scan / pull / finish kernels over the N entity keys only
void pull2_entity_plan(const PullPlan& e, int N, hipStream_t st) {
  const bool big = scan_sums_launch(e.cnt, N, e.chunk, e.tsum, e.ssum, st);
  hipLaunchKernelGGL(scan_mb_kernel, dim3((int)cdiv(N, kScanTile)), dim3(256), 0, st, e.cnt, N, e.chunk, e.off, e.eoff, e.poff, e.totals, e.emap, e.max_extra,
                     big ? static_cast<const Tri*>(e.tsum) : nullptr, big ? static_cast<const Tri*>(e.ssum) : nullptr);
}

// (the finish of the entity keys is fused into pull2.hip's rel_finish_kernel)
int pull2_entity_pull(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const PullPlan& e, uint64_t seed,
                      hipStream_t st, const LossFinish* lf) {
  const int N = md->num_entities;
  const Shape sh = shape_of(md, E, Rel, gE, gR);
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    const int pgrid = (int)cdiv(N + e.max_extra, 4);
    if (md->model == KGE_DISTMULT) launch_pull<KGE_DISTMULT, 1, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, e, seed, N, e.coef, lf);
    else if (md->model == KGE_COMPLEX) launch_pull<KGE_COMPLEX, 1, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, e, seed, N, e.coef, lf);
    else if (md->norm == 1) launch_pull<KGE_TRANSE, 1, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, e, seed, N, e.coef, lf);
    else if (md->norm == 3) launch_pull<KGE_TRANSE, 3, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, e, seed, N, e.coef, lf);
    else launch_pull<KGE_TRANSE, 2, 4>(sh.nchunk, pgrid, st, E, Rel, gE, gR, md, e, seed, N, e.coef, lf);
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
