// Pull gradient path, variant 2 (pointwise losses; DistMult / TransE / ComplEx with rows of <= 256 floats per half):
// the forward gather walks the triples GROUPED BY RELATION, so the relation row is read once per work unit and the
// relation gradient  d/dp = g * f(s, o)  is accumulated in registers from the very rows the score needed -- the
// backward pull then only has the entity rows left (2 list entries and 2 entity-row reads per triple instead of 3 and 4).
//
//   plan (ids only; built inside the step or ahead of it on any stream)
//   plan_hist_kernel  one pass over the rows: materialises {s, p, o, +-multiplicity}, LDS histogram of p and LDS hash of s / o per
//                     1024 rows, one global atomic per distinct key -> rank of every row in its relation / entity lists
//   plan_scan_kernel  list offsets of the N entity keys (one workgroup per 1024 keys) + chunk map; last workgroup: relation
//                     offsets and the (relation, chunk) unit map
//   rel_fill_kernel   relation lists {s, o, t, multiplicity}
//   step (parameters)
//   fwd_rel_kernel    one wave per (relation, chunk of 64): score, loss, coef[t], partial d/dp row -> slab
//   ent_fill_kernel   entity lists {p | role, other entity, coef, t}
//   pull_kernel       (pull.hip) over entity keys only
//   rel_finish_kernel d/dp rows = sum of the relation's slab rows (chunk order); pull_finish for hub entities
//
// Duplicate rows.  oversample_data (KGEkeras/utils.py:123-147) draws both arrays WITH replacement: at B=2048, ns=100 the
// 204 800 positive rows hold at most 2048 distinct triples and the negatives ~63 % distinct ones.  Identical rows have
// identical scores and identical gradient terms (no dropout), so the step first folds them: an open-addressing hash table
// in HBM (64-bit keys, one CAS per new triple, a plain load for a repeat) yields the distinct rows with their
// multiplicities, the whole pipeline above then runs on {s, p, o, +-multiplicity} rows (count known on the device only,
// grids are sized for the worst case), and the loss / dL/dscore of a row are weighted by its multiplicity.  Same sums as
// TensorFlow's IndexedSlices accumulation up to fp32 summation order.  Off when dropout is on (every row has its own mask).
#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "dropout.hpp"
#include "kge_internal.hpp"
#include "pull.hpp"
#include "fwd_rel.hpp"  // fwd_rel_kernel (shared with step1.hip)

namespace kge {

constexpr int kRelMax = 4096;   // relations handled by the LDS histogram / single-workgroup plan

struct Dedup {
  unsigned long long* keys;  // [cap]  0 = empty, else key + 1
  int32_t* cnt;              // [cap]  multiplicity of the slot's triple
  int32_t* uid;              // [cap]  folded row id of the slot
  int32_t* nu;               // [64]   number of folded rows (keys | cnt are one region, cleared at the start of a step)
  int32_t* slot_of;          // [n]    slot of input row t (-1: ids outside the tables)
  int32_t* frank;            // [n]    rank of input row t among its block's newly found triples (-1: a repeat)
  int32_t* bcnt;             // [blocks] newly found triples per block of 256 input rows
  int4* uniq;                // [n]
  float* uscore;             // [n]    scores of the folded rows (only when the caller asks for scores)
  int64_t cap;
  int64_t clear_bytes;
};

struct Plan2 {
  Dedup dd;
  // relation side
  int32_t* rcnt;    // [R]
  int32_t* roff;    // [R]  list start
  int32_t* ruoff;   // [R]  first unit of the relation
  int32_t* rtotal;  // [4]  {units}
  int2* rumap;      // [max_units] (relation, chunk)
  int32_t* rrank;   // [n]  rank of triple t in its relation list (-1: invalid ids)
  int4* rentries;   // [n]  {s, o, t, 0}
  float* rslab;     // [max_units][d]
  uint32_t* bflags; // chained steps (StepChain): a flag word per forward workgroup
  int max_units;
  // entity side (same meaning as PullPlan, keys = entities only)
  PullPlan ent;
  int32_t* erank;   // [2][n]
  char* det;        // deterministic mode: 12 n int32 of (key, item) pairs + the sort's temporary storage (null otherwise)
};

// Where the step's rows come from: the caller's (pos, neg) arrays, or the materialised rows {s, p, o, +-multiplicity} of the plan
// (multiplicity 0: the row takes no part -- a candidate nobody picked, or ids outside the tables).
struct Src {
  const int32_t* pos; int64_t m_pos;
  const int32_t* neg; int64_t m_neg;
  const int4* uniq;     // null: the caller's arrays
  const int32_t* nu;    // device count of materialised rows
};
__device__ __forceinline__ int64_t src_rows(const Src& x) { return x.uniq ? (int64_t)x.nu[0] : x.m_pos + x.m_neg; }
__device__ __forceinline__ void src_load(const Src& x, int64_t t, int& s, int& p, int& o, int& w) {
  if (x.uniq) {
    const int4 e = x.uniq[t];
    s = e.x; p = e.y; o = e.z; w = e.w;
  } else {
    const int32_t* tr = t < x.m_pos ? x.pos + 3 * t : x.neg + 3 * (t - x.m_pos);
    s = tr[0]; p = tr[1]; o = tr[2]; w = t < x.m_pos ? 1 : -1;
  }
}


__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// One thread per input row: find-or-insert its key.  A slot changes exactly once (0 -> key), so the plain probe load can only
// see 0 (then the CAS decides) or a final key.  The thread whose CAS installs a key owns the new folded row; its id is
// block offset + rank inside the block (ballots + LDS, no global counter): frank[t] = rank or -1, bcnt[block] = owners.
__global__ __launch_bounds__(256) void dedup_insert_kernel(const int32_t* __restrict__ pos, int64_t m_pos, const int32_t* __restrict__ neg,
                                                            int64_t m_neg, int N, int R, unsigned long long* keys, int32_t* cnt,
                                                            unsigned long long mask, int32_t* __restrict__ slot_of,
                                                            int32_t* __restrict__ frank, int32_t* __restrict__ bcnt,
                                                            uint32_t* __restrict__ err) {
  __shared__ int wcnt[4];
  const int64_t n = m_pos + m_neg;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool first = false;
  if (t < n) {
    const bool is_pos = t < m_pos;
    const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
    const int s = tr[0], p = tr[1], o = tr[2];
    if ((unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N) {
      const unsigned long long key = ((((unsigned long long)s * (unsigned)R + (unsigned)p) * (unsigned)N + (unsigned)o) << 1 | (is_pos ? 1ull : 0ull)) + 1ull;
      unsigned long long slot = mix64(key) & mask;
      for (;;) {
        unsigned long long cur = *(volatile unsigned long long*)(keys + slot);
        if (cur == 0ull) {
          cur = atomicCAS(keys + slot, 0ull, key);
          if (cur == 0ull) { first = true; break; }
        }
        if (cur == key) break;
        slot = (slot + 1) & mask;
      }
      atomicAdd(cnt + slot, 1);
      slot_of[t] = (int32_t)slot;
    } else {
      slot_of[t] = -1;
      atomicOr(err, 1u);
    }
  }
  const unsigned long long m = __ballot(first);
  if (lane == 0) wcnt[wave] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wcnt[w];
  if (t < n) frank[t] = first ? before + __popcll(m & ((1ull << lane) - 1ull)) : -1;
  if (threadIdx.x == 0) bcnt[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// Same grid: block b's folded rows start at sum(bcnt[0..b)) (every block sums that prefix itself; block 0 publishes the total).
__global__ __launch_bounds__(256) void dedup_rows_kernel(const int32_t* __restrict__ pos, int64_t m_pos, const int32_t* __restrict__ neg,
                                                          int64_t m_neg, const int32_t* __restrict__ bcnt, const int32_t* __restrict__ frank,
                                                          const int32_t* __restrict__ slot_of, const int32_t* __restrict__ cnt,
                                                          int32_t* __restrict__ uid, int4* __restrict__ uniq, int32_t* __restrict__ nu) {
  __shared__ int red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int upto = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;
  int acc = 0;
  for (int i = threadIdx.x; i < upto; i += 256) acc += bcnt[i];
  for (int d = 32; d; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  const int sum = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) nu[0] = sum;
  const int base = blockIdx.x == 0 ? 0 : sum;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m_pos + m_neg) return;
  const int fr = frank[t];
  if (fr < 0) return;
  const bool is_pos = t < m_pos;
  const int32_t* tr = is_pos ? pos + 3 * t : neg + 3 * (t - m_pos);
  const int sl = slot_of[t];
  const int w = cnt[sl];
  uid[sl] = base + fr;
  uniq[base + fr] = make_int4(tr[0], tr[1], tr[2], is_pos ? w : -w);
}

__global__ __launch_bounds__(256) void dedup_scores_kernel(const int32_t* __restrict__ slot_of, const int32_t* __restrict__ uid,
                                                            const float* __restrict__ uscore, int64_t n, float* __restrict__ scores) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int sl = slot_of[t];
  scores[t] = sl < 0 ? __builtin_nanf("") : uscore[uid[sl]];
}

// One workgroup of 256 threads: exclusive scans over R <= 4096 relations (list offsets, unit offsets) and the unit map.
__device__ __forceinline__ void rel_plan_body(const int32_t* __restrict__ rcnt, int R, int32_t* __restrict__ roff,
                                              int32_t* __restrict__ ruoff, int32_t* __restrict__ rtotal,
                                              int2* __restrict__ rumap, int max_units, int* wa, int* wb) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (R + 255) / 256;  // <= 16
  const int k0 = tid * per, k1 = min(R, k0 + per);
  int a = 0, b = 0;
  for (int k = k0; k < k1; ++k) { const int c = rcnt[k]; a += c; b += rel_units(c); }
  int ia = a, ib = b;
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
  __syncthreads();
  int ra = ia - a, rb = ib - b;
  for (int w = 0; w < wave; ++w) { ra += wa[w]; rb += wb[w]; }
  for (int k = k0; k < k1; ++k) {
    const int c = rcnt[k], nch = rel_units(c);
    roff[k] = ra; ruoff[k] = rb;
    for (int j = 0; j < nch; ++j) if (rb + j < max_units) rumap[rb + j] = make_int2(k, j);
    ra += c; rb += nch;
  }
  if (tid == 255) { rtotal[0] = rb; rtotal[1] = ra; }  // units, list entries
}

__global__ __launch_bounds__(256) void rel_plan_kernel(const int32_t* __restrict__ rcnt, int R, int32_t* __restrict__ roff,
                                                        int32_t* __restrict__ ruoff, int32_t* __restrict__ rtotal,
                                                        int2* __restrict__ rumap, int max_units) {
  __shared__ int wa[4], wb[4];
  rel_plan_body(rcnt, R, roff, ruoff, rtotal, rumap, max_units, wa, wb);
}

// Offsets of both index halves in ONE launch: workgroups [0, etiles) scan the entity keys (scan_tile_body, pull.hpp), the last
// workgroup plans the relations.
__global__ __launch_bounds__(256) void plan_scan_kernel(const int32_t* __restrict__ ecnt, int N, int C, int32_t* __restrict__ off,
                                                         int32_t* __restrict__ eoff, int32_t* __restrict__ poff,
                                                         int32_t* __restrict__ totals, int2* __restrict__ emap, int max_extra, int etiles,
                                                         const int32_t* __restrict__ rcnt, int R, int32_t* __restrict__ roff,
                                                         int32_t* __restrict__ ruoff, int32_t* __restrict__ rtotal,
                                                         int2* __restrict__ rumap, int max_units, const Tri* __restrict__ tsum,
                                                         const Tri* __restrict__ ssum) {
  __shared__ Tri red[8];
  __shared__ int wa[4], wb[4];
  if ((int)blockIdx.x < etiles) scan_tile_body(ecnt, N, C, blockIdx.x, off, eoff, poff, totals, emap, max_extra, red, tsum, ssum);
  else rel_plan_body(rcnt, R, roff, ruoff, rtotal, rumap, max_units, wa, wb);
}

__global__ __launch_bounds__(256) void rel_fill_kernel(Src src, const int32_t* __restrict__ roff, const int32_t* __restrict__ rrank,
                                                        int4* __restrict__ rentries, float* __restrict__ scores,
                                                        uint32_t* __restrict__ err) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= src_rows(src)) return;
  const int r = rrank[t];
  if (r == -2) return;  // a candidate row nobody picked (multiplicity 0)
  if (r < 0) {  // ids outside the tables: the row takes no part in the step, scores NaN, error flag raised
    if (scores) scores[t] = __builtin_nanf("");
    atomicOr(err, 1u);
    return;
  }
  int s, p, o, w;
  src_load(src, t, s, p, o, w);
  rentries[roff[p] + r] = make_int4(s, o, (int)t, w);  // w: +multiplicity for a positive row, -multiplicity for a negative one
}

// The one pass of the plan over the rows of a batch (1024 rows per workgroup, 4 per thread):
//   * where the rows come from -- MODE 0: the plan's materialised rows (the hash fold wrote them); MODE 1: the caller's arrays
//     (copied into the plan when `uniq_out` is given: a plan built ahead must not refer to them afterwards); MODE 2: a batch in
//     folded form, candidate rows + pick counts (the device sampler's by-product): every candidate becomes a row, multiplicity 0
//     where nobody picked it -- no compaction, no atomics on a row counter;
//   * rank of every row inside its relation's list: LDS histogram of the relation column, one global atomic per (workgroup,
//     relation);
//   * rank of every (row, role) inside its destination entity's list.  Entities of real graphs are heavy tailed: a hub would take
//     thousands of same-address returning atomics per batch, which serialise (0.48 ms measured on a Zipf batch of 409 600 rows).
//     Each workgroup therefore first aggregates its rows in an LDS hash table (key -> occurrences so far), then issues ONE global
//     returning atomic per distinct key, and adds the returned base to the local ranks.
// rrank: -1 = ids outside the tables (rel_fill_kernel raises the error flag), -2 = multiplicity 0.
constexpr int kEntRows = 512;    // rows per workgroup (kEntPer per thread): 400 workgroups per 204 800-row batch, so that the bursts of
constexpr int kEntPer = kEntRows / 256;  // returning atomics of one workgroup hide behind the LDS phase of another on the same CU
constexpr int kEntSlots = 2048;  // LDS hash slots (2 * kEntRows keys at most -> load factor <= 0.5)
struct HistSrc {
  Src src;                      // MODE 0 / 1
  const int32_t* cand_rows;     // MODE 2
  const int32_t* cand_cnt;
  int lanes;
  int64_t n_pos_cand, n_all;
};
template <int MODE>
__global__ __launch_bounds__(256) void plan_hist_kernel(HistSrc hs, int64_t n, int N, int R, int32_t* __restrict__ ecnt,
                                                         int32_t* __restrict__ erank, int32_t* __restrict__ rcnt,
                                                         int32_t* __restrict__ rrank, int4* __restrict__ uniq_out,
                                                         int32_t* __restrict__ nu_out, uint32_t* __restrict__ err) {
  __shared__ int lrel[kRelMax];
  __shared__ int hkey[kEntSlots];
  __shared__ int hcnt[kEntSlots];
  for (int i = threadIdx.x; i < R; i += 256) lrel[i] = 0;
  for (int i = threadIdx.x; i < kEntSlots; i += 256) { hkey[i] = -1; hcnt[i] = 0; }
  __syncthreads();
  const int64_t rows = MODE == 2 ? hs.n_all : src_rows(hs.src);
  if (MODE != 0 && nu_out && blockIdx.x == 0 && threadIdx.x == 0) nu_out[0] = (int32_t)rows;
  const int64_t t0 = (int64_t)blockIdx.x * kEntRows;
  int slot[2 * kEntPer], lrank[2 * kEntPer], rp[kEntPer], pp[kEntPer];
#pragma unroll
  for (int i = 0; i < kEntPer; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;  // n = m_pos + m_neg: stride of the two entity rank planes
    slot[2 * i] = slot[2 * i + 1] = -1;
    rp[i] = -2; pp[i] = 0;
    if (t < rows) {
      int s, p, o, w;
      if (MODE == 2) {
        w = 0;
        if (t < hs.n_pos_cand) { for (int l = 0; l < hs.lanes; ++l) w += hs.cand_cnt[l * hs.n_pos_cand + t]; }  // striped counters of the positives
        else w = -hs.cand_cnt[(int64_t)(hs.lanes - 1) * hs.n_pos_cand + t];
        s = hs.cand_rows[3 * t]; p = hs.cand_rows[3 * t + 1]; o = hs.cand_rows[3 * t + 2];
      } else {
        src_load(hs.src, t, s, p, o, w);
      }
      const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
      if (MODE == 2 && w != 0 && !ok) { atomicOr(err, 1u); w = 0; }  // a picked row outside the tables: flagged, takes no part
      if (MODE != 0 && uniq_out) uniq_out[t] = make_int4(s, p, o, w);
      if (w != 0 && !ok) rp[i] = -1;
      if (w != 0 && ok) {
        pp[i] = p;
        rp[i] = atomicAdd(&lrel[p], 1);
#pragma unroll
        for (int role = 0; role < 2; ++role) {
          const int key = role == 0 ? s : o;
          int h = (int)(((unsigned)key * 2654435761u) >> 20) & (kEntSlots - 1);
          for (;;) {
            const int cur = atomicCAS(&hkey[h], -1, key);
            if (cur == -1 || cur == key) break;
            h = (h + 1) & (kEntSlots - 1);
          }
          slot[2 * i + role] = h;
          lrank[2 * i + role] = atomicAdd(&hcnt[h], 1);
        }
      }
    }
  }
  __syncthreads();
  {  // one returning global atomic per distinct key: base of this workgroup's occurrences in the key's list.  All of a thread's
     // atomics are issued before the first result is used (a loop of issue -> wait -> store pays the round trip 16 times in a row).
    constexpr int SW = kEntSlots / 256;
    int keys[SW], vals[SW];
#pragma unroll
    for (int u = 0; u < SW; ++u) { const int i = threadIdx.x + u * 256; keys[u] = hkey[i]; vals[u] = hcnt[i]; }
#pragma unroll
    for (int u = 0; u < SW; ++u) if (keys[u] >= 0) vals[u] = atomicAdd(&ecnt[keys[u]], vals[u]);
#pragma unroll
    for (int u = 0; u < SW; ++u) if (keys[u] >= 0) hcnt[threadIdx.x + u * 256] = vals[u];
  }
  for (int i0 = 0; i0 < R; i0 += 4 * 256) {
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + threadIdx.x + u * 256; v[u] = i < R ? lrel[i] : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (v[u]) v[u] = atomicAdd(&rcnt[i0 + threadIdx.x + u * 256], v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + threadIdx.x + u * 256; if (i < R) lrel[i] = v[u]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kEntPer; ++i) {
    const int64_t t = t0 + threadIdx.x + i * 256;
    if (t < rows) {
      rrank[t] = rp[i] < 0 ? rp[i] : lrel[pp[i]] + rp[i];
      erank[t] = slot[2 * i] >= 0 ? hcnt[slot[2 * i]] + lrank[2 * i] : -1;
      erank[n + t] = slot[2 * i + 1] >= 0 ? hcnt[slot[2 * i + 1]] + lrank[2 * i + 1] : -1;
    }
  }
}

// Entity lists: entry {relation | role bit, other entity, signed multiplicity, row} at off[destination] + rank.  Ids only -- part of the plan; the
// coefficient of a row (dL/dscore, known after the forward) is looked up by pull_kernel through the row number.
__global__ __launch_bounds__(256) void ent_fill_kernel(Src src, int64_t n, const int32_t* __restrict__ off,
                                                        const int32_t* __restrict__ erank, int4* __restrict__ entries) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= src_rows(src)) return;
  const int r0 = erank[t];
  if (r0 < 0) return;
  int s, p, o, w;
  src_load(src, t, s, p, o, w);
  const int cb = w;  // signed multiplicity (+ positive row, - negative row): what the self-contained entity half needs of the row
  entries[off[s] + r0] = make_int4(p, o, cb, (int)t);
  entries[off[o] + erank[n + t]] = make_int4(p | (int)0x80000000, s, cb, (int)t);
}



// Both fills in one launch (a plan built in one piece): one load of the row, one kernel boundary less on the plan stream.
__global__ __launch_bounds__(256) void fill_both_kernel(Src src, int64_t n, const int32_t* __restrict__ roff, const int32_t* __restrict__ rrank,
                                                         int4* __restrict__ rentries, float* __restrict__ scores, uint32_t* __restrict__ err,
                                                         const int32_t* __restrict__ off, const int32_t* __restrict__ erank, int4* __restrict__ entries) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= src_rows(src)) return;
  const int r = rrank[t];
  const int r0 = erank[t];
  if (r == -2 && r0 < 0) return;  // a candidate row nobody picked (multiplicity 0)
  if (r == -1) {  // ids outside the tables: the row takes no part in the step, scores NaN, error flag raised
    if (scores) scores[t] = __builtin_nanf("");
    atomicOr(err, 1u);
  }
  if (r < 0 && r0 < 0) return;
  int s, p, o, w;
  src_load(src, t, s, p, o, w);
  if (r >= 0) rentries[roff[p] + r] = make_int4(s, o, (int)t, w);
  if (r0 >= 0) {
    entries[off[s] + r0] = make_int4(p, o, w, (int)t);
    entries[off[o] + erank[n + t]] = make_int4(p | (int)0x80000000, s, w, (int)t);
  }
}

// gR[p] += sum of the relation's slab rows (one workgroup per relation)
// Also: leaves rcnt all-zero for the next step (the workspace is self-cleaning; it is zeroed once at allocation), and the
// extra last block reduces the forward's loss partials (fixed order) into *loss_out.
// Blocks [0, ent_blocks): the finish of the entity keys (hub rows: slab rows summed in chunk order; counters left zero).
__global__ __launch_bounds__(256) void rel_finish_kernel(float* __restrict__ gR, int R, int d_e, int32_t* __restrict__ rcnt,
                                                          const int32_t* __restrict__ ruoff, const float* __restrict__ rslab,
                                                          const float* __restrict__ loss_partials, int n_partials,
                                                          float* __restrict__ loss_out, int ent_blocks, float* __restrict__ gE, int N,
                                                          int32_t* __restrict__ ecnt, const int32_t* __restrict__ epoff,
                                                          const float* __restrict__ eslab, int C, StepAdvance adv) {
  __shared__ float red[16];
  __shared__ float part[4][512];  // relation rows of this path are at most 512 floats wide (pull2_supported)
  if ((int)blockIdx.x < ent_blocks) {
    pull_finish_wave(blockIdx.x * 4 + (threadIdx.x >> 6), gE, gR, N, N, d_e, ecnt, epoff, eslab, C);
    return;
  }
  const int rb = blockIdx.x - ent_blocks;
  if (blockIdx.x == gridDim.x - 1) {  // loss = sum of block partials
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += loss_partials[i];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) { loss_out[0] = b; step_advance(adv, b); }
    return;
  }
  // one workgroup per relation: the four waves take every fourth slab row (eight loads in flight each) and meet in LDS,
  // summed in wave order -> the result does not depend on timing
  const int p = rb;
  if (p >= R) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = rcnt[p];
  __syncthreads();
  if (threadIdx.x == 0) rcnt[p] = 0;
  const int nch = rel_units(c);
  if (nch == 0) return;
  const float* src = rslab + (int64_t)ruoff[p] * d_e;
  float* dst = gR + (int64_t)p * d_e;
  for (int k = lane; k < d_e; k += 64) {
    float s = 0.f;
    int j = wave;
    for (; j + 28 < nch; j += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(j + 4 * u) * d_e + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < nch; j += 4) s += src[(int64_t)j * d_e + k];
    part[wave][k] = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < d_e; k += 256) dst[k] += (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
}

// ---------------------------------------------------------------------------------------------------------------
static int64_t b256(int64_t x) { return align_up(x, 256); }
int pull_chunk();
int dedup_mode();

static int64_t dedup_cap(int64_t n) {
  int64_t c = 1024;
  while (c < 2 * n) c <<= 1;
  return c;
}
static int64_t dedup_bytes(int64_t n) {
  const int64_t cap = dedup_cap(n);
  return b256(cap * 8) + b256(cap * 4) + 256 + b256(cap * 4) + b256(n * 4) * 2 + b256((n / 256 + 1) * 4) + b256(n * 16) + b256(n * 4);
}
// Folding needs the packed key 2*((s*R + p)*N + o) + side + 1 to fit 64 bits, and pays off only on batches of some size.
bool deterministic_mode();
static bool dedup_applies(const kge_model_desc* md, int64_t n) {
  const int mode = dedup_mode();
  if (mode == 0 || md->dropout > 0.f || (mode == 1 && n < 4096)) return false;
  if (deterministic_mode()) return false;  // (which thread's CAS installs a key, hence the folded rows' numbering, is a race)
  return (long double)md->num_entities * md->num_entities * md->num_relations * 2.0L + 2.0L < 1.8e19L;
}

bool pull2_supported(const kge_model_desc* md, const kge_loss_desc* ld) {
  if (md->model != KGE_DISTMULT && md->model != KGE_TRANSE && md->model != KGE_COMPLEX) return false;
  if (md->model == KGE_TRANSE && md->norm == 5) return false;  // the general p-norm trains on the push path
  if (ld->loss != KGE_POINTWIZE_HINGE && ld->loss != KGE_POINTWIZE_LOGISTIC && ld->loss != KGE_POINTWIZE_SQUARE &&
      ld->loss != KGE_POINTWIZE_CROSS_ENTROPY)
    return false;
  if (md->num_relations > kRelMax) return false;
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  return width % 4 == 0 && md->d_e % 4 == 0 && width / 4 <= 64;
}

// The per-batch state has two parts.  The PLAN depends on the ids only (folded rows, relation lists, entity offsets and
// ranks): it can be built ahead of time, on another stream, while the previous step still runs (kge_batch_plan_build).
// The SCRATCH depends on the parameters (coefficients, list entries carrying them, partial gradient rows).
struct Sizes { int64_t R, N, n, max_units, max_extra, cap; int C; };
static Sizes sizes_of(const kge_model_desc* md, int64_t n) {
  Sizes z;
  z.R = md->num_relations; z.N = md->num_entities; z.n = n;
  z.max_units = n / kRelChunk + z.R + 1;
  z.C = pull_chunk();
  z.max_extra = 2 * n / z.C + 1;
  z.cap = dedup_cap(n);
  return z;
}
static int64_t counters_bytes(const Sizes& z) { return b256(z.N * 4) + b256(z.R * 4); }
static int64_t det_bytes(int64_t n);
static int64_t plan_bytes(const Sizes& z) {
  return det_bytes(z.n) + b256(z.R * 4) * 2 + 256 + b256(z.max_units * 8) + b256(z.n * 4) + b256(z.n * 16) +                 // relation side
         b256(z.N * 4) * 3 + 256 + b256(z.max_extra * 8) + b256(2 * z.n * 4) + b256(2 * z.n * 16) + scan_sums_bytes(z.N) +  // entity side (incl. list entries)
         b256(z.cap * 8) + b256(z.cap * 4) + 256 + b256(z.cap * 4) + b256(z.n * 4) * 2 + b256((z.n / 256 + 1) * 4) + b256(z.n * 16) + 512;
}
static int64_t scratch_bytes(const Sizes& z, int d_e) {
  return b256(z.max_units * d_e * 4) + b256(z.n * 4) * 2 + b256(2 * z.max_extra * d_e * 4) + 512 +
         b256((z.max_units / 4 + 1) * 4);  // chained steps: a flag word per forward workgroup
}

int64_t pull2_workspace_bytes(const kge_model_desc* md, int64_t n) {
  const Sizes z = sizes_of(md, n);
  return plan_bytes(z) + scratch_bytes(z, md->d_e);
}

// counters: [N] entity counters then [R] relation counters, all zero when a build starts (and left zero by the step)
static void carve_plan(void* counters, void* region, const kge_model_desc* md, int64_t n, Plan2* p) {
  const Sizes z = sizes_of(md, n);
  char* q = reinterpret_cast<char*>(((uintptr_t)region + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += b256(bytes); return r; };
  char* hq = reinterpret_cast<char*>(((uintptr_t)counters + 255) & ~(uintptr_t)255);
  p->ent.cnt = (int32_t*)hq;
  p->rcnt = (int32_t*)(hq + b256(z.N * 4));
  Dedup& d = p->dd;
  d.cap = z.cap;
  d.nu = (int32_t*)take(256);                      // nu | keys | cnt first: contiguous with the counters of a plan buffer, one memset
  d.keys = (unsigned long long*)take(d.cap * 8);
  d.cnt = (int32_t*)take(d.cap * 4);
  d.clear_bytes = (char*)d.cnt + b256(d.cap * 4) - (char*)d.keys;
  d.uid = (int32_t*)take(d.cap * 4);
  d.slot_of = (int32_t*)take(n * 4);
  d.frank = (int32_t*)take(n * 4);
  d.bcnt = (int32_t*)take((n / 256 + 1) * 4);
  d.uniq = (int4*)take(n * 16);
  p->roff = (int32_t*)take(z.R * 4);
  p->ruoff = (int32_t*)take(z.R * 4);
  p->rtotal = (int32_t*)take(16);
  p->rumap = (int2*)take(z.max_units * 8);
  p->rrank = (int32_t*)take(n * 4);
  p->rentries = (int4*)take(n * 16);
  p->max_units = (int)z.max_units;
  p->ent.off = (int32_t*)take(z.N * 4);
  p->ent.eoff = (int32_t*)take(z.N * 4);
  p->ent.poff = (int32_t*)take(z.N * 4);
  p->ent.totals = (int32_t*)take(16);
  p->ent.emap = (int2*)take(z.max_extra * 8);
  p->erank = (int32_t*)take(2 * n * 4);
  p->ent.entries = (int4*)take(2 * n * 16);
  p->ent.tsum = take(cdiv(z.N, kScanTile) * (int64_t)sizeof(Tri));
  p->ent.ssum = take(cdiv(cdiv(z.N, kScanTile), kScanGroup) * (int64_t)sizeof(Tri));
  p->ent.ranks = p->erank;
  p->ent.max_extra = (int)z.max_extra;
  p->ent.chunk = z.C;
  p->det = det_bytes(n) ? take(det_bytes(n) - 256) : nullptr;
}

static void carve_scratch(void* region, const kge_model_desc* md, int64_t n, Plan2* p) {
  const Sizes z = sizes_of(md, n);
  char* q = reinterpret_cast<char*>(((uintptr_t)region + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t bytes) { char* r = q; q += b256(bytes); return r; };
  p->rslab = (float*)take(z.max_units * md->d_e * 4);
  p->ent.coef = (float*)take(n * 4);
  p->ent.slab = (float*)take(2 * z.max_extra * md->d_e * 4);
  p->dd.uscore = (float*)take(n * 4);
  p->bflags = (uint32_t*)take((z.max_units / 4 + 1) * 4);
}

// kernels of pull.hip reused on the entity keys
void pull2_entity_plan(const PullPlan& e, int N, hipStream_t st);
int pull2_entity_pull(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const PullPlan& e, uint64_t seed,
                      hipStream_t st, const LossFinish* lf);
// pullself.hip
int pull2_entity_self_step(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                           const PullPlan& e, const StepTail& tail, float lr_t, uint64_t seed, hipStream_t st);
int rel_tail_launch(const StepTail& t, const float* Rel, const LossFinish& lf, hipStream_t st);
int pull2_entity_self_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                            const PullPlan& e, float* gE, uint64_t seed, hipStream_t st);
// step1.hip
int step1_launch(const kge_model_desc* md, const kge_loss_desc* ld, const FwdRelArgs& fa, int rel_blocks, const PullPlan& e, const StepTail& tail,
                 float* loss_partials, float* loss_out, const StepAdvance& adv, hipStream_t st);

template <int MODEL, int NORM, int LOSS>
static void launch_fwd_rel(int nv, bool drop, int grid, hipStream_t st, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg,
                           const kge_model_desc* md, int nchunk, float margin, const Plan2& p, float* scores, float* partials,
                           uint64_t seed) {
  const uint32_t thr = drop_thr(md->dropout);
  const float dscale = 1.f / (1.f - md->dropout);
  const FwdRelArgs fa{E, Rel, m_pos, m_neg, md->d_e, nchunk, md->gamma, md->modulus, margin, p.rcnt, p.roff, p.rtotal, p.rumap, p.rentries, p.ent.coef,
                      scores, p.rslab, seed, thr, dscale};
#define KGE_FR(NVV, DR) hipLaunchKernelGGL((fwd_rel_kernel<MODEL, NORM, LOSS, NVV, DR>), dim3(grid), dim3(256), 0, st, fa, partials)
#define KGE_FR2(NVV) do { if (drop) KGE_FR(NVV, true); else KGE_FR(NVV, false); } while (0)
  if (nv <= 1) KGE_FR2(1);
  else if (nv <= 2) KGE_FR2(2);
  else KGE_FR2(4);
#undef KGE_FR2
#undef KGE_FR
}

template <int MODEL, int NORM>
static void launch_fwd_rel_loss(int loss, int nv, bool drop, int grid, hipStream_t st, const float* E, const float* Rel, int64_t m_pos,
                                int64_t m_neg, const kge_model_desc* md, int nchunk, float margin, const Plan2& p, float* scores,
                                float* partials, uint64_t seed) {
  if (loss == KGE_POINTWIZE_HINGE) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_HINGE>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
  else launch_fwd_rel<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nv, drop, grid, st, E, Rel, m_pos, m_neg, md, nchunk, margin, p, scores, partials, seed);
}

__global__ __launch_bounds__(256) void zero_words_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ---- deterministic list order (kge_set_deterministic(1) / KGE_DETERMINISTIC=1).  By default a row's place in its relation list and in
// its two entity lists is the arrival order of plan_hist_kernel's integer atomics (LDS within a workgroup, global across them), so two
// runs from the same seeds add a row's gradient terms in different orders and agree to fp32 rounding only.  Here the place is the row's
// RANK BY BATCH ROW among the rows of the same key: (key, item) pairs through a stable radix sort (hipCUB), rank = position - first
// position of the key.  Slower (two sorts per plan: the plan stage runs beside the step, off its critical path) and bit-reproducible:
// every list, hence every sum of the step, has one order.
struct DetBufs { int32_t *ekey, *eitem, *ekey_s, *eitem_s, *rkey, *ritem, *rkey_s, *ritem_s; void* temp; size_t temp_bytes; };
constexpr int32_t kDetNone = 0x7fffffff;  // key of a row that takes no part: sorts behind every real key

template <int MODE>
__global__ __launch_bounds__(256) void det_rows_kernel(HistSrc hs, int64_t n, int N, int R, int32_t* __restrict__ ekey, int32_t* __restrict__ eitem,
                                                        int32_t* __restrict__ rkey, int32_t* __restrict__ ritem, int32_t* __restrict__ erank,
                                                        int32_t* __restrict__ rrank, int4* __restrict__ uniq_out, int32_t* __restrict__ nu_out,
                                                        uint32_t* __restrict__ err) {
  const int64_t rows = MODE == 2 ? hs.n_all : src_rows(hs.src);
  if (MODE != 0 && nu_out && blockIdx.x == 0 && threadIdx.x == 0) nu_out[0] = (int32_t)rows;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  int32_t ks = kDetNone, ko = kDetNone, kr = kDetNone, rr = -2;
  if (t < rows) {
    int s, p, o, w;
    if (MODE == 2) {
      w = 0;
      if (t < hs.n_pos_cand) { for (int l = 0; l < hs.lanes; ++l) w += hs.cand_cnt[l * hs.n_pos_cand + t]; }
      else w = -hs.cand_cnt[(int64_t)(hs.lanes - 1) * hs.n_pos_cand + t];
      s = hs.cand_rows[3 * t]; p = hs.cand_rows[3 * t + 1]; o = hs.cand_rows[3 * t + 2];
    } else {
      src_load(hs.src, t, s, p, o, w);
    }
    const bool ok = (unsigned)s < (unsigned)N && (unsigned)p < (unsigned)R && (unsigned)o < (unsigned)N;
    if (MODE == 2 && w != 0 && !ok) { atomicOr(err, 1u); w = 0; }
    if (MODE != 0 && uniq_out) uniq_out[t] = make_int4(s, p, o, w);
    if (w != 0 && !ok) rr = -1;
    if (w != 0 && ok) { ks = s; ko = o; kr = p; }
  }
  ekey[t] = ks; eitem[t] = (int32_t)t;
  ekey[n + t] = ko; eitem[n + t] = (int32_t)(n + t);
  rkey[t] = kr; ritem[t] = (int32_t)t;
  erank[t] = -1; erank[n + t] = -1;
  rrank[t] = rr;
}

// sorted (key, item): rank of an item = its position - the first position of its key (binary search); the key's last item files the count
__global__ __launch_bounds__(256) void det_rank_kernel(const int32_t* __restrict__ key_s, const int32_t* __restrict__ item_s, int64_t n,
                                                        int32_t* __restrict__ rank_of_item, int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t k = key_s[i];
  if (k == kDetNone) return;
  int64_t lo = 0, hi = i;
  while (lo < hi) {  // first index with key_s >= k
    const int64_t mid = (lo + hi) >> 1;
    if (key_s[mid] < k) lo = mid + 1; else hi = mid;
  }
  rank_of_item[item_s[i]] = (int32_t)(i - lo);
  if (i + 1 == n || key_s[i + 1] != k) cnt[k] = (int32_t)(i - lo + 1);
}

static int g_deterministic = -1;
bool deterministic_mode() {
  if (g_deterministic < 0) { const char* v = getenv("KGE_DETERMINISTIC"); g_deterministic = v && atoi(v) != 0 ? 1 : 0; }
  return g_deterministic != 0;
}
void set_deterministic_mode(int on) { g_deterministic = on ? 1 : 0; }

static size_t det_temp_bytes(int64_t n) {
  size_t a = 0, b = 0;
  const int32_t* k = nullptr;
  int32_t* ko = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k, ko, k, ko, (int)(2 * n), 0, 31, (hipStream_t)0);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, ko, k, ko, (int)n, 0, 31, (hipStream_t)0);
  return std::max(a, b) + 256;
}
static int64_t det_bytes(int64_t n) { return deterministic_mode() && 2 * n < 0x7fffffffll ? 12 * b256(n * 4) + (int64_t)b256((int64_t)det_temp_bytes(n)) + 256 : 0; }

// ---- plan: everything that depends on the ids only.  `own_rows`: the plan must not refer to the caller's arrays afterwards
// (kge_batch_plan_build), so rows that are not folded are copied.
struct Counted { const int32_t* rows; const int32_t* cnt; int lanes; int64_t n_pos, n_all; };  // batch in folded form (rows == null: not given)

static int plan_build(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const Plan2& p, bool own_rows,
                      bool clear_counters, int parts, const Counted& cd, Src* src_out, bool* folded, float* scores, uint32_t* err, hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  const int N = md->num_entities, R = md->num_relations;
  const int tgrid = (int)cdiv(n, 256);
  Src src{pos, m_pos, neg, m_neg, nullptr, nullptr};
  const bool fold = !cd.rows && dedup_applies(md, n);
  *folded = fold;
  const Dedup& d = p.dd;
  if (fold || own_rows || cd.rows) { src.uniq = d.uniq; src.nu = d.nu; }
  *src_out = src;
  if (parts & 1) {  // rows (duplicates folded / copied / taken from the candidates) + both histograms
    // a plan buffer keeps its counters right in front of the row counter and the hash table: one memset clears what is needed
    const int64_t cbytes = clear_counters ? (char*)d.keys - (char*)p.ent.cnt : 0;
    // (small clears by kernel: two replayed hipGraphs that both hold memset nodes stall each other's launches on ROCm 7.2)
    if (!fold && clear_counters) hipLaunchKernelGGL(zero_words_kernel, dim3((int)cdiv(cbytes / 4, 256)), dim3(256), 0, st, (int32_t*)p.ent.cnt, cbytes / 4);
    if (fold) {
      ProfScope prof(KGE_PROF_DEDUP, st);
      KGE_HIP_TRY(hipMemsetAsync(clear_counters ? (void*)p.ent.cnt : (void*)d.keys, 0, cbytes + d.clear_bytes, st));
      hipLaunchKernelGGL(dedup_insert_kernel, dim3(tgrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, N, R, d.keys, d.cnt,
                         (unsigned long long)(d.cap - 1), d.slot_of, d.frank, d.bcnt, err);
      hipLaunchKernelGGL(dedup_rows_kernel, dim3(tgrid), dim3(256), 0, st, pos, m_pos, neg, m_neg, d.bcnt, d.frank, d.slot_of, d.cnt, d.uid,
                         d.uniq, d.nu);
    }
    ProfScope prof(KGE_PROF_INDEX, st);
    const int hgrid = (int)cdiv(cd.rows ? cd.n_all : n, kEntRows);
    if (p.det) {  // deterministic list order: ranks by batch row through two stable sorts (see det_rows_kernel)
      DetBufs db;
      char* q = p.det;
      auto tk = [&](int64_t cnt) { int32_t* r = (int32_t*)q; q += b256(cnt * 4); return r; };
      db.ekey = tk(2 * n); db.eitem = tk(2 * n); db.ekey_s = tk(2 * n); db.eitem_s = tk(2 * n);
      db.rkey = tk(n); db.ritem = tk(n); db.rkey_s = tk(n); db.ritem_s = tk(n);
      db.temp = q; db.temp_bytes = det_temp_bytes(n);
      const int dgrid = (int)cdiv(n, 256);
      if (cd.rows) {
        const HistSrc hs{src, cd.rows, cd.cnt, cd.lanes, cd.n_pos, cd.n_all};
        hipLaunchKernelGGL((det_rows_kernel<2>), dim3(dgrid), dim3(256), 0, st, hs, n, N, R, db.ekey, db.eitem, db.rkey, db.ritem, p.erank, p.rrank, d.uniq, d.nu, err);
      } else {
        const HistSrc hs{Src{pos, m_pos, neg, m_neg, nullptr, nullptr}, nullptr, nullptr, 1, 0, 0};
        hipLaunchKernelGGL((det_rows_kernel<1>), dim3(dgrid), dim3(256), 0, st, hs, n, N, R, db.ekey, db.eitem, db.rkey, db.ritem, p.erank, p.rrank, own_rows ? d.uniq : nullptr,
                           own_rows ? d.nu : nullptr, err);
      }
      size_t tb = db.temp_bytes;
      KGE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(db.temp, tb, (const int32_t*)db.ekey, db.ekey_s, (const int32_t*)db.eitem, db.eitem_s, (int)(2 * n), 0, 31, st));
      tb = db.temp_bytes;
      KGE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(db.temp, tb, (const int32_t*)db.rkey, db.rkey_s, (const int32_t*)db.ritem, db.ritem_s, (int)n, 0, 31, st));
      hipLaunchKernelGGL(det_rank_kernel, dim3((int)cdiv(2 * n, 256)), dim3(256), 0, st, db.ekey_s, db.eitem_s, 2 * n, p.erank, p.ent.cnt);
      hipLaunchKernelGGL(det_rank_kernel, dim3(dgrid), dim3(256), 0, st, db.rkey_s, db.ritem_s, n, p.rrank, p.rcnt);
    } else if (cd.rows) {
      const HistSrc hs{src, cd.rows, cd.cnt, cd.lanes, cd.n_pos, cd.n_all};
      hipLaunchKernelGGL((plan_hist_kernel<2>), dim3(hgrid), dim3(256), 0, st, hs, n, N, R, p.ent.cnt, p.erank, p.rcnt, p.rrank, d.uniq, d.nu, err);
    } else if (fold) {
      const HistSrc hs{src, nullptr, nullptr, 1, 0, 0};
      hipLaunchKernelGGL((plan_hist_kernel<0>), dim3(hgrid), dim3(256), 0, st, hs, n, N, R, p.ent.cnt, p.erank, p.rcnt, p.rrank, nullptr, nullptr, err);
    } else {  // the caller's arrays as they are (copied into the plan when it must own its rows)
      const HistSrc hs{Src{pos, m_pos, neg, m_neg, nullptr, nullptr}, nullptr, nullptr, 1, 0, 0};
      hipLaunchKernelGGL((plan_hist_kernel<1>), dim3(hgrid), dim3(256), 0, st, hs, n, N, R, p.ent.cnt, p.erank, p.rcnt, p.rrank, own_rows ? d.uniq : nullptr,
                         own_rows ? d.nu : nullptr, err);
    }
  }
  if (parts & 6) {
    ProfScope prof(KGE_PROF_INDEX, st);
    const int etiles = (int)cdiv(N, kScanTile);
    if ((parts & 6) == 6) {  // offsets of the entity lists and of the relation lists in one launch (a large table: tile sums first)
      const bool big = scan_sums_launch(p.ent.cnt, N, p.ent.chunk, p.ent.tsum, p.ent.ssum, st);
      hipLaunchKernelGGL(plan_scan_kernel, dim3(etiles + 1), dim3(256), 0, st, p.ent.cnt, N, p.ent.chunk, p.ent.off, p.ent.eoff, p.ent.poff, p.ent.totals,
                         p.ent.emap, p.ent.max_extra, etiles, p.rcnt, R, p.roff, p.ruoff, p.rtotal, p.rumap, p.max_units,
                         big ? static_cast<const Tri*>(p.ent.tsum) : nullptr, big ? static_cast<const Tri*>(p.ent.ssum) : nullptr);
    }
    else if (parts & 2) pull2_entity_plan(p.ent, N, st);
    else hipLaunchKernelGGL(rel_plan_kernel, dim3(1), dim3(256), 0, st, p.rcnt, R, p.roff, p.ruoff, p.rtotal, p.rumap, p.max_units);
    if ((parts & 6) == 6) {
      hipLaunchKernelGGL(fill_both_kernel, dim3(tgrid), dim3(256), 0, st, src, n, p.roff, p.rrank, p.rentries, scores, err, p.ent.off, p.erank, p.ent.entries);
    } else {
      if (parts & 4) hipLaunchKernelGGL(rel_fill_kernel, dim3(tgrid), dim3(256), 0, st, src, p.roff, p.rrank, p.rentries, scores, err);
      if (parts & 2) hipLaunchKernelGGL(ent_fill_kernel, dim3(tgrid), dim3(256), 0, st, src, n, p.ent.off, p.erank, p.ent.entries);
    }
  }
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// KGE_GRAD_SELF=0 (A/B timing): kge_train_grads_planned keeps the coefficient-lookup pull
bool grad_halves_supported(const kge_model_desc* md) {
  static const bool on = [] { const char* v = getenv("KGE_GRAD_SELF"); return !v || atoi(v) != 0; }();
  return on && pull_step_supported(md);
}

// ---- step: everything that depends on the parameters.  Consumes the plan (the finish kernels leave its counters zero).
static int plan_exec(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR, int64_t m_pos,
                     int64_t m_neg, const Plan2& p, const Src& src, float* fwd_scores, float* loss_partials, int* n_partials, float* loss_out,
                     uint64_t seed, hipStream_t st, const StepAdvance& adv = StepAdvance{nullptr, nullptr, 0, 0.f, 0.f, 0.f}, OptFinish* finish = nullptr,
                     const StepTail* tail = nullptr, int grad_parts = 3) {
  const int R = md->num_relations;
  const int width = md->model == KGE_COMPLEX ? md->d_e / 2 : md->d_e;
  const int nchunk = width / 4;
  const bool drop = md->dropout > 0.f;
  const int fgrid = (int)cdiv(p.max_units, 4);
  *n_partials = fgrid;
  const bool self = tail != nullptr;
  // both halves in one call: the forward and the entity half share ONE launch (step1.hip; KGE_STEP1=0: launch by launch, for A/B timing)
  static const bool one_launch = [] { const char* v = getenv("KGE_STEP1"); return !v || atoi(v) != 0; }();
  const bool windowed = self && (tail->ent_lo != 0 || tail->ent_hi != md->num_entities || tail->rel_lo != 0 || tail->rel_hi != R);
  if (windowed && !(tail->parts == 3 && one_launch)) return fail(KGE_EUNSUPPORTED, "a row window needs the whole step in one call (parts = 3) on the one-launch path (KGE_STEP1 unset)");
  if (self && tail->ch.timeouts && !(tail->parts == 3 && one_launch)) return fail(KGE_EUNSUPPORTED, "chained steps need the whole step in one call on the one-launch path (KGE_STEP1 unset)");
  if (self && tail->parts == 3 && one_launch) {
    StepTail t = *tail;
    t.roff = p.roff; t.rtotal = p.rtotal; t.ruoff = p.ruoff; t.rslab = p.rslab; t.rcnt = p.rcnt; t.R = R; t.d_r = md->d_r;
    t.ecnt = p.ent.cnt; t.N = md->num_entities;
    if (t.ch.timeouts) {
      if (windowed) return fail(KGE_EUNSUPPORTED, "chained steps take no row window");
      t.ch.bflags = p.bflags;
    }
    const FwdRelArgs fa{E, Rel, m_pos, m_neg, md->d_e, nchunk, md->gamma, md->modulus, ld->margin, p.rcnt, p.roff, p.rtotal, p.rumap, p.rentries, nullptr,
                        nullptr, p.rslab, seed, drop ? drop_thr(md->dropout) : 0u, drop ? 1.f / (1.f - md->dropout) : 1.f};
    return step1_launch(md, ld, fa, fgrid, p.ent, t, loss_partials, loss_out, adv, st);
  }
  // gradient tables only (kge_train_grads_planned[_parts]), halves apart: the entity half forms its own scores (pullself.hip)
  // (under dropout only on request -- parts 1 | 2 --: the whole gradient half keeps the coefficient-lookup pull there, which measures faster)
  const bool split = !self && !finish && grad_halves_supported(md) && (grad_parts != 3 || !drop);
  if (!split && grad_parts != 3) return fail(KGE_EUNSUPPORTED, "gradient halves apart: dropout or row width not covered");
  if (split && (grad_parts & 1)) {
    if (int rc = pull2_entity_self_grads(md, ld, E, Rel, m_pos, m_neg, p.ent, gE, seed, st)) return rc;
    if (!(grad_parts & 2)) return KGE_OK;
  }
  // (a whole step launch by launch -- KGE_STEP1=0 -- with lr_t read from the step state: the entity half first, it keeps lr_t for the
  // closing launch, which moves the state on)
  const bool ent_first = self && tail->parts == 3 && tail->lr_state;
  if (ent_first) {
    StepTail t = *tail;
    if (int rc = pull2_entity_self_step(md, ld, E, Rel, m_pos, m_neg, p.ent, t, t.lr_t, seed, st)) return rc;
  }
  if (!self || (tail->parts & 2)) {
    ProfScope prof(KGE_PROF_FWD, st);
    const int nv = (int)cdiv(nchunk, 16);
    if (md->model == KGE_DISTMULT) launch_fwd_rel_loss<KGE_DISTMULT, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, fwd_scores, loss_partials, seed);
    else if (md->model == KGE_COMPLEX) launch_fwd_rel_loss<KGE_COMPLEX, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, fwd_scores, loss_partials, seed);
    else if (md->norm == 1) launch_fwd_rel_loss<KGE_TRANSE, 1>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, fwd_scores, loss_partials, seed);
    else if (md->norm == 3) launch_fwd_rel_loss<KGE_TRANSE, 3>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, fwd_scores, loss_partials, seed);
    else launch_fwd_rel_loss<KGE_TRANSE, 2>(ld->loss, nv, drop, fgrid, st, E, Rel, m_pos, m_neg, md, nchunk, ld->margin, p, fwd_scores, loss_partials, seed);
  }
  if (self) {
    StepTail t = *tail;
    t.roff = p.roff; t.rtotal = p.rtotal; t.ruoff = p.ruoff; t.rslab = p.rslab; t.rcnt = p.rcnt; t.R = R; t.d_r = md->d_r;
    if (t.parts & 2) {  // relation rows, loss, step state: behind the forward
      const LossFinish lf{loss_partials, *n_partials, loss_out, adv};
      if (int rc = rel_tail_launch(t, Rel, lf, st)) return rc;
    }
    if ((t.parts & 1) && !ent_first)
      if (int rc = pull2_entity_self_step(md, ld, E, Rel, m_pos, m_neg, p.ent, t, t.lr_t, seed, st)) return rc;
    return KGE_OK;
  }
  if (finish) {
    // a whole step in one call: no finish kernel.  The pull launch carries one extra workgroup for the loss (and the step state);
    // the optimizer sweep that follows sums the slab rows and leaves the counters zero (OptFinish, pull.hpp).
    const LossFinish lf{loss_partials, *n_partials, loss_out, adv};
    if (int rc = pull2_entity_pull(md, E, Rel, gE, gR, p.ent, seed, st, &lf)) return rc;
    *finish = OptFinish{p.ent.off, p.ent.totals, p.ent.poff, p.ent.slab, p.ent.cnt, p.ent.chunk, md->num_entities,
                        p.roff, p.rtotal, p.ruoff, p.rslab, p.rcnt, R};
    KGE_HIP_TRY(hipGetLastError());
    return KGE_OK;
  }
  if (!split)
    if (int rc = pull2_entity_pull(md, E, Rel, gE, gR, p.ent, seed, st, nullptr)) return rc;
  const int eb = split ? 0 : (int)cdiv(md->num_entities, 4);  // (split: the entity half summed its hub rows and left its counters zero)
  hipLaunchKernelGGL(rel_finish_kernel, dim3(eb + R + 1), dim3(256), 0, st, gR, R, md->d_e, p.rcnt, p.ruoff, p.rslab, loss_partials,
                     *n_partials, loss_out, eb, gE, md->num_entities, p.ent.cnt, p.ent.poff, p.ent.slab, p.ent.chunk, adv);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// One step with the plan built on the spot (kge_train_step / kge_train_grads).
int train_grads_pull2(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR,
                      const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* scores, float* loss_partials,
                      int* n_partials, float* loss_out, void* hdr, void* ws, uint32_t* err, uint64_t seed, hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  Plan2 p;
  carve_plan(hdr, ws, md, n, &p);
  carve_scratch((char*)ws + plan_bytes(sizes_of(md, n)), md, n, &p);
  Src src;
  bool folded = false;
  // folded rows: the forward writes the scores of the distinct rows, expanded to the caller's rows afterwards
  const bool fold = dedup_applies(md, n);
  float* fwd_scores = fold && scores ? p.dd.uscore : scores;
  if (int rc = plan_build(md, pos, m_pos, neg, m_neg, p, false, false, 7, Counted{nullptr, nullptr, 1, 0, 0}, &src, &folded, fold ? nullptr : scores, err, st)) return rc;
  if (int rc = plan_exec(md, ld, E, Rel, gE, gR, m_pos, m_neg, p, src, fwd_scores, loss_partials, n_partials, loss_out, seed, st)) return rc;
  if (folded && scores) hipLaunchKernelGGL(dedup_scores_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, st, p.dd.slot_of, p.dd.uid, p.dd.uscore, n, scores);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

// ---- plan built ahead of time (kge_batch_plan_build) and consumed by a later step (kge_train_grads_planned)
int64_t batch_plan_bytes(const kge_model_desc* md, int64_t n) {
  const Sizes z = sizes_of(md, n);
  return counters_bytes(z) + plan_bytes(z) + 512;
}

static void carve_plan_buffer(void* plan, const kge_model_desc* md, int64_t n, Plan2* p) {
  char* base = reinterpret_cast<char*>(((uintptr_t)plan + 255) & ~(uintptr_t)255);
  carve_plan(base, base + counters_bytes(sizes_of(md, n)), md, n, p);
}

// parts: bit 0 = rows (fold / copy; clears the counters), bit 1 = entity index, bit 2 = relation index.  Bits 1 and 2 need
// bit 0 done and are independent of each other (the caller may run them on two streams).
int batch_plan_build(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const int32_t* cand_rows,
                     const int32_t* cand_cnt, int pos_lanes, int64_t n_pos_cand, int64_t n_neg_cand, void* plan, int parts, uint32_t* err,
                     hipStream_t st) {
  const int64_t n = m_pos + m_neg;
  Plan2 p;
  carve_plan_buffer(plan, md, n, &p);
  Src src;
  bool folded;
  // parts bit 3: the caller vouches that the plan buffer's counters are all zero (zero-filled at allocation, or left so by the step
  // that consumed the previous build): no clearing launch
  return plan_build(md, pos, m_pos, neg, m_neg, p, true, !(parts & 8), parts & 7, Counted{cand_rows, cand_cnt, pos_lanes, n_pos_cand, n_pos_cand + n_neg_cand}, &src, &folded,
                    nullptr, err, st);
}

int train_grads_planned(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR, void* plan,
                        int64_t m_pos, int64_t m_neg, float* loss_partials, int* n_partials, float* loss_out, void* ws, uint64_t seed,
                        hipStream_t st, const StepAdvance& adv, OptFinish* finish, const StepTail* tail, int grad_parts) {
  const int64_t n = m_pos + m_neg;
  Plan2 p;
  carve_plan_buffer(plan, md, n, &p);
  carve_scratch((char*)ws + plan_bytes(sizes_of(md, n)), md, n, &p);
  const Src src{nullptr, m_pos, nullptr, m_neg, p.dd.uniq, p.dd.nu};
  return plan_exec(md, ld, E, Rel, gE, gR, m_pos, m_neg, p, src, nullptr, loss_partials, n_partials, loss_out, seed, st, adv, finish, tail, grad_parts);
}

}  // namespace kge
