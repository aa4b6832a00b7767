// Shared between pull.hip and train.hip: the per-batch plan of the pull gradient path (all device pointers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/kge_hip.h"

namespace kge {

// rows per forward work unit of the relation-grouped path (pull2.hip): 32, or 128 for a relation with more than 4096 rows in the batch,
// so that a dominant relation (real graphs: the top relation of FB15k-237 carries 16 % of the triples) leaves a few hundred partial
// gradient rows to sum, not thousands
#ifndef KGE_REL_CHUNK
#define KGE_REL_CHUNK 64
#endif
constexpr int kRelChunk = KGE_REL_CHUNK;
__host__ __device__ __forceinline__ int rel_chunk_rows(int c) { return c > 4096 ? 4 * kRelChunk : kRelChunk; }
__host__ __device__ __forceinline__ int rel_units(int c) { const int ch = rel_chunk_rows(c); return (c + ch - 1) / ch; }

// A whole step in one library call on the relation-grouped path: the optimizer sweep closes the gradient half itself instead of
// reading finished gradient tables -- a hub entity row (more than C list entries) is the sum of its slab rows, every relation row
// the sum of its units' partial rows -- and leaves the plan's counters zero; the loss partials are reduced (and the step state
// advanced) by one extra workgroup of the pull launch.  That takes the finish kernel (10 us + a launch) off the step's chain.
struct OptFinish {
  const int32_t* eoff;     // [N] starts of the entity lists (length of row r: eoff[r+1] - eoff[r]; last row: etotals[2] - eoff[N-1])
  const int32_t* etotals;
  const int32_t* epoff;    // [N] first slab slot of a hub row
  const float* eslab;
  int32_t* ecnt;           // [N] zeroed here
  int C, N;
  const int32_t* roff;     // [R] starts of the relation lists (last row: rtotal[1] - roff[R-1])
  const int32_t* rtotal;
  const int32_t* ruoff;    // [R] first unit of the relation
  const float* rslab;
  int32_t* rcnt;           // [R] zeroed here
  int R;
};

// The step between twin tables (kge_train_step_swap; pullself.hip, step1.hip): the wave that has a row's gradient in registers applies
// the optimizer to that row -- parameters read from the step's input tables, written to the twin tables (every other wave still
// gathers the OLD rows), moments in place -- so neither a gradient table nor an optimizer sweep exists.  lr_t of the update: read from
// the device-resident step state when the whole step is one call (lr_state / lr_slot), by value (lr_t) when its halves are issued apart.
// Chained steps (kge_train_step_swap_chain; step1.hip): the closing launch of a step does ~1 us of work and costs ~10 us plus a ~7 us
// kernel boundary on the step stream, so in a run of steps the relation rows, the loss and the step state are produced INSIDE the
// step's one launch, by R "closer" workgroups (one per relation row) and one "state" workgroup placed late in the grid -- behind the
// forward workgroups and most of the entity-role ones, so that by the time they are dispatched the forward has long finished and what
// they do (a few dependent memory round trips) hides behind the rest of the entity role.  The forward workgroups store their partial
// rows and loss partials write-through and then a flag word each; a closer polls the flags of the workgroups that hold its relation's
// units (the state workgroup: all of them) and reads the rows with cache-bypassing loads.  No atomics, no word that many waves write
// or that everybody polls.  Relation rows go to Rel_out, which no wave of the launch reads: the next launch sees them across the
// kernel boundary like everything else.  lr_t of the update comes from a two-slot ring (a wave may read it after the state workgroup
// has moved the step state on); the first step of a chain is preceded by a small launch that fills the ring and clears the flags.
// (Versions that put the closers of step i at the START of step i + 1's launch, with a gate in front of every other wave, measured
// 6 us more per step; a gate counted up with atomics 40 us more: profiles/r04_experiments.txt.)
struct StepChain {
  uint32_t* bflags;     // [forward workgroups] token of the launch whose results the workgroup stored last (null: not chained)
  uint32_t* timeouts;   // [1] flag waits that gave up (a fault, reported through the error word)
  float* lr_ring;       // [2] lr_t of update seq in slot seq & 1
  uint32_t* err;
  int64_t seq;          // number of this step in the chain
};

struct StepTail {
  int kind;                    // KGE_OPT_SGD | KGE_OPT_ADAM_TF
  float *E_out, *mE, *vE;      // entity side (E_in is the kernels' E)
  float *Rel_out, *mR, *vR;    // relation side (Rel_in is the kernels' Rel)
  float lr_t;                  // used when lr_state is null
  const float* lr_state;       // the step state's "lr_t of the next update" word: what the first launch of a whole step reads ...
  float* lr_slot;              // ... and copies here for the closing launch (whose last workgroup overwrites the state word)
  float b1, b2, eps;
  const int32_t* roff;         // [R] starts of the relation lists (last row: rtotal[1] - roff[R-1])
  const int32_t* rtotal;
  const int32_t* ruoff;        // [R] first unit of the relation
  const float* rslab;
  int32_t* rcnt;               // [R] zeroed here
  int R, d_r;
  int parts;                   // 1 = entity half, 2 = relation half (forward + relation rows + loss + step state), 3 = both
  // Owner-computes window (kge_train_step_swap_window; SURVEY 8(e): tables row-sharded, every rank holds the same batch): this call
  // updates entity rows [ent_lo, ent_hi) and relation rows [rel_lo, rel_hi) only -- the rest of E_out / Rel_out is some other rank's
  // to write -- and its loss is the sum over the rows of ITS relations.  Default: everything ([0, N), [0, R)).  ecnt: the plan's
  // entity counters, cleared by the closing launch when the window leaves rows out (nobody else would).
  int ent_lo, ent_hi, rel_lo, rel_hi;
  int32_t* ecnt;
  int N;
  StepChain ch;                // ch.timeouts non-null: a chained step
};
inline void step_tail_whole(StepTail& t, int N, int R) { t.ent_lo = 0; t.ent_hi = N; t.rel_lo = 0; t.rel_hi = R; t.ecnt = nullptr; t.N = N; }

struct PullPlan {
  int32_t* cnt;     // [NK]   contributions per key (key = entity id, or N + relation id)
  int32_t* off;     // [NK]   list start
  int32_t* ranks;   // [3][n] rank of (triple, role) inside its destination list (-1: invalid ids)
  int32_t* eoff;    // [NK]   first extra-chunk unit of the key
  int32_t* poff;    // [NK]   first slab slot of a multi-chunk key
  int32_t* totals;  // [4]    {extra units, slab slots}
  void* tsum;       // [tiles] / [tiles / 1024] Tri: sums of the counters per tile of 1024 keys and per group of 1024 tiles (scan of a
  void* ssum;       //          large table: scan_sums_launch); may be null where the table never has more than kScanDirect tiles
  int2* emap;       // [max_extra] (key, chunk) of extra units
  int4* entries;    // [3n]   {a | kind<<31, b, coef bits, t}
  float* coef;      // [n]
  float* slab;      // [2*max_extra][d]
  int max_extra;
  int chunk;
};

// One wave per key; multi-chunk keys add their slab rows, in chunk order, into the gradient table.
// Also leaves cnt all-zero for the next step: the workspace is self-cleaning (it is zeroed once, at allocation).
__device__ __forceinline__ void pull_finish_wave(int key, float* __restrict__ gE, float* __restrict__ gR, int N, int NK, int d_e,
                                                 int32_t* __restrict__ cnt, const int32_t* __restrict__ poff, const float* __restrict__ slab,
                                                 int C) {
  if (key >= NK) return;
  const int lane = threadIdx.x & 63;
  const int c = cnt[key];
  if (lane == 0 && c) cnt[key] = 0;
  const int nch = (c + C - 1) / C;
  if (nch <= 1) return;
  float* dst = key >= N ? gR + (int64_t)(key - N) * d_e : gE + (int64_t)key * d_e;
  const float* src = slab + (int64_t)poff[key] * d_e;
  for (int k = lane; k < d_e; k += 64) {
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= nch; j += 8) {  // eight independent loads in flight, summed in chunk order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(j + u) * d_e + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < nch; ++j) s += src[(int64_t)j * d_e + k];
    dst[k] += s;
  }
}

// ---- offsets of the per-key lists (shared by both pull variants): exclusive scans of {list length, extra-chunk units, slab
// slots} over NK keys, plus the (key, chunk) map of the extra units.  Multi-workgroup: every workgroup owns a tile of 1024 keys,
// sums the counters of all keys in front of its tile itself (the counters of 15 000 - 125 000 keys are L2 hits; no look-back
// protocol, no second launch) and scans its tile.  The single-workgroup scan this replaces walked the tiles one after the other:
// 26 us for 14 541 keys, most of it load latency per tile.
struct Tri { int a, b, c; };
__device__ __forceinline__ Tri tri_add(Tri x, Tri y) { return Tri{x.a + y.a, x.b + y.b, x.c + y.c}; }
__device__ __forceinline__ Tri tri_of(int n, int C) { const int nch = (n + C - 1) / C; return Tri{n, nch > 1 ? nch - 1 : 0, nch > 1 ? nch : 0}; }
constexpr int kScanTile = 1024;
// Summing everything in front of a tile is quadratic in the number of tiles: fine for the tables of the BASELINE configs (15 - 121
// tiles), not for the tables 288 GB of HBM hold (10 M entities = 9 766 tiles: 45 ms per plan, measured).  Past kScanDirect tiles two
// small launches run first -- the sum of every tile (tsum) and of every group of kScanGroup tiles (ssum) -- and a workgroup then adds
// the group sums in front of its group and the tile sums in front of it inside the group: at most ntiles / 1024 + 1024 loads.
constexpr int kScanDirect = 128, kScanGroup = 1024;
// 256 threads; `red` holds 8 Tri.  The workgroup of tile `tile` writes off / eoff / poff of its keys and their emap entries; the
// workgroup of the last tile writes totals = {extra units, slab slots, list entries}.  tsum / ssum: null = sum the counters directly.
__device__ __forceinline__ void scan_tile_body(const int32_t* __restrict__ cnt, int NK, int C, int tile, int32_t* __restrict__ off,
                                               int32_t* __restrict__ eoff, int32_t* __restrict__ poff, int32_t* __restrict__ totals,
                                               int2* __restrict__ emap, int max_extra, Tri* red, const Tri* __restrict__ tsum = nullptr,
                                               const Tri* __restrict__ ssum = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = tile * kScanTile;
  // (1) everything in front of the tile
  Tri pre{0, 0, 0};
  if (tsum) {
    const int g = tile / kScanGroup;
    for (int i = tid; i < g; i += 256) pre = tri_add(pre, ssum[i]);
    for (int i = g * kScanGroup + tid; i < tile; i += 256) pre = tri_add(pre, tsum[i]);
  } else {
    int k = tid;
    for (; k + 7 * 256 < base; k += 8 * 256) {  // eight independent loads in flight
      int c8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c8[u] = cnt[k + u * 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) pre = tri_add(pre, tri_of(c8[u], C));
    }
    for (; k < base; k += 256) pre = tri_add(pre, tri_of(cnt[k], C));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pre = tri_add(pre, Tri{__shfl_xor(pre.a, d, 64), __shfl_xor(pre.b, d, 64), __shfl_xor(pre.c, d, 64)});
  if (lane == 0) red[wave] = pre;
  // (2) the tile: 4 consecutive keys per thread
  const int k0 = base + tid * 4;
  int n[4];
  Tri v[4], run{0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    n[j] = k0 + j < NK ? cnt[k0 + j] : 0;
    v[j] = run;
    run = tri_add(run, tri_of(n[j], C));
  }
  Tri inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const Tri up{__shfl_up(inc.a, d, 64), __shfl_up(inc.b, d, 64), __shfl_up(inc.c, d, 64)};
    if (lane >= d) inc = tri_add(inc, up);
  }
  if (lane == 63) red[4 + wave] = inc;
  __syncthreads();
  Tri p = tri_add(tri_add(red[0], red[1]), tri_add(red[2], red[3]));
  for (int w = 0; w < wave; ++w) p = tri_add(p, red[4 + w]);
  p = tri_add(p, Tri{inc.a - run.a, inc.b - run.b, inc.c - run.c});
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k0 + j < NK) {
      const Tri x = tri_add(p, v[j]);
      off[k0 + j] = x.a; eoff[k0 + j] = x.b; poff[k0 + j] = x.c;
      const int nch = (n[j] + C - 1) / C;
      for (int c = 1; c < nch; ++c)
        if (x.b + c - 1 < max_extra) emap[x.b + c - 1] = make_int2(k0 + j, c);
    }
  if (base + kScanTile >= NK && tid == 255) {
    const Tri t = tri_add(p, run);
    totals[0] = t.b; totals[1] = t.c; totals[2] = t.a;
  }
}

bool pull_supported(const kge_model_desc* md);
int64_t pull_workspace_bytes(const kge_model_desc* md, int64_t n);
// histogram of destination rows + scans + chunk map (depends on the ids only)
int64_t pull_header_bytes(const kge_model_desc* md);  // zero-on-entry counters at a fixed place of the workspace
int pull_prepare(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, void* pull_hdr,
                 void* pull_ws, hipStream_t st, PullPlan* plan);
// gather + score; loss >= 0: pointwise loss fused (plan->coef, loss_partials[0..*n_partials)); loss < 0: scores only
int pull_forward(const kge_model_desc* md, int loss, float margin, const float* E, const float* Rel, const int32_t* pos,
                 int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& plan, float* scores, float* loss_partials,
                 int* n_partials, uint32_t* err, uint64_t seed, hipStream_t st);
// list fill + per-destination gather-reduce into gE/gR (+=).  g_is_dscore: plan.coef holds dL/dscore (pairwise path)
int pull_backward(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const int32_t* pos,
                  int64_t m_pos, const int32_t* neg, int64_t m_neg, const PullPlan& plan, bool g_is_dscore, const float* scores,
                  uint64_t seed, hipStream_t st);

// variant 2 (pull2.hip): relation-grouped forward with the relation gradient fused in; pointwise losses only
bool pull2_supported(const kge_model_desc* md, const kge_loss_desc* ld);
int64_t pull2_workspace_bytes(const kge_model_desc* md, int64_t n);
int train_grads_pull2(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR,
                      const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* scores, float* loss_partials,
                      int* n_partials, float* loss_out, void* hdr, void* ws, uint32_t* err, uint64_t seed, hipStream_t st);

// plan built ahead of the step (ids only) and consumed by it
int64_t batch_plan_bytes(const kge_model_desc* md, int64_t n);
// pos / neg arrays, or (cand_rows != null) the batch in folded form: candidate rows + multiplicities
int batch_plan_build(const kge_model_desc* md, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, const int32_t* cand_rows,
                     const int32_t* cand_cnt, int pos_lanes, int64_t n_pos_cand, int64_t n_neg_cand, void* plan, int parts, uint32_t* err,
                     hipStream_t st);
struct StepAdvance;  // kge_internal.hpp
// pull.hip: tile and group sums of NK counters in front of the scan (no-op up to kScanDirect tiles); returns true when they were made
bool scan_sums_launch(const int32_t* cnt, int NK, int C, void* tsum, void* ssum, hipStream_t st);
int64_t scan_sums_bytes(int64_t NK);  // room for tsum + ssum (256-aligned halves: tsum first)
int train_grads_planned(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gE, float* gR, void* plan,
                        int64_t m_pos, int64_t m_neg, float* loss_partials, int* n_partials, float* loss_out, void* ws, uint64_t seed,
                        hipStream_t st, const StepAdvance& adv, OptFinish* finish, const StepTail* tail = nullptr, int grad_parts = 3);
bool grad_halves_supported(const kge_model_desc* md);
// pullself.hip: the step between twin tables
bool pull_step_supported(const kge_model_desc* md);

// HolE (hole.hip)
int hole_forward(const kge_model_desc* md, int loss, float margin, const float* E, const float* Rel, const int32_t* pos,
                 int64_t m_pos, const int32_t* neg, int64_t m_neg, float* coef, float* scores, float* loss_partials,
                 int* n_partials, uint32_t* err, uint64_t seed, bool training, hipStream_t st);
int hole_backward(const kge_model_desc* md, const float* E, const float* Rel, float* gE, float* gR, const PullPlan& plan,
                  uint64_t seed, hipStream_t st);
int hole_build_q(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t nq, int side, int Kpad,
                 float* Qv, uint32_t* err, hipStream_t st);

// hole_spec.hip: HolE without dropout as ComplEx on half spectra
int hole_spec_half(int d);
int hole_spec_basis(int d, float* WE, float* WR, hipStream_t st);
int hole_spec_forward(const float* X, int64_t rows, int d, const float* W, float* out, hipStream_t st);
int hole_spec_adjoint(float* Ghat, int64_t rows, int d, const float* W, float* G, int zero_hat, hipStream_t st);

}  // namespace kge
