// The training step between twin tables (kge_train_step_swap, parts = 3) in one launch + a small closing launch: workgroups
// [0, rel_blocks) play the relation half -- the relation-grouped forward (fwd_rel.hpp: scores, loss terms, partial relation-gradient
// rows) -- and the rest the self-contained entity half (pull_self.hpp: per entity row gather, own scores, gradient row, optimizer).
// Neither half reads what the other writes, so they share the launch.  rel_tail_kernel (pullself.hip) follows on the stream: relation
// rows = sums of the partial rows + optimizer -> Rel_out, the loss, the step state.
// Chained steps (kge_train_step_swap_chain; StepChain in pull.hpp): no closing launch -- R closer workgroups and a state workgroup sit
// late in this launch's grid and wait on per-workgroup flags of the forward (write-through stores, cache-bypassing loads).
// (Round 2 closed the relation rows inside the launch with tickets -- the last unit of a relation sums its partial rows -- which needs
// a device-scope release fence in each of the ~4 100 unit waves; on gfx950 that is an L2 write-back while the entity half keeps 35 MB
// of rows dirty: 265 us for the launch against 71 us + 10 us.  profiles/r02_experiments.txt, r04_experiments.txt.)
// The relation workgroups come first in the grid (they hold the longest dependent chains of the launch).
// Reference: the default Keras train step over build_model's wrapper (tests.py:63-89,130-134; KGEkeras/models.py:134-222;
// utils.py:288-298), optimizer 'adam' / 'sgd' (tests.py:83).
#include "fwd_rel.hpp"
#include "pull_self.hpp"

namespace kge {

// ---- chained steps (StepChain, pull.hpp): the closers and the state workgroup of the launch
struct ChainDev {
  int n;                       // R + 1 (closers + the state workgroup); 0: not chained
  int at;                      // they sit behind this many entity-role workgroups in the grid
  uint32_t tok;                // this launch's token
  uint32_t* bflags;            // [forward workgroups]
  uint32_t* timeouts;
  const float* Rel_in; float* Rel_out; float *mR, *vR;
  const float* slab; const float* partials; int n_partials;
  const float* lr_mine;        // lr_t of this update: ring[seq & 1]
  float* lr_next;              // ring[(seq + 1) & 1]
  int kind; float b1, b2, eps; int d_r;
  const int32_t* roff; const int32_t* rtotal; const int32_t* ruoff; int32_t* rcnt;
  float* loss_out; uint32_t* err;
  StepAdvance adv;
};

// One workgroup per relation row (tail_relation_row's arithmetic and summation order: bit-identical rows): waits for the forward
// workgroups that hold the relation's units, then wave w sums its share of the partial rows, wave 0 applies the optimizer.
__device__ __forceinline__ void chain_close_row(const ChainDev& c, int row, int R) {
  __shared__ float4 part[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d4 = c.d_r >> 2;
  const int rows = (row + 1 < R ? c.roff[row + 1] : c.rtotal[1]) - c.roff[row];
  const int nch = rel_units(rows);
  const int per = (nch + 3) >> 2;
  const int j0 = min(wave * per, nch), j1 = min(j0 + per, nch);
  const float lr = c.lr_mine[0];
  const int64_t u0 = c.ruoff[row];
  if (nch > 0 && !flags_wait(c.bflags, (int)(u0 >> 2), (int)((u0 + nch - 1) >> 2), c.tok) && threadIdx.x == 0) atomicAdd(c.timeouts, 1u);
  if (threadIdx.x == 0) c.rcnt[row] = 0;  // (every unit of the relation has read it)
  for (int q0 = 0; q0 < d4; q0 += 64) {
    const int q = q0 + lane;
    const int64_t at = (int64_t)row * c.d_r + q * 4;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f), m = w, v = w;
    if (wave == 0 && q < d4) {  // (issued in front of the partial rows: nothing here depends on them)
      w = *reinterpret_cast<const float4*>(c.Rel_in + at);
      if (c.kind == KGE_OPT_ADAM_TF) { m = *reinterpret_cast<const float4*>(c.mR + at); v = *reinterpret_cast<const float4*>(c.vR + at); }
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d4) {
      const float* src = c.slab + (u0 * d4 + q) * 4;
      int j = j0;
      for (; j + 4 <= j1; j += 4) {
        const float4 a = load4_agent(src + (int64_t)j * c.d_r), b = load4_agent(src + (int64_t)(j + 1) * c.d_r), c2 = load4_agent(src + (int64_t)(j + 2) * c.d_r),
                     d = load4_agent(src + (int64_t)(j + 3) * c.d_r);
        s = (((s + a) + b) + c2) + d;
      }
      for (; j < j1; ++j) s = s + load4_agent(src + (int64_t)j * c.d_r);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && q < d4) {
      const float4 g = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      tail_upd(c.kind, lr, c.b1, c.b2, c.eps, w, g, m, v);
      *reinterpret_cast<float4*>(c.Rel_out + at) = w;
      if (c.kind == KGE_OPT_ADAM_TF) { *reinterpret_cast<float4*>(c.mR + at) = m; *reinterpret_cast<float4*>(c.vR + at) = v; }
    }
    __syncthreads();
  }
}

// The state workgroup: the loss (rel_tail_kernel's summation order) and the step state, once every forward workgroup has stored its
// loss partial; lr_t of the next update into the other slot of the ring.
__device__ __forceinline__ void chain_state_block(const ChainDev& c, float* red) {
  if (!flags_wait(c.bflags, 0, c.n_partials - 1, c.tok) && threadIdx.x == 0) atomicAdd(c.timeouts, 1u);
  float s = 0.f;
  for (int i = threadIdx.x; i < c.n_partials; i += 256) s += __hip_atomic_load(c.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float b = block_sum(s, red);
  if (threadIdx.x == 0) {
    c.loss_out[0] = b;
    step_advance(c.adv, b);                         // t += 1, the state's lr word = lr_t of update t + 2 (nobody of this launch reads that word), loss sum
    c.lr_next[0] = step_state_lr(c.adv.step)[0];
    if (c.err && c.timeouts[0]) { atomicOr(c.err, 32u); c.timeouts[0] = 0u; }
  }
}

// In front of the first step of a chain: lr_t of that update into slot 0 of the ring, the forward workgroups' flags cleared.
__global__ __launch_bounds__(256) void chain_begin_kernel(float* ring, const float* state_lr, uint32_t* bflags, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) bflags[i] = 0u;
  if (i == 0) ring[0] = state_lr[0];
}

// (waves_per_eu: the DistMult d <= 256 instance sits at 93 - 98 registers depending on what else shares the kernel; 97+ rounds up to 104 and
//  costs every role its fifth wave per SIMD -- 63 -> 66 us alone, 8.4 -> 9.4 ms on the 10 M-entity table -- so the compiler is held to 96 there)
template <int MODEL, int NORM, int LOSS, int NV, int G, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODEL == KGE_DISTMULT && NV == 4 && !DROP ? 5 : 1))) void step1_kernel(FwdRelArgs fa, SelfArgs sa, float* __restrict__ loss_partials, int rel_blocks, int R, ChainDev cd) {
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  __shared__ float4 selfrow[4][NACC * G];
  __shared__ float red[16];
  const int wave = threadIdx.x >> 6;
  int bid = blockIdx.x;
  if (cd.n && bid >= rel_blocks) {  // chained: closers and the state workgroup sit behind cd.at entity-role workgroups
    const int e = bid - rel_blocks;
    if (e >= cd.at && e < cd.at + cd.n) {
      if (e - cd.at < R) chain_close_row(cd, e - cd.at, R);
      else chain_state_block(cd, red);
      return;
    }
    if (e >= cd.at + cd.n) bid -= cd.n;
  }
  if (bid >= rel_blocks) {
    pull_self_unit<MODEL, NORM, LOSS, NV, G, DROP>(sa, (bid - rel_blocks) * 4 + wave, selfrow[wave]);
    return;
  }
  // lr_t of this update for the closing launch (rel_tail_kernel), whose last workgroup overwrites the state word
  if (bid == 0 && threadIdx.x == 0 && sa.lr_keep) sa.lr_keep[0] = sa.lr_state[0];
  const int unit = bid * 4 + wave;
  float lsum = 0.f;
  int len, c;
  if (unit < fa.rtotal[0] && rel_in_window(fa, unit)) fwd_rel_unit<MODEL, NORM, LOSS, NV, DROP, G>(fa, unit, R, lsum, &len, &c);
  const float b = block_sum(lsum, red);  // (its barrier: every wave of the workgroup has stored its partial row -- and, chained, seen it acknowledged)
  if (threadIdx.x == 0) {
    if (cd.n) {
      __hip_atomic_store(loss_partials + bid, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(cd.bflags + bid, cd.tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      loss_partials[bid] = b;
    }
  }
}

// Sub-group width: rows of up to 32 float4 chunks on 16 lanes (NV = 1, 2).  Wider rows (33..64 chunks, d = 200): 16 lanes x NV = 4
// or 32 lanes x NV = 2 -- the same instruction count per row, half the row-sized register arrays.  Measured at the FB15k-237 shape:
// DistMult 65.7 us on 16 lanes (93 VGPRs, 5 waves per SIMD) against 67.2 us on 32 (59 VGPRs, 8 waves: the launch is bound by the
// fabric, not by occupancy); TransE-L2 91.6 us (98 VGPRs, 4 waves) against 75.4 us.  So: DistMult 16 lanes, the others 32.
template <int MODEL, int NORM, int LOSS>
static void launch_step1_nv(int nchunk, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, float* lp, int rb, int R, const ChainDev& cd) {
  const bool drop = fa.thr != 0u;
#define KGE_S1(NVV, GW) do { if (drop) hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, NVV, GW, true>), dim3(grid), dim3(256), 0, st, fa, sa, lp, rb, R, cd); \
                             else hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, NVV, GW, false>), dim3(grid), dim3(256), 0, st, fa, sa, lp, rb, R, cd); } while (0)
  if (nchunk <= 16) KGE_S1(1, 16);
  else if (nchunk <= 32) KGE_S1(2, 16);
  else if (MODEL == KGE_DISTMULT) KGE_S1(4, 16);
  else KGE_S1(2, 32);
#undef KGE_S1
}
template <int MODEL, int NORM>
static void launch_step1(int loss, int nchunk, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, float* lp, int rb, int R, const ChainDev& cd) {
  if (loss == KGE_POINTWIZE_HINGE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_HINGE>(nchunk, grid, st, fa, sa, lp, rb, R, cd);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nchunk, grid, st, fa, sa, lp, rb, R, cd);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nchunk, grid, st, fa, sa, lp, rb, R, cd);
  else launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nchunk, grid, st, fa, sa, lp, rb, R, cd);
}

int rel_tail_launch(const StepTail& t, const float* Rel, const LossFinish& lf, hipStream_t st);  // pullself.hip
void self_args_of(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg, const PullPlan& e,
                  const StepTail& tail, float lr_t, int nchunk, SelfArgs* out);  // pullself.hip

// fa: the forward's arguments as pull2.hip fills them (coef / scores may be null); tail.{roff, rtotal, ruoff, rslab, rcnt, R, d_r} set.
int step1_launch(const kge_model_desc* md, const kge_loss_desc* ld, const FwdRelArgs& fa_in, int rel_blocks, const PullPlan& e, const StepTail& tail,
                 float* loss_partials, float* loss_out, const StepAdvance& adv, hipStream_t st) {
  if (!pull_step_supported(md)) return fail(KGE_EUNSUPPORTED, "step between twin tables: row width not covered");
  FwdRelArgs fa = fa_in;
  if (tail.rel_lo != 0 || tail.rel_hi != tail.R) { fa.rel_lo = tail.rel_lo; fa.rel_hi = tail.rel_hi; }
  SelfArgs sa;
  self_args_of(md, ld, fa.E, fa.Rel, fa.m_pos, fa.m_neg, e, tail, tail.lr_t, fa.nchunk, &sa);
  sa.seed = fa.seed;
  sa.lr_state = tail.lr_state;
  sa.lr_keep = tail.lr_state ? tail.lr_slot : nullptr;
  ChainDev cd{};
  const StepChain& ch = tail.ch;
  const bool chained = ch.bflags != nullptr;
  const int ent_blocks = (int)cdiv(sa.row_n + e.max_extra, 4);  // (row_n: the rows of the step's entity window; all of them by default)
  if (chained) {
    if (ch.seq == 0) {  // the chain starts: lr_t of its first update into the ring, flags cleared
      hipLaunchKernelGGL(chain_begin_kernel, dim3((int)cdiv(rel_blocks, 256)), dim3(256), 0, st, ch.lr_ring, tail.lr_state, ch.bflags, rel_blocks);
      KGE_HIP_TRY(hipGetLastError());
    }
    cd.n = tail.R + 1;
    cd.at = (int)((int64_t)ent_blocks * 13 / 16);  // late enough for the forward to have finished, early enough to hide behind the rest of the entity role (4/16 .. 16/16: 0.0978 .. 0.0965 .. 0.0969 ms per step)
    cd.tok = (uint32_t)(ch.seq & 0x3fffffff) + 1u;
    cd.bflags = ch.bflags; cd.timeouts = ch.timeouts;
    cd.Rel_in = fa.Rel; cd.Rel_out = tail.Rel_out; cd.mR = tail.mR; cd.vR = tail.vR;
    cd.slab = fa.rslab; cd.partials = loss_partials; cd.n_partials = rel_blocks;
    cd.lr_mine = ch.lr_ring + (ch.seq & 1); cd.lr_next = ch.lr_ring + ((ch.seq + 1) & 1);
    cd.kind = tail.kind; cd.b1 = tail.b1; cd.b2 = tail.b2; cd.eps = tail.eps; cd.d_r = tail.d_r;
    cd.roff = tail.roff; cd.rtotal = tail.rtotal; cd.ruoff = tail.ruoff; cd.rcnt = tail.rcnt;
    cd.loss_out = loss_out; cd.err = ch.err; cd.adv = adv; cd.adv.lr_slot = nullptr;
    fa.chain_tok = cd.tok;
    sa.lr_state = cd.lr_mine;  // (not the state's word: the state workgroup moves that on while entity waves are still running)
    sa.lr_keep = nullptr;
  }
  const int grid = rel_blocks + ent_blocks + cd.n;
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (md->model == KGE_DISTMULT) launch_step1<KGE_DISTMULT, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->model == KGE_COMPLEX) launch_step1<KGE_COMPLEX, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->norm == 1) launch_step1<KGE_TRANSE, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->norm == 3) launch_step1<KGE_TRANSE, 3>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else launch_step1<KGE_TRANSE, 2>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
  }
  KGE_HIP_TRY(hipGetLastError());
  if (chained) return KGE_OK;  // (the launch closed its own relation rows, loss and step state)
  const LossFinish lf{loss_partials, rel_blocks, loss_out, adv};
  return rel_tail_launch(tail, fa.Rel, lf, st);
}

}  // namespace kge
