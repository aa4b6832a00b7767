// The training step between twin tables (kge_train_step_swap, parts = 3) in one launch + a small closing launch: workgroups
// [0, rel_blocks) play the relation half -- the relation-grouped forward (fwd_rel.hpp: scores, loss terms, partial relation-gradient
// rows) -- and the rest the self-contained entity half (pull_self.hpp: per entity row gather, own scores, gradient row, optimizer).
// Neither half reads what the other writes, so they share the launch.  rel_tail_kernel (pullself.hip) follows on the stream: relation
// rows = sums of the partial rows + optimizer -> Rel_out, the loss, the step state.
// (Closing the relation rows inside the same launch with tickets -- the last unit of a relation sums its partial rows -- needs a
// device-scope release fence in each of the ~4 100 unit waves; on gfx950 that is an L2 write-back while the entity half keeps 35 MB of
// rows dirty: measured 265 us for the launch against 71 us + 10 us this way.  profiles/r02_experiments.txt.)
// The relation workgroups come first in the grid (they hold the longest dependent chains of the launch).
// Reference: the default Keras train step over build_model's wrapper (tests.py:63-89,130-134; KGEkeras/models.py:134-222;
// utils.py:288-298), optimizer 'adam' / 'sgd' (tests.py:83).
#include "fwd_rel.hpp"
#include "pull_self.hpp"

namespace kge {

// ---- chained steps (StepChain, pull.hpp): what the first workgroups of a launch do for the PREVIOUS step
struct ChainDev {
  int pre;                     // workgroups in front of the relation role: closers (R of them, from the chain's second step on) + 1 state workgroup; 0: not chained
  int close;                   // closers present
  uint32_t* flags;             // [R] closer r stores the token here behind its row (the flush passes null: nobody waits)
  uint32_t* ready;             // [64 lines] the gate: the state workgroup stores the token here once all flags carry it
  uint32_t token;              // this launch's token (the step's number in the chain, >= 1 where closers run); 0: the chain's first launch clears flags and gate
  uint32_t* timeouts;
  // closers: the previous step's relation rows -- parameters from ITS input table (this step's Rel_out), written to its output table (this step's Rel_in)
  const float* Rel_from; float* Rel_to; float *mR, *vR;
  const float* slab_prev; const int32_t* meta_prev; const float* partials_prev; int n_partials;
  const float* lr_prev;        // lr_t of the previous update
  int kind; float b1, b2, eps; int d_r;
  // state workgroup: this step's relation list bounds for the next launch's closers, the plan's relation counters left zero, lr_t of the next update
  int32_t* meta_mine; const int32_t* roff; const int32_t* rtotal; const int32_t* ruoff; int32_t* rcnt;
  float* lr_next;              // lr_ring[(seq + 1) & 3]
  float* lr_first;             // seq == 0: the state's lr word is copied here (lr_ring[0]) for the closers of the next launch
  float* loss_out; uint32_t* err;
  StepAdvance adv;
};

__device__ __forceinline__ void store4_sc1(float* p, float4 v) {  // write-through: visible to the other compute units' misses once acknowledged
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}

// One workgroup per relation row of the previous step (tail_relation_row's arithmetic and summation order: bit-identical rows): wave w
// sums its share of the row's partial rows, wave 0 applies the optimizer, stores the row write-through and counts the gate up.
__device__ __forceinline__ void chain_close_row(const ChainDev& c, int row, int R) {
  __shared__ float4 part[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int d4 = c.d_r >> 2;
  const int rows = c.meta_prev[R + row];
  const int nch = rel_units(rows);
  const int per = (nch + 3) >> 2;
  const int j0 = min(wave * per, nch), j1 = min(j0 + per, nch);
  const float lr = c.lr_prev[0];
  const int64_t u0 = c.meta_prev[row];
  for (int q0 = 0; q0 < d4; q0 += 64) {
    const int q = q0 + lane;
    const int64_t at = (int64_t)row * c.d_r + q * 4;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f), m = w, v = w;
    if (wave == 0 && q < d4) {  // (issued in front of the partial rows: nothing here depends on them)
      w = *reinterpret_cast<const float4*>(c.Rel_from + at);
      if (c.kind == KGE_OPT_ADAM_TF) { m = *reinterpret_cast<const float4*>(c.mR + at); v = *reinterpret_cast<const float4*>(c.vR + at); }
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < d4) {
      const float4* src = reinterpret_cast<const float4*>(c.slab_prev) + u0 * d4 + q;
      int j = j0;
      for (; j + 4 <= j1; j += 4) {
        const float4 a = src[(int64_t)j * d4], b = src[(int64_t)(j + 1) * d4], c2 = src[(int64_t)(j + 2) * d4], d = src[(int64_t)(j + 3) * d4];
        s = (((s + a) + b) + c2) + d;
      }
      for (; j < j1; ++j) s = s + src[(int64_t)j * d4];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && q < d4) {
      const float4 g = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
      tail_upd(c.kind, lr, c.b1, c.b2, c.eps, w, g, m, v);
      store4_sc1(c.Rel_to + at, w);
      if (c.kind == KGE_OPT_ADAM_TF) { *reinterpret_cast<float4*>(c.mR + at) = m; *reinterpret_cast<float4*>(c.vR + at) = v; }
    }
    __syncthreads();
  }
  if (wave == 0 && c.flags) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's row stores are acknowledged ...
    if (lane == 0) __hip_atomic_store(c.flags + row, c.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ... before this row's flag says so
  }
}

// The state workgroup: the previous step's loss and step state (when there is one), then what the NEXT launch needs of this step.
__device__ __forceinline__ void chain_state_block(const ChainDev& c, int R, float* red) {
  if (c.close) {
    float s = 0.f;
    for (int i = threadIdx.x; i < c.n_partials; i += 256) s += c.partials_prev[i];
    const float b = block_sum(s, red);
    if (threadIdx.x == 0) { c.loss_out[0] = b; step_advance(c.adv, b); }
  }
  if (threadIdx.x == 0) {
    if (c.lr_first) c.lr_first[0] = step_state_lr(c.adv.step)[0];                                          // (the chain's first launch: update t0 + 1)
    if (c.lr_next) c.lr_next[0] = lr_of_step(c.adv.kind, c.adv.lr, c.adv.b1, c.adv.b2, c.adv.step[0] + 2);  // this launch is update step[0] + 1
    if (c.err && c.timeouts && c.timeouts[0]) { atomicOr(c.err, 32u); c.timeouts[0] = 0u; }
  }
  if (c.meta_mine)
    for (int r = threadIdx.x; r < R; r += 256) {
      const int o = c.roff[r];
      c.meta_mine[r] = c.ruoff[r];
      c.meta_mine[R + r] = (r + 1 < R ? c.roff[r + 1] : c.rtotal[1]) - o;
      c.rcnt[r] = 0;
    }
  if (!c.ready) return;
  if (c.token == 0u) {  // the chain's first launch (nobody waits in it): flags and gate start from zero whatever an earlier chain left
    for (int r = threadIdx.x; r < R; r += 256) c.flags[r] = 0u;
    if (threadIdx.x < kGateLines) c.ready[threadIdx.x * kGateStride] = 0u;
    return;
  }
  // the gate: this workgroup alone polls the closers' flags; when all R carry the token, the 64 lines the readers poll get it
  for (int spins = 0;; ++spins) {
    int ok = 1;
    for (int r = threadIdx.x; r < R; r += 256) ok &= __hip_atomic_load(c.flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.token;
    if (__syncthreads_and(ok)) break;
    if (spins >= (1 << 16)) { if (threadIdx.x == 0) atomicAdd(c.timeouts, 1u); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (threadIdx.x < kGateLines) __hip_atomic_store(c.ready + threadIdx.x * kGateStride, c.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The end of a chain (kge_train_step_swap_flush): the last step's relation rows, loss and step state; both gates left zero.
__global__ __launch_bounds__(256) void chain_flush_kernel(ChainDev c, int R) {
  __shared__ float red[16];
  if ((int)blockIdx.x < R) { chain_close_row(c, blockIdx.x, R); return; }
  chain_state_block(c, R, red);
}

template <int MODEL, int NORM, int LOSS, int NV, int G, bool DROP>
__global__ __launch_bounds__(256) void step1_kernel(FwdRelArgs fa, SelfArgs sa, float* __restrict__ loss_partials, int rel_blocks, int R, ChainDev cd) {
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  __shared__ float4 selfrow[4][NACC * G];
  __shared__ float red[16];
  const int wave = threadIdx.x >> 6;
  int bid = blockIdx.x;
  if (cd.pre) {  // chained: closers and the state workgroup come first in the grid -- they wait for nothing, everybody else waits for them
    if (bid < cd.pre - 1) { chain_close_row(cd, bid, R); return; }
    if (bid == cd.pre - 1) { chain_state_block(cd, R, red); return; }
    bid -= cd.pre;
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
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_partials[bid] = b;
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
  const bool chained = ch.flags != nullptr;
  if (chained) {
    const int me = (int)(ch.seq & 1), prev = me ^ 1;
    cd.close = ch.seq > 0;
    cd.pre = (cd.close ? tail.R : 0) + 1;
    cd.flags = ch.flags; cd.ready = ch.ready; cd.token = (uint32_t)ch.seq; cd.timeouts = ch.timeouts;
    cd.Rel_from = tail.Rel_out; cd.Rel_to = const_cast<float*>(fa.Rel); cd.mR = tail.mR; cd.vR = tail.vR;
    cd.slab_prev = ch.slab[prev]; cd.meta_prev = ch.relmeta[prev]; cd.partials_prev = ch.partials[prev]; cd.n_partials = rel_blocks;
    cd.lr_prev = ch.lr_ring + ((ch.seq + 3) & 3);
    cd.kind = tail.kind; cd.b1 = tail.b1; cd.b2 = tail.b2; cd.eps = tail.eps; cd.d_r = tail.d_r;
    cd.meta_mine = ch.relmeta[me]; cd.roff = tail.roff; cd.rtotal = tail.rtotal; cd.ruoff = tail.ruoff; cd.rcnt = tail.rcnt;
    cd.lr_next = ch.lr_ring + ((ch.seq + 1) & 3);
    cd.lr_first = ch.seq == 0 ? ch.lr_ring : nullptr;
    cd.loss_out = loss_out; cd.err = nullptr; cd.adv = adv;
    cd.adv.lr_slot = nullptr;
    // this launch: partial rows and loss partials into this step's buffers; lr_t from the ring (the chain's first step: from the state
    // word, which nobody moves during that launch); every reader of a relation row behind the gate
    fa.rslab = ch.slab[me];
    loss_partials = ch.partials[me];
    if (cd.close) {
      sa.lr_state = ch.lr_ring + (ch.seq & 3);
      sa.gate = ch.ready; sa.gate_need = cd.token; sa.gate_to = ch.timeouts;
    }
    fa.gate = ch.ready; fa.gate_need = cd.token; fa.gate_to = ch.timeouts;  // (first step: token 0, no wait -- but still the list starts for the rows of a relation)
    fa.gate_R = tail.R;
    sa.lr_keep = nullptr;
  }
  const int grid = cd.pre + rel_blocks + (int)cdiv(sa.row_n + e.max_extra, 4);  // (row_n: the rows of the step's entity window; all of them by default)
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (md->model == KGE_DISTMULT) launch_step1<KGE_DISTMULT, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->model == KGE_COMPLEX) launch_step1<KGE_COMPLEX, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->norm == 1) launch_step1<KGE_TRANSE, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else if (md->norm == 3) launch_step1<KGE_TRANSE, 3>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
    else launch_step1<KGE_TRANSE, 2>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R, cd);
  }
  KGE_HIP_TRY(hipGetLastError());
  if (chained) return KGE_OK;  // (closed by the next step of the chain, or by chain_flush_launch)
  const LossFinish lf{loss_partials, rel_blocks, loss_out, adv};
  return rel_tail_launch(tail, fa.Rel, lf, st);
}

// The end of a chain: step `seq` (the last one issued) is closed -- relation rows Rel_in -> Rel_out of THAT step, loss, step state.
int chain_flush_launch(const StepTail& tail, const float* Rel_in, int rel_blocks, float* loss_out, const StepAdvance& adv, uint32_t* err, hipStream_t st) {
  const StepChain& ch = tail.ch;
  if (!ch.flags) return fail(KGE_EINVAL, "kge_train_step_swap_flush: no chain state");
  const int last = (int)(ch.seq & 1);
  ChainDev cd{};
  cd.close = 1; cd.pre = tail.R + 1;
  cd.flags = nullptr; cd.ready = nullptr; cd.token = 0; cd.timeouts = ch.timeouts;
  cd.Rel_from = Rel_in; cd.Rel_to = tail.Rel_out; cd.mR = tail.mR; cd.vR = tail.vR;
  cd.slab_prev = ch.slab[last]; cd.meta_prev = ch.relmeta[last]; cd.partials_prev = ch.partials[last]; cd.n_partials = rel_blocks;
  cd.lr_prev = ch.lr_ring + (ch.seq & 3);
  cd.kind = tail.kind; cd.b1 = tail.b1; cd.b2 = tail.b2; cd.eps = tail.eps; cd.d_r = tail.d_r;
  cd.loss_out = loss_out; cd.err = err; cd.adv = adv; cd.adv.lr_slot = nullptr;
  ProfScope prof(KGE_PROF_OPT, st);
  hipLaunchKernelGGL(chain_flush_kernel, dim3(tail.R + 1), dim3(256), 0, st, cd, tail.R);
  KGE_HIP_TRY(hipGetLastError());
  return KGE_OK;
}

}  // namespace kge
