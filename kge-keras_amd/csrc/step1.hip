// The whole training step between twin tables in ONE launch (kge_train_step_swap, parts = 3): workgroups [0, rel_blocks) play the
// relation half -- the relation-grouped forward (fwd_rel.hpp: scores, loss terms, partial relation-gradient rows) -- and the rest the
// self-contained entity half (pull_self.hpp: per entity row gather, own scores, gradient row, optimizer).  Neither half reads what
// the other writes, so they share the launch and fill each other's latency gaps; what used to be launch boundaries are tickets:
//   * the wave that counts a relation's row counter down to zero (its units subtract their rows; release / acquire fences around the
//     atomic) sums that relation's partial rows in unit order and applies the optimizer to the relation row -> Rel_out;
//   * the relation workgroup that draws the last ticket reduces the loss partials in block order, moves the step state on and
//     applies the optimizer to relations without rows in this batch (dense update, Keras semantics).
// Fixed summation orders everywhere: results do not depend on timing.  The relation workgroups come first in the grid (they hold
// the longest dependent chains of the launch).
// Reference: the default Keras train step over build_model's wrapper (tests.py:63-89,130-134; KGEkeras/models.py:134-222;
// utils.py:288-298), optimizer 'adam' / 'sgd' (tests.py:83).
#include "fwd_rel.hpp"
#include "pull_self.hpp"

namespace kge {

struct Step1Tail {
  StepTail t;            // relation side: Rel_out, moments, lr_t, the plan's relation lists
  float* loss_partials;  // [rel_blocks]
  float* loss_out;
  StepAdvance adv;
  int* ticket;           // zero between launches
  int rel_blocks;
  int tickets;           // 0: no tickets, no fences -- rel_tail_kernel follows on the stream and closes the relation rows and the loss
};

// One wave: relation row `row` = sum of its units' partial rows (unit order, eight loads in flight), optimizer -> Rel_out.
__device__ __forceinline__ void relation_row_finish(const StepTail& t, const float* __restrict__ Rel, int row, int c) {
  const int lane = threadIdx.x & 63;
  const int d4 = t.d_r >> 2;
  const int nch = c > 0 ? rel_units(c) : 0;
  for (int q = lane; q < d4; q += 64) {
    const float4* src = reinterpret_cast<const float4*>(t.rslab) + (int64_t)t.ruoff[row] * d4 + q;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = 0;
    for (; j + 8 <= nch; j += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = src[(int64_t)(j + u) * d4];
#pragma unroll
      for (int u = 0; u < 8; ++u) g = g + x[u];
    }
    for (; j < nch; ++j) g = g + src[(int64_t)j * d4];
    const int64_t at = (int64_t)row * t.d_r + q * 4;
    float4 w = *reinterpret_cast<const float4*>(Rel + at), m = make_float4(0.f, 0.f, 0.f, 0.f), v = m;
    if (t.kind == KGE_OPT_ADAM_TF) { m = *reinterpret_cast<const float4*>(t.mR + at); v = *reinterpret_cast<const float4*>(t.vR + at); }
    tail_upd(t.kind, t.lr_t, t.b1, t.b2, t.eps, w, g, m, v);
    *reinterpret_cast<float4*>(t.Rel_out + at) = w;
    if (t.kind == KGE_OPT_ADAM_TF) { *reinterpret_cast<float4*>(t.mR + at) = m; *reinterpret_cast<float4*>(t.vR + at) = v; }
  }
}

template <int MODEL, int NORM, int LOSS, int NV>
__global__ __launch_bounds__(256) void step1_kernel(FwdRelArgs fa, SelfArgs sa, Step1Tail tl) {
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  __shared__ float4 selfrow[4][NACC * 16];
  __shared__ float red[16];
  __shared__ int last_block;
  const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
  if ((int)blockIdx.x >= tl.rel_blocks) {
    pull_self_unit<MODEL, NORM, LOSS, NV>(sa, ((int)blockIdx.x - tl.rel_blocks) * 4 + wave, selfrow[wave]);
    return;
  }
  // ---- relation half
  const StepTail& t = tl.t;
  const int unit = blockIdx.x * 4 + wave;
  float lsum = 0.f;
  if (unit < fa.rtotal[0]) {
    int len = 0, c = 0;
    const int p = fwd_rel_unit<MODEL, NORM, LOSS, NV, false, true>(fa, unit, t.R, lsum, &len, &c);
    if (!tl.tickets) goto block_loss;
    __threadfence();  // this unit's partial row, before the ticket
    int old = 0;
    if (lane64 == 0) old = atomicSub(t.rcnt + p, len);
    old = __shfl(old, 0, 64);
    if (old == len) {  // the relation's last unit
      __threadfence();
      relation_row_finish(t, fa.Rel, p, c);
    }
  }
block_loss:
  const float b = block_sum(lsum, red);
  if (!tl.tickets) {
    if (threadIdx.x == 0) tl.loss_partials[blockIdx.x] = b;
    return;
  }
  if (threadIdx.x == 0) {
    tl.loss_partials[blockIdx.x] = b;
    __threadfence();
    last_block = atomicAdd(tl.ticket, 1) == tl.rel_blocks - 1;
  }
  __syncthreads();
  if (!last_block) return;
  __threadfence();
  float s = 0.f;
  for (int i = threadIdx.x; i < tl.rel_blocks; i += 256) s += tl.loss_partials[i];
  const float total = block_sum(s, red);
  if (threadIdx.x == 0) {
    tl.loss_out[0] = total;
    step_advance(tl.adv, total);
    tl.ticket[0] = 0;
  }
  for (int r = wave; r < t.R; r += 4) {  // relations without rows in this batch
    const int c = (r + 1 < t.R ? t.roff[r + 1] : t.rtotal[1]) - t.roff[r];
    if (c == 0) relation_row_finish(t, fa.Rel, r, 0);
  }
}

template <int MODEL, int NORM, int LOSS>
static void launch_step1_nv(int nv, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, const Step1Tail& tl) {
  if (nv <= 1) hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, 1>), dim3(grid), dim3(256), 0, st, fa, sa, tl);
  else if (nv <= 2) hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, 2>), dim3(grid), dim3(256), 0, st, fa, sa, tl);
  else hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, 4>), dim3(grid), dim3(256), 0, st, fa, sa, tl);
}
template <int MODEL, int NORM>
static void launch_step1(int loss, int nv, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, const Step1Tail& tl) {
  if (loss == KGE_POINTWIZE_HINGE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_HINGE>(nv, grid, st, fa, sa, tl);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nv, grid, st, fa, sa, tl);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nv, grid, st, fa, sa, tl);
  else launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nv, grid, st, fa, sa, tl);
}

int rel_tail_launch(const StepTail& t, const float* Rel, const LossFinish& lf, hipStream_t st);  // pullself.hip
void self_args_of(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, int64_t m_pos, int64_t m_neg, const PullPlan& e,
                  const StepTail& tail, float lr_t, int nchunk, SelfArgs* out);  // pullself.hip

// fa: the forward's arguments as pull2.hip fills them (coef / scores may be null); tail.{roff, rtotal, ruoff, rslab, rcnt, R, d_r} set.
int step1_launch(const kge_model_desc* md, const kge_loss_desc* ld, const FwdRelArgs& fa, int rel_blocks, const PullPlan& e, const StepTail& tail,
                 float* loss_partials, float* loss_out, const StepAdvance& adv, int* ticket, hipStream_t st) {
  if (!pull_step_supported(md)) return fail(KGE_EUNSUPPORTED, "step between twin tables: dropout or row width not covered");
  SelfArgs sa;
  self_args_of(md, ld, fa.E, fa.Rel, fa.m_pos, fa.m_neg, e, tail, tail.lr_t, fa.nchunk, &sa);
  // KGE_STEP1_TICKETS=1: the relation rows and the loss closed inside the launch (fences + tickets); default: by rel_tail_kernel behind it
  static const int tickets = [] { const char* v = getenv("KGE_STEP1_TICKETS"); return v ? atoi(v) : 0; }();
  const Step1Tail tl{tail, loss_partials, loss_out, adv, ticket, rel_blocks, tickets};
  const int grid = rel_blocks + (int)cdiv(sa.N + e.max_extra, 4);
  const int nv = (int)cdiv(fa.nchunk, 16);
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (md->model == KGE_DISTMULT) launch_step1<KGE_DISTMULT, 1>(ld->loss, nv, grid, st, fa, sa, tl);
    else if (md->model == KGE_COMPLEX) launch_step1<KGE_COMPLEX, 1>(ld->loss, nv, grid, st, fa, sa, tl);
    else if (md->norm == 1) launch_step1<KGE_TRANSE, 1>(ld->loss, nv, grid, st, fa, sa, tl);
    else if (md->norm == 3) launch_step1<KGE_TRANSE, 3>(ld->loss, nv, grid, st, fa, sa, tl);
    else launch_step1<KGE_TRANSE, 2>(ld->loss, nv, grid, st, fa, sa, tl);
  }
  KGE_HIP_TRY(hipGetLastError());
  if (!tickets) {
    const LossFinish lf{loss_partials, rel_blocks, loss_out, adv};
    return rel_tail_launch(tail, fa.Rel, lf, st);
  }
  return KGE_OK;
}

}  // namespace kge
