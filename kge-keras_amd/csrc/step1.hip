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

template <int MODEL, int NORM, int LOSS, int NV, int G, bool DROP>
__global__ __launch_bounds__(256) void step1_kernel(FwdRelArgs fa, SelfArgs sa, float* __restrict__ loss_partials, int rel_blocks, int R) {
  constexpr int NACC = MODEL == KGE_COMPLEX ? 2 * NV : NV;
  __shared__ float4 selfrow[4][NACC * G];
  __shared__ float red[16];
  const int wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= rel_blocks) {
    pull_self_unit<MODEL, NORM, LOSS, NV, G, DROP>(sa, ((int)blockIdx.x - rel_blocks) * 4 + wave, selfrow[wave]);
    return;
  }
  // lr_t of this update for the closing launch (rel_tail_kernel), whose last workgroup overwrites the state word
  if (blockIdx.x == 0 && threadIdx.x == 0 && sa.lr_keep) sa.lr_keep[0] = sa.lr_state[0];
  const int unit = blockIdx.x * 4 + wave;
  float lsum = 0.f;
  int len, c;
  if (unit < fa.rtotal[0] && rel_in_window(fa, unit)) fwd_rel_unit<MODEL, NORM, LOSS, NV, DROP, G>(fa, unit, R, lsum, &len, &c);
  const float b = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = b;
}

// Sub-group width: rows of up to 32 float4 chunks on 16 lanes (NV = 1, 2).  Wider rows (33..64 chunks, d = 200): 16 lanes x NV = 4
// or 32 lanes x NV = 2 -- the same instruction count per row, half the row-sized register arrays.  Measured at the FB15k-237 shape:
// DistMult 65.7 us on 16 lanes (93 VGPRs, 5 waves per SIMD) against 67.2 us on 32 (59 VGPRs, 8 waves: the launch is bound by the
// fabric, not by occupancy); TransE-L2 91.6 us (98 VGPRs, 4 waves) against 75.4 us.  So: DistMult 16 lanes, the others 32.
template <int MODEL, int NORM, int LOSS>
static void launch_step1_nv(int nchunk, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, float* lp, int rb, int R) {
  const bool drop = fa.thr != 0u;
#define KGE_S1(NVV, GW) do { if (drop) hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, NVV, GW, true>), dim3(grid), dim3(256), 0, st, fa, sa, lp, rb, R); \
                             else hipLaunchKernelGGL((step1_kernel<MODEL, NORM, LOSS, NVV, GW, false>), dim3(grid), dim3(256), 0, st, fa, sa, lp, rb, R); } while (0)
  if (nchunk <= 16) KGE_S1(1, 16);
  else if (nchunk <= 32) KGE_S1(2, 16);
  else if (MODEL == KGE_DISTMULT) KGE_S1(4, 16);
  else KGE_S1(2, 32);
#undef KGE_S1
}
template <int MODEL, int NORM>
static void launch_step1(int loss, int nchunk, int grid, hipStream_t st, const FwdRelArgs& fa, const SelfArgs& sa, float* lp, int rb, int R) {
  if (loss == KGE_POINTWIZE_HINGE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_HINGE>(nchunk, grid, st, fa, sa, lp, rb, R);
  else if (loss == KGE_POINTWIZE_LOGISTIC) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_LOGISTIC>(nchunk, grid, st, fa, sa, lp, rb, R);
  else if (loss == KGE_POINTWIZE_SQUARE) launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_SQUARE>(nchunk, grid, st, fa, sa, lp, rb, R);
  else launch_step1_nv<MODEL, NORM, KGE_POINTWIZE_CROSS_ENTROPY>(nchunk, grid, st, fa, sa, lp, rb, R);
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
  const int grid = rel_blocks + (int)cdiv(sa.row_n + e.max_extra, 4);  // (row_n: the rows of the step's entity window; all of them by default)
  {
    ProfScope prof(KGE_PROF_TRAIN_GRADS, st);
    if (md->model == KGE_DISTMULT) launch_step1<KGE_DISTMULT, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R);
    else if (md->model == KGE_COMPLEX) launch_step1<KGE_COMPLEX, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R);
    else if (md->norm == 1) launch_step1<KGE_TRANSE, 1>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R);
    else if (md->norm == 3) launch_step1<KGE_TRANSE, 3>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R);
    else launch_step1<KGE_TRANSE, 2>(ld->loss, fa.nchunk, grid, st, fa, sa, loss_partials, rel_blocks, tail.R);
  }
  KGE_HIP_TRY(hipGetLastError());
  const LossFinish lf{loss_partials, rel_blocks, loss_out, adv};
  return rel_tail_launch(tail, fa.Rel, lf, st);
}

}  // namespace kge
