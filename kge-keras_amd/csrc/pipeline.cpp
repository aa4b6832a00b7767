// Native driver of the training pipeline (see include/kge_hip.h, "training pipeline").  What Keras' fit() does for the reference
// with prefetching worker threads (tests.py:130-134) is here one host loop in C over four device streams:
//
//   gen   G(k)  extend the two MT19937 word rings by one round if the consumers are close to their heads   (ahead of need)
//   draw  S(k)  batch k: masked-rejection select, choices, assembly -> candidate rows + pick counts          (needs G(k))
//   plan  P(k)  ids-only half of step k: histograms, list offsets, relation lists                            (needs S(k))
//   step  T(k)  forward, loss, gradients, optimizer -- between twin tables where possible: two launches         (needs P(k))
//
// double-buffered batches and plans; per tick the loop issues T(t) and then G, S, P of batch t+2.  By default every stage is
// launched kernel by kernel from this loop (about 15 launches and a dozen event operations per step, ~0.05 ms of host time: the
// device, not the host, sets the pace).  KGE_PIPE_GRAPHS=1 replays each stage from a hipGraph captured once per buffer parity
// instead; on ROCm 7.2 consecutive graph launches on one stream cost 20-30 us of idle time each, which makes that the slower form
// (measured: 0.162 ms per step against 0.150 ms).  The draw and plan streams are confined to half of the compute units of every
// shader engine (see kge_pipeline_create).  Host-only translation unit.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <cstdlib>
#include <new>
#include <vector>

#include "kge_host.hpp"

namespace kge {

// sampler_dev.hip
int sampler_stream_init(void* buf, int64_t bytes, const void* np_state, const void* py_state, int64_t B, int64_t ns, int64_t num_e, hipStream_t st);
int sampler_stream_produce(void* buf, int64_t bytes, int64_t B, int64_t ns, int64_t num_e, const void* plan_dev, hipStream_t st);
int sampler_stream_draw(void* buf, int64_t bytes, void* np_state, void* py_state, const int32_t* batch, const int32_t* kg, const int64_t* index,
                        const int64_t* starts, int64_t* cursor, int64_t B, int64_t ns, int64_t num_e, int32_t* pos_out, int32_t* neg_out,
                        int32_t* cand_rows, int32_t* cand_cnt, uint32_t* err, hipStream_t st);
int64_t sampler_stream_bytes(int64_t B, int64_t ns, int64_t num_e);
int sampler_stream_round();

namespace {

#define KGE_PIPE_TRY(expr)                                                                                  \
  do {                                                                                                      \
    hipError_t e__ = (expr);                                                                                \
    if (e__ != hipSuccess) return ::kge::fail(KGE_EHIP, "%s: %s", #expr, hipGetErrorString(e__));           \
  } while (0)

constexpr int kStages = 4;
enum { GEN = 0, DRAW = 1, PLAN = 2, STEP = 3 };

struct Pipeline {
  kge_pipeline_desc d;
  hipStream_t s_gen = nullptr, s_draw = nullptr, s_plan = nullptr;
  bool borrowed[3] = {false, false, false};  // the stream is one of the process-wide CU-masked streams: not destroyed with the pipeline
  static constexpr int kMaxPlans = 6, kRecs = 8;
  hipEvent_t gen[2] = {}, ready[2] = {}, planned[kMaxPlans] = {}, pfreed[kMaxPlans] = {};  // (planned / pfreed: per plan buffer)
  void* plan_buf[kMaxPlans] = {};  // plan[0], plan[1], plan3, plan_more[...]
  int np = 2;                      // plan buffers in use: batch k is planned into buffer k % np, which step k - np read last
  // "plan consumed" records of the step stream: behind every rec_every-th step (rec_every = max(1, np - 2)); the plan of batch k waits
  // for the first record at or behind step k - np -- one of the steps k - np .. k - 3, i.e. never later than what three buffers with a
  // record per step wait for.  Every kge_pipeline_run ends with a record behind its last step, so nothing ever waits for a record
  // that a later call would have to make.  (More than three buffers were an experiment -- fewer records on the step stream -- and
  // measured slower: 0.0999 / 0.1005 / 0.102 / 0.104 ms per step with 3 / 4 / 5 / 6 buffers of 53 MB; the default stays three.)
  int rec_every = 1;
  struct StepRec { int64_t step; hipEvent_t ev; };
  StepRec recs[kRecs] = {};        // ring of the last records (direct launches; the captured form keeps a record per step and buffer)
  hipEvent_t rec_pool[kRecs] = {};
  int64_t n_recs = 0, last_rec_step = -1;
  hipEvent_t fork = nullptr;
  hipStream_t s_comm = nullptr;    // data-parallel steps: the stream the all-reduces run on (beside the relation half of the gradients)
  hipEvent_t half = nullptr, grads = nullptr, reduced = nullptr;
  bool dp_halves = false;          // the gradient half comes in two independent halves (kge_train_grads_halves_supported)
  hipEvent_t done[kStages][2] = {};      // graphs only: behind the last launch of graph (stage, slot)
  bool launched[kStages][2] = {};
  hipGraphExec_t exec[kStages][2] = {};
  bool graphs = false;             // replay the stages from hipGraphs instead of launching their kernels from the loop
  bool ring = false;               // word generation ahead of the batches (kge_sampler_stream_*); else kge_build_batch_device_indexed
  int64_t n_batches = 0;           // batches of the loaded epoch
  int64_t t = 0;                   // ticks done
  int64_t nround = 0;              // production rounds offered in this epoch
  bool started = false;
  // host-side accounting of kge_pipeline_run (kge_pipeline_stats): where the issuing thread spends its time
  double ns_spin = 0.0, ns_launch = 0.0, ns_total = 0.0, ns_spin_stage[kStages] = {};
  int64_t n_launch = 0, n_spun = 0;
  // stage timeline (kge_pipeline_trace): timing events around every stage launch of the next `trace_left` launches
  struct Rec { int stage; int64_t batch; hipEvent_t a, b; };
  std::vector<Rec> trace;
  int64_t trace_left = 0;
  bool arrays = false;             // batches as plain oversampled arrays (dropout): no folded form, plan from the arrays
  uint64_t seed = 0;               // arrays: dropout seed of the last step issued (kge_pipeline_set_seed)
  bool swap = false;               // steps alternate between the caller's tables and their twins (kge_train_step_swap)
  bool chain = false;              // ... as chained steps (kge_train_step_swap_chain: one launch per step); KGE_PIPE_CHAIN=0 keeps the two-launch step
  int64_t chain_seq = 0;           // steps issued in the running chain (a chain lives inside one kge_pipeline_run: seq 0 refills the lr ring)
  int parity = 0;                  // swap: 1 = the current tables are the twins
  int stage_mask = 15;  // experiments only (KGE_PIPE_STAGES): bit s clear = stage s is not launched (its events still are recorded)
};

inline double now_ns() { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// slot: DRAW the candidate / array buffer (batch & 1); PLAN that buffer in the low nibble, the plan buffer (batch % np) above it;
// STEP the plan buffer
int stage_calls(Pipeline* p, int stage, int slot, hipStream_t st) {
  const kge_pipeline_desc& d = p->d;
  const int64_t m = d.B * d.ns;
  const int pslot = stage == PLAN && !p->graphs ? slot >> 4 : slot;  // (captured stages: two plan buffers, one index)
  if (stage == PLAN && !p->graphs) slot &= 15;
  void* const plan_buf = p->plan_buf[pslot];
  switch (stage) {
    case GEN:
      return p->ring ? sampler_stream_produce(d.sampler_stream, d.sampler_stream_bytes, d.B, d.ns, d.num_e, d.sampler_stream_plan, st) : KGE_OK;
    case DRAW: {  // positives read in place; folded output (candidate rows + pick counts) or, arrays mode, the oversampled arrays;
                  // the last kernel moves the cursor
      int32_t* po = p->arrays ? d.pos[slot] : nullptr;
      int32_t* ne = p->arrays ? d.neg[slot] : nullptr;
      int32_t* cr = p->arrays ? nullptr : d.cand_rows[slot];
      int32_t* cc = p->arrays ? nullptr : d.cand_counts[slot];
      if (p->ring)
        return sampler_stream_draw(d.sampler_stream, d.sampler_stream_bytes, d.np_state, d.py_state, nullptr, d.kg, d.index, d.starts, d.cursor, d.B, d.ns,
                                   d.num_e, po, ne, cr, cc, d.sampler_err, st);
      return kge_build_batch_device_indexed(d.np_state, d.py_state, d.kg, d.index, d.starts, d.cursor, d.B, d.ns, d.num_e, po, ne, cr, cc, d.sampler_plan,
                                            d.sampler_ws, d.sampler_ws_bytes, d.sampler_err, st);
    }
    case PLAN:  // (8: the plan buffers' counters are zero -- zero-filled by the caller, left zero by every step)
      if (p->arrays)
        return kge_batch_plan_build(&d.md, &d.ld, d.pos[slot], m, d.neg[slot], m, plan_buf, d.plan_bytes, 7 | 8, d.err_flag, st);
      return kge_batch_plan_build_counted(&d.md, &d.ld, d.cand_rows[slot], d.cand_counts[slot], KGE_CAND_POS_LANES, d.B, m, m, m, plan_buf,
                                          d.plan_bytes, 7 | 8, d.err_flag, st);
    default:
      if (d.comm) {
        // data-parallel step: [entity half | all-reduce of gradE beside the relation half] -> all-reduce of the rest -> optimizer
        const uint64_t seed = p->arrays ? ++p->seed : 0;
        const int64_t nE = (int64_t)d.md.num_entities * d.md.d_e;
        const float scale = 1.f / (float)d.world;
        if (p->dp_halves) {
          if (int rc = kge_train_grads_planned_parts(&d.md, &d.ld, d.E, d.Rel, d.gradE, d.gradRel, plan_buf, d.plan_bytes, m, m, d.loss_out, seed, d.train_ws,
                                                     d.train_ws_bytes, 1, st)) return rc;
          KGE_PIPE_TRY(hipEventRecord(p->half, st));
          KGE_PIPE_TRY(hipStreamWaitEvent(p->s_comm, p->half, 0));
          if (int rc = kge_allreduce_f32(d.comm, d.grad_flat, nE, p->s_comm)) return rc;
          if (int rc = kge_train_grads_planned_parts(&d.md, &d.ld, d.E, d.Rel, d.gradE, d.gradRel, plan_buf, d.plan_bytes, m, m, d.loss_out, seed, d.train_ws,
                                                     d.train_ws_bytes, 2, st)) return rc;
          KGE_PIPE_TRY(hipEventRecord(p->grads, st));
          KGE_PIPE_TRY(hipStreamWaitEvent(p->s_comm, p->grads, 0));
          if (int rc = kge_allreduce_f32(d.comm, d.grad_flat + nE, d.grad_flat_n - nE, p->s_comm)) return rc;
          KGE_PIPE_TRY(hipEventRecord(p->reduced, p->s_comm));
          KGE_PIPE_TRY(hipStreamWaitEvent(st, p->reduced, 0));
        } else {
          if (int rc = kge_train_grads_planned(&d.md, &d.ld, d.E, d.Rel, d.gradE, d.gradRel, plan_buf, d.plan_bytes, m, m, d.loss_out, seed, d.train_ws,
                                               d.train_ws_bytes, st)) return rc;
          if (int rc = kge_allreduce_f32(d.comm, d.grad_flat, d.grad_flat_n, st)) return rc;
        }
        return kge_opt_apply(&d.md, &d.ld, &d.od, d.E, d.Rel, d.gradE, d.gradRel, scale, d.loss_out, d.train_ws, d.train_ws_bytes, st);
      }
      if (p->swap) {  // (the caller moves the parity on; lr_t of the update comes from the device-resident step state)
        const int q = p->parity;
        const int parts = 3;
        if (p->chain)  // one launch per step: relation rows, loss and step state closed by late workgroups of the step's own launch
          return kge_train_step_swap_chain(&d.md, &d.ld, &d.od, q ? d.E_twin : d.E, q ? d.Rel_twin : d.Rel, q ? d.E : d.E_twin, q ? d.Rel : d.Rel_twin,
                                           plan_buf, d.plan_bytes, m, m, d.loss_out, p->arrays ? ++p->seed : 0, d.train_ws, d.train_ws_bytes, p->chain_seq++,
                                           d.err_flag, st);
        return kge_train_step_swap(&d.md, &d.ld, &d.od, q ? d.E_twin : d.E, q ? d.Rel_twin : d.Rel, q ? d.E : d.E_twin, q ? d.Rel : d.Rel_twin, plan_buf,
                                   d.plan_bytes, m, m, d.loss_out, p->arrays ? ++p->seed : 0, d.train_ws, d.train_ws_bytes, parts, st);
      }
      return kge_train_step_planned(&d.md, &d.ld, &d.od, d.E, d.Rel, d.gradE, d.gradRel, plan_buf, d.plan_bytes, m, m, d.loss_out,
                                    p->arrays ? ++p->seed : 0, d.train_ws, d.train_ws_bytes, st);
  }
}

int capture(Pipeline* p, int stage, int slot, hipStream_t st) {
  hipGraph_t g = nullptr;
  KGE_PIPE_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const int rc = stage_calls(p, stage, slot, st);
  const hipError_t e = hipStreamEndCapture(st, &g);
  if (rc != KGE_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return fail(KGE_EHIP, "hipStreamEndCapture (stage %d, slot %d): %s", stage, slot, hipGetErrorString(e));
  const hipError_t ei = hipGraphInstantiate(&p->exec[stage][slot], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ei != hipSuccess) return fail(KGE_EHIP, "hipGraphInstantiate (stage %d, slot %d): %s", stage, slot, hipGetErrorString(ei));
  return KGE_OK;
}

int launch(Pipeline* p, int stage, int slot, hipStream_t st, int64_t batch = -1) {
  if (!((p->stage_mask >> stage) & 1)) return KGE_OK;
  if (stage == GEN && !p->ring) return KGE_OK;
  if (!p->graphs) {
    Pipeline::Rec rec{stage, batch, nullptr, nullptr};
    const bool tr = p->trace_left > 0 && hipEventCreate(&rec.a) == hipSuccess && hipEventCreate(&rec.b) == hipSuccess;
    if (tr) (void)hipEventRecord(rec.a, st);
    const double t0 = now_ns();
    if (int rc = stage_calls(p, stage, slot, st)) return rc;
    p->ns_launch += now_ns() - t0;
    p->n_launch += 1;
    if (tr) { (void)hipEventRecord(rec.b, st); p->trace.push_back(rec); p->trace_left -= 1; }
    return KGE_OK;
  }
  // A graph exec cannot run twice at once: launching it again while its previous launch is in flight blocks inside the runtime
  // with a sleeping wait (milliseconds on a busy host).  Spin on the event behind the previous launch instead -- bounded: a stage
  // that never finishes (a fault on the device) becomes an error, not a hang.
  const double ta = now_ns();
  if (p->launched[stage][slot]) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
      const hipError_t q = hipEventQuery(p->done[stage][slot]);
      if (q == hipSuccess) { if (spins) p->n_spun += 1; break; }
      if (q != hipErrorNotReady) return fail(KGE_EHIP, "hipEventQuery (stage %d, slot %d): %s", stage, slot, hipGetErrorString(q));
      if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
        return fail(KGE_EHIP, "kge_pipeline: stage %d of slot %d did not finish within 20 s", stage, slot);
    }
  }
  const double tb = now_ns();
  KGE_PIPE_TRY(hipGraphLaunch(p->exec[stage][slot], st));
  KGE_PIPE_TRY(hipEventRecord(p->done[stage][slot], st));
  p->launched[stage][slot] = true;
  p->ns_spin += tb - ta;
  p->ns_spin_stage[stage] += tb - ta;
  p->ns_launch += now_ns() - tb;
  p->n_launch += 1;
  return KGE_OK;
}

// One production round of the word rings (R batches' worth of words) on the generator stream.
int offer_round(Pipeline* p, int after_ready) {
  if (after_ready >= 0) KGE_PIPE_TRY(hipStreamWaitEvent(p->s_gen, p->ready[after_ready], 0));
  if (int rc = launch(p, GEN, 0, p->s_gen, p->nround)) return rc;
  KGE_PIPE_TRY(hipEventRecord(p->gen[p->nround & 1], p->s_gen));
  p->nround += 1;
  return KGE_OK;
}

// generator round, draw and plan of batch k on their streams
int produce(Pipeline* p, int64_t k) {
  if (k >= p->n_batches) return KGE_OK;
  const int j = (int)(k & 1);
  if (p->ring && k % sampler_stream_round() == 0) {
    // The draws of batches k .. k+R-1 need the rounds offered so far -- the last of them one period ago (at the start for k = 0) --
    // and the round offered now has until batch k+R to finish: generation never sits on a batch's critical path.  The new round
    // sees the consumer position of batch k-2 at the latest (`ready[j]` still is the record behind draw k-2): stale by one batch at
    // most, which is what the rings' low-water mark is sized for (sampler_dev.hip).
    KGE_PIPE_TRY(hipStreamWaitEvent(p->s_draw, p->gen[(p->nround - 1) & 1], 0));
    if (int rc = offer_round(p, j)) return rc;
  }
  const int pj = (int)(k % p->np);
  KGE_PIPE_TRY(hipStreamWaitEvent(p->s_draw, p->planned[(int)((k + p->np - 2) % p->np)], 0));  // (still the record behind plan k-2: its candidate buffers are free)
  if (!p->graphs && k - p->np >= 0) {
    // the first record at or behind step k - np (see Pipeline::rec_every): the step that consumed the plan buffer of batch k.  The DRAW
    // stream waits for it (it has half a step of slack per batch) and the plan, which waits for the draw, is ordered behind it that way:
    // one barrier packet less on the plan stream, which paces the pipeline once the step is a single launch.
    const int64_t need = k - p->np;
    const int64_t first = std::max<int64_t>(0, p->n_recs - Pipeline::kRecs);
    bool found = false;
    for (int64_t r = first; r < p->n_recs && !found; ++r)
      if (p->recs[r % Pipeline::kRecs].step >= need) { KGE_PIPE_TRY(hipStreamWaitEvent(p->s_draw, p->recs[r % Pipeline::kRecs].ev, 0)); found = true; }
    if (!found) return fail(KGE_EINVAL, "kge_pipeline: no record of step %lld for the plan of batch %lld (internal)", (long long)need, (long long)k);
  }
  if (int rc = launch(p, DRAW, j, p->s_draw, k)) return rc;
  KGE_PIPE_TRY(hipEventRecord(p->ready[j], p->s_draw));
  hipStream_t s_plan = p->s_plan;
  KGE_PIPE_TRY(hipStreamWaitEvent(s_plan, p->ready[j], 0));
  if (p->graphs) {
    KGE_PIPE_TRY(hipStreamWaitEvent(s_plan, p->pfreed[pj], 0));  // (the step that consumed this plan buffer: k-2)
  }
  if (int rc = launch(p, PLAN, p->graphs ? j : (j | (pj << 4)), s_plan, k)) return rc;
  KGE_PIPE_TRY(hipEventRecord(p->planned[pj], s_plan));
  return KGE_OK;
}

void destroy(Pipeline* p) {
  for (int s = 0; s < kStages; ++s)
    for (int j = 0; j < 2; ++j) {
      if (p->exec[s][j]) (void)hipGraphExecDestroy(p->exec[s][j]);
      if (p->done[s][j]) (void)hipEventDestroy(p->done[s][j]);
    }
  for (int j = 0; j < 2; ++j)
    for (hipEvent_t e : {p->gen[j], p->ready[j]})
      if (e) (void)hipEventDestroy(e);
  for (int j = 0; j < Pipeline::kMaxPlans; ++j)
    for (hipEvent_t e : {p->planned[j], p->pfreed[j]})
      if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->rec_pool)
    if (e) (void)hipEventDestroy(e);
  if (p->fork) (void)hipEventDestroy(p->fork);
  for (hipEvent_t e : {p->half, p->grads, p->reduced})
    if (e) (void)hipEventDestroy(e);
  if (p->s_comm) (void)hipStreamDestroy(p->s_comm);
  hipStream_t owned[3] = {p->s_gen, p->s_draw, p->s_plan};
  for (int i = 0; i < 3; ++i)
    if (owned[i] && !p->borrowed[i]) (void)hipStreamDestroy(owned[i]);
  delete p;
}

}  // namespace

}  // namespace kge

using namespace kge;

extern "C" {

int kge_pipeline_create(const kge_pipeline_desc* desc, kge_stream_t stream, void** handle) {
  (void)hipGetLastError();
  if (!desc || !handle) return fail(KGE_EINVAL, "kge_pipeline_create: null argument");
  const kge_pipeline_desc& d = *desc;
  if (!d.od.step_dev) return fail(KGE_EINVAL, "kge_pipeline_create: the optimizer needs its device-resident step state (kge_opt_desc.step_dev)");
  const int n_arr = (d.pos[0] != nullptr) + (d.pos[1] != nullptr) + (d.neg[0] != nullptr) + (d.neg[1] != nullptr);
  if (n_arr != 0 && n_arr != 4) return fail(KGE_EINVAL, "kge_pipeline_create: pos / neg arrays: all four or none");
  const bool arrays = n_arr == 4;
  if (d.md.dropout > 0.f && !arrays)
    return fail(KGE_EUNSUPPORTED, "kge_pipeline_create: folded batches and dropout exclude each other (dropout needs the pos / neg arrays of the descriptor)");
  if (!d.E || !d.Rel || !d.gradE || !d.gradRel || !d.loss_out || !d.train_ws || !d.err_flag || !d.np_state || !d.py_state || !d.kg || !d.index ||
      !d.starts || !d.cursor || !d.sampler_ws || !d.sampler_err || !d.plan[0] || !d.plan[1] ||
      (!arrays && (!d.cand_rows[0] || !d.cand_rows[1] || !d.cand_counts[0] || !d.cand_counts[1])))
    return fail(KGE_EINVAL, "kge_pipeline_create: null pointer in the descriptor");
  if (d.B <= 0 || d.ns <= 0 || d.num_e <= 0) return fail(KGE_EINVAL, "kge_pipeline_create: bad batch shape");
  if (d.comm) {
    const int64_t nE = (int64_t)d.md.num_entities * d.md.d_e, nR = (int64_t)d.md.num_relations * d.md.d_r;
    if (d.world < 1 || !d.grad_flat || d.grad_flat != d.gradE || d.grad_flat_n < nE + nR || d.gradRel < d.grad_flat + nE ||
        d.gradRel + nR > d.grad_flat + d.grad_flat_n || d.loss_out < d.grad_flat + nE || d.loss_out >= d.grad_flat + d.grad_flat_n)
      return fail(KGE_EINVAL, "kge_pipeline_create: with comm, grad_flat = gradE must cover gradE, gradRel and the loss word (world >= 1)");
  }
  const int64_t need = kge_batch_plan_bytes(&d.md, &d.ld, d.B * d.ns, d.B * d.ns);
  if (need < 0) return (int)need;
  if (d.plan_bytes < need) return fail(KGE_EWORKSPACE, "kge_pipeline_create: plan buffers have %lld bytes, need %lld", (long long)d.plan_bytes, (long long)need);
  if (d.sampler_stream && d.sampler_stream_bytes < sampler_stream_bytes(d.B, d.ns, d.num_e))
    return fail(KGE_EWORKSPACE, "kge_pipeline_create: sampler stream buffer has %lld bytes, needs %lld", (long long)d.sampler_stream_bytes,
                (long long)sampler_stream_bytes(d.B, d.ns, d.num_e));
  Pipeline* p = new (std::nothrow) Pipeline();
  if (!p) return fail(KGE_EINVAL, "kge_pipeline_create: out of host memory");
  p->d = d;
  p->arrays = arrays;
  p->ring = d.sampler_stream != nullptr;
  if (const char* e = std::getenv("KGE_PIPE_STAGES")) p->stage_mask = std::atoi(e) & 15;
  if (const char* e = std::getenv("KGE_PIPE_GRAPHS")) p->graphs = std::atoi(e) != 0;
  if (arrays) p->graphs = false;  // (a captured step would replay one dropout seed)
  // a third plan buffer (direct launches only: the captured stages are per buffer pair): the plan of batch k+2 then waits for step
  // k-1 instead of step k, so at the end of a run no plan stage is left behind the last step (measured at --steps 20: see DESIGN 3.4)
  {
    int avail = 2;
    p->plan_buf[0] = d.plan[0]; p->plan_buf[1] = d.plan[1];
    if (d.plan3) {
      p->plan_buf[avail++] = d.plan3;
      for (void* b : d.plan_more) { if (!b || avail == Pipeline::kMaxPlans) break; p->plan_buf[avail++] = b; }
    }
    p->np = p->graphs ? 2 : avail;
    p->rec_every = std::max(1, p->np - 2);
  }
  // twin tables: the step closed by its pull launch (direct launches only: a captured step has its table pointers baked in)
  if ((d.E_twin != nullptr) != (d.Rel_twin != nullptr)) { delete p; return fail(KGE_EINVAL, "kge_pipeline_create: E_twin and Rel_twin go together"); }
  // (with dropout the twin step is available but measured slower than the three-launch step -- 0.46 against 0.39 ms per step at the
  // bench shape, where nothing is folded: 409 600 rows, three masks per element and entry in a kernel that is already register bound)
  p->swap = !d.comm && d.E_twin && !p->graphs && d.md.dropout == 0.f && kge_train_step_swap_supported(&d.md, &d.ld, &d.od) == 1;
  { const char* e = std::getenv("KGE_PIPE_CHAIN"); p->chain = p->swap && !(e && std::atoi(e) == 0); }
  if (d.comm) p->graphs = false;  // (a collective never sits inside a captured graph)
  p->dp_halves = d.comm && kge_train_grads_halves_supported(&d.md, &d.ld) == 1;
  if (const char* e = std::getenv("KGE_PIPE_DP_HALVES")) { if (std::atoi(e) == 0) p->dp_halves = false; }  // (tests run both forms)
  if (const char* e = std::getenv("KGE_PIPE_SWAP")) { if (std::atoi(e) == 0) p->swap = false; }
  if (p->graphs && prof_is_on()) {
    delete p;
    return fail(KGE_EINVAL, "kge_pipeline_create: switch kernel timing off first (kge_prof_enable(0)): timing events cannot be captured");
  }
  auto bail = [&](int rc) { destroy(p); return rc; };
#define KGE_CREATE_TRY(expr)                                                                                                    \
  do {                                                                                                                          \
    hipError_t e__ = (expr);                                                                                                    \
    if (e__ != hipSuccess) return bail(fail(KGE_EHIP, "%s: %s", #expr, hipGetErrorString(e__)));                                \
  } while (0)
  {
    // the producer stages run below the training stream's priority: the step is the critical chain, everything else has a batch
    // or more of slack
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = numerically greatest = lowest priority
    const int prio = lo;
    // The producer streams are confined to half of the compute units -- in EVERY shader engine of every XCD: the queue's CU mask
    // numbers the units XCD-first, then shader engine, then unit (32 consecutive bits = unit i of all 8 x 4 engines), so "every other
    // group of 32 bits" keeps units 0, 2, 4, 6 of each engine.  The step's waves then find half of the units free of producer waves
    // (which hold registers and LDS for tens of microseconds): 0.1126 -> 0.1040 ms per step at the bench shape; every other XCD, or
    // every other engine, does not help (profiles/r02_experiments.txt).  Such a stream (hipExtStreamCreateWithCUMask) has the default
    // priority and is a BLOCKING stream: it synchronises with the legacy default stream, so this form needs the caller on a stream of
    // its own (kge_pipeline_create's `stream` != 0; pipeline.py drives the loop from one); on the default stream, and with
    // KGE_PIPE_CUMASK=0, the producers are plain low-priority streams over all units.
    // (KGE_PIPE_CUMASK=0 switches the confinement off, k > 1 keeps one of every k groups of 32 mask bits.)
    const char* em = std::getenv("KGE_PIPE_CUMASK");
    const int stride = em ? std::atoi(em) : (stream ? 2 : 0);
    if (stride > 1) {
      const int group = 32, keep = 1;
      hipDeviceProp_t prop;
      int dev = 0;
      KGE_CREATE_TRY(hipGetDevice(&dev));  // (one process per GPU: the rank's own device)
      KGE_CREATE_TRY(hipGetDeviceProperties(&prop, dev));
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
      for (int c = 0; c < ncu; ++c) if ((c / group) % stride < keep) mask[(size_t)c / 32] |= 1u << (c % 32);
      // which producer streams are confined (1 generator rounds, 2 draw, 4 plan): draw and plan -- the rounds run thin and long, cost
      // the step nothing where they are, and sit on the critical path of an epoch's first batches
      const int which = 6;
      // CU-masked streams are made once per process and shared by all pipelines: a masked queue that is destroyed goes back to the
      // runtime's queue pool with its mask still set and is handed to the next stream that is created -- measured on ROCm 7.2: the
      // steps of a second pipeline ran at half speed after the first one had been destroyed (profiles/r02_experiments.txt)
      // One set per device (a stream belongs to the device that was current when it was made).  Pipelines that are alive at the same
      // time on one device share these producer streams, i.e. their draw / plan stages queue behind each other: correct, not concurrent.
      struct SharedSet { hipStream_t s[3] = {nullptr, nullptr, nullptr}; int key[3] = {0, 0, 0}; };
      static std::map<int, SharedSet> shared_by_device;
      static std::mutex shared_mutex;  // (pipelines may be created from several host threads)
      std::lock_guard<std::mutex> lock(shared_mutex);
      SharedSet& sh = shared_by_device[dev];
      const int key = stride * 1000000 + group * 1000 + keep;
      hipStream_t* ss[3] = {&p->s_gen, &p->s_draw, &p->s_plan};
      for (int i = 0; i < 3; ++i) {
        if ((which >> i) & 1) {
          if (!sh.s[i] || sh.key[i] != key) {
            if (hipExtStreamCreateWithCUMask(&sh.s[i], (uint32_t)mask.size(), mask.data()) != hipSuccess) {
              // (a runtime or partition mode without CU masks: the plain low-priority stream instead)
              (void)hipGetLastError();
              sh.s[i] = nullptr;
              KGE_CREATE_TRY(hipStreamCreateWithPriority(ss[i], hipStreamNonBlocking, prio));
              continue;
            }
            sh.key[i] = key;
            // gone before the runtime's own teardown (handlers run in reverse order of registration; the runtime registered its own
            // when it was loaded): a masked stream still alive at that point crashed the process under rocprofv3
            static bool registered = false;
            if (!registered) {
              registered = true;
              std::atexit([] {
                for (auto& kv : shared_by_device)
                  for (hipStream_t& s : kv.second.s)
                    if (s) { (void)hipStreamDestroy(s); s = nullptr; }
              });
            }
          }
          *ss[i] = sh.s[i];
          p->borrowed[i] = true;
        } else {
          KGE_CREATE_TRY(hipStreamCreateWithPriority(ss[i], hipStreamNonBlocking, prio));
        }
      }
    } else {
      KGE_CREATE_TRY(hipStreamCreateWithPriority(&p->s_gen, hipStreamNonBlocking, prio));
      KGE_CREATE_TRY(hipStreamCreateWithPriority(&p->s_draw, hipStreamNonBlocking, prio));
      KGE_CREATE_TRY(hipStreamCreateWithPriority(&p->s_plan, hipStreamNonBlocking, prio));
    }
  }
  KGE_CREATE_TRY(hipEventCreateWithFlags(&p->fork, hipEventDisableTiming));
  if (d.comm) {
    int lo2 = 0, hi2 = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo2, &hi2);  // the exchange is on the step's critical path: highest priority
    KGE_CREATE_TRY(hipStreamCreateWithPriority(&p->s_comm, hipStreamNonBlocking, hi2));
    for (hipEvent_t* e : {&p->half, &p->grads, &p->reduced}) KGE_CREATE_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  for (int j = 0; j < Pipeline::kMaxPlans; ++j)
    for (hipEvent_t* e : {&p->planned[j], &p->pfreed[j]}) KGE_CREATE_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  for (hipEvent_t& e : p->rec_pool) KGE_CREATE_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int j = 0; j < 2; ++j) {
    for (hipEvent_t* e : {&p->gen[j], &p->ready[j]}) KGE_CREATE_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int s = 0; s < kStages; ++s) KGE_CREATE_TRY(hipEventCreateWithFlags(&p->done[s][j], hipEventDisableTiming));
  }
#undef KGE_CREATE_TRY
  if (p->graphs) {  // capture on streams of the library's own (the caller's may be the legacy default stream, which cannot be captured)
    for (int j = 0; j < 2; ++j) {
      if (p->ring) if (int rc = capture(p, GEN, j, p->s_gen)) return bail(rc);
      if (int rc = capture(p, DRAW, j, p->s_draw)) return bail(rc);
      if (int rc = capture(p, PLAN, j, p->s_plan)) return bail(rc);
      if (int rc = capture(p, STEP, j, p->s_draw)) return bail(rc);
    }
  }
  *handle = p;
  return KGE_OK;
}

int kge_pipeline_start(void* handle, int64_t n_batches, kge_stream_t stream) {
  (void)hipGetLastError();
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p || n_batches < 0) return fail(KGE_EINVAL, "kge_pipeline_start: bad arguments");
  hipStream_t main_st = static_cast<hipStream_t>(stream);
  const kge_pipeline_desc& d = p->d;
  p->n_batches = n_batches;
  p->t = 0;
  // the rings restart from the generator states the caller loaded (the reference's global generators at the start of an epoch)
  if (p->ring)
    if (int rc = sampler_stream_init(d.sampler_stream, d.sampler_stream_bytes, d.np_state, d.py_state, d.B, d.ns, d.num_e, main_st)) return rc;
  // everything the caller enqueued so far (epoch permutation, cursor reset, generator states) precedes the first draw and plan
  KGE_PIPE_TRY(hipEventRecord(p->fork, main_st));
  for (hipStream_t s : {p->s_gen, p->s_draw, p->s_plan}) KGE_PIPE_TRY(hipStreamWaitEvent(s, p->fork, 0));
  p->started = true;
  p->nround = 0;
  p->n_recs = 0;
  p->last_rec_step = -1;
  if (p->ring) if (int rc = offer_round(p, -1)) return rc;  // the words of batches 0 .. R-1
  if (int rc = produce(p, 0)) return rc;
  return produce(p, 1);
}

int kge_pipeline_set_seed(void* handle, uint64_t seed) {
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p) return fail(KGE_EINVAL, "kge_pipeline_set_seed: null handle");
  p->seed = seed;
  return KGE_OK;
}

int kge_pipeline_run(void* handle, int64_t n_steps, kge_stream_t stream) {
  (void)hipGetLastError();
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p || !p->started || n_steps < 0) return fail(KGE_EINVAL, "kge_pipeline_run: not started / bad arguments");
  if (p->t + n_steps > p->n_batches)
    return fail(KGE_EINVAL, "kge_pipeline_run: %lld steps asked, %lld batches left in the epoch", (long long)n_steps, (long long)(p->n_batches - p->t));
  hipStream_t main_st = static_cast<hipStream_t>(stream);
  if (const char* e = std::getenv("KGE_PIPE_STAGES")) p->stage_mask = std::atoi(e) & 15;  // (experiments: may change between calls)
  const double t_in = now_ns();
  int rc = KGE_OK;
  p->chain_seq = 0;
  for (int64_t i = 0; i < n_steps && rc == KGE_OK; ++i) {
    const int j = (int)(p->t % p->np);
    if (hipStreamWaitEvent(main_st, p->planned[j], 0) != hipSuccess) { rc = fail(KGE_EHIP, "kge_pipeline_run: hipStreamWaitEvent failed"); break; }
    if ((rc = launch(p, STEP, j, main_st, p->t)) != KGE_OK) break;
    if (p->swap) p->parity ^= 1;
    p->t += 1;  // (the step is issued: what follows may fail, the step count and the parity of the tables stay true)
    const int64_t step = p->t - 1;  // the step just issued
    if (p->graphs) {
      if (hipEventRecord(p->pfreed[j], main_st) != hipSuccess) { rc = fail(KGE_EHIP, "kge_pipeline_run: hipEventRecord failed"); break; }
    } else if ((step + 1) % p->rec_every == 0 || i + 1 == n_steps) {
      Pipeline::StepRec& r = p->recs[p->n_recs % Pipeline::kRecs];
      r.step = step;
      r.ev = p->rec_pool[p->n_recs % Pipeline::kRecs];
      if (hipEventRecord(r.ev, main_st) != hipSuccess) { rc = fail(KGE_EHIP, "kge_pipeline_run: hipEventRecord failed"); break; }
      p->n_recs += 1;
      p->last_rec_step = step;
    }
    rc = produce(p, p->t + 1);
  }
  // the caller reads (E, Rel) -- also after a failure half way: the steps issued so far must not be left in the twins
  if (p->swap && p->parity) {
    const kge_pipeline_desc& d = p->d;
    const hipError_t e1 = hipMemcpyAsync(d.E, d.E_twin, (size_t)d.md.num_entities * d.md.d_e * sizeof(float), hipMemcpyDeviceToDevice, main_st);
    const hipError_t e2 = hipMemcpyAsync(d.Rel, d.Rel_twin, (size_t)d.md.num_relations * d.md.d_r * sizeof(float), hipMemcpyDeviceToDevice, main_st);
    if (e1 == hipSuccess && e2 == hipSuccess) p->parity = 0;
    else if (rc == KGE_OK) rc = fail(KGE_EHIP, "kge_pipeline_run: copying the twin tables back failed (kge_pipeline_state reports the parity)");
  }
  p->ns_total += now_ns() - t_in;
  return rc;
}

// out[0..3] = {plan buffers in use (batch k is planned into buffer k % that), steps issued in this epoch, 1 if the current tables are
// the twins (only after a failed kge_pipeline_run), batches of the epoch}.  What a caller needs to take an epoch over from the driver.
int kge_pipeline_state(void* handle, int64_t out[4]) {
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p || !out) return fail(KGE_EINVAL, "kge_pipeline_state: null argument");
  out[0] = p->np; out[1] = p->t; out[2] = p->parity; out[3] = p->n_batches;
  return KGE_OK;
}

int kge_pipeline_stats(void* handle, double out[8], int32_t reset) {
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p || !out) return fail(KGE_EINVAL, "kge_pipeline_stats: null argument");
  out[0] = p->ns_total * 1e-6;   // ms inside kge_pipeline_run
  out[1] = p->ns_spin * 1e-6;    // graphs only: ms of it waiting for a graph's previous launch to finish
  out[2] = p->ns_launch * 1e-6;  // ms of it inside the stage launchers (kernel launches, or hipGraphLaunch + the event record behind it)
  out[3] = (double)p->n_launch;  // stage launches
  out[4] = (double)p->n_spun;    // graphs only: launches that had to wait
  out[5] = p->ns_spin_stage[DRAW] * 1e-6;
  out[6] = p->ns_spin_stage[PLAN] * 1e-6;
  out[7] = p->ns_spin_stage[STEP] * 1e-6;
  if (reset) {
    p->ns_total = p->ns_spin = p->ns_launch = 0.0;
    p->n_launch = p->n_spun = 0;
    for (double& x : p->ns_spin_stage) x = 0.0;
  }
  return KGE_OK;
}

// Stage timeline: n > 0 arms timing events around the next n stage launches; n == 0 reads them back (synchronises): out receives up
// to cap records of {stage (0 rounds, 1 draw, 2 plan, 3 step), batch / round number, begin ms, end ms} relative to the first begin.
int64_t kge_pipeline_trace(void* handle, int64_t n, double* out, int64_t cap) {
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p) return fail(KGE_EINVAL, "kge_pipeline_trace: null handle");
  if (n > 0) { p->trace_left = n; return 0; }
  p->trace_left = 0;
  if (hipDeviceSynchronize() != hipSuccess) return fail(KGE_EHIP, "kge_pipeline_trace: hipDeviceSynchronize failed");
  int64_t k = 0;
  for (const auto& r : p->trace) {
    if (out && k < cap && !p->trace.empty()) {
      float a = 0.f, b = 0.f;
      (void)hipEventElapsedTime(&a, p->trace[0].a, r.a);
      (void)hipEventElapsedTime(&b, p->trace[0].a, r.b);
      out[4 * k] = r.stage; out[4 * k + 1] = (double)r.batch; out[4 * k + 2] = a; out[4 * k + 3] = b;
      ++k;
    }
  }
  for (auto& r : p->trace) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  p->trace.clear();
  return k;
}

int kge_pipeline_finish(void* handle, kge_stream_t stream) {
  (void)hipGetLastError();
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p) return fail(KGE_EINVAL, "kge_pipeline_finish: null handle");
  hipStream_t main_st = static_cast<hipStream_t>(stream);
  // join: the caller's stream waits for whatever the producer streams still hold (batches drawn ahead of the last step)
  for (hipStream_t s : {p->s_gen, p->s_draw, p->s_plan}) {
    KGE_PIPE_TRY(hipEventRecord(p->fork, s));
    KGE_PIPE_TRY(hipStreamWaitEvent(main_st, p->fork, 0));
  }
  p->started = false;
  return KGE_OK;
}

int kge_pipeline_destroy(void* handle) {
  Pipeline* p = static_cast<Pipeline*>(handle);
  if (!p) return KGE_OK;
  (void)hipDeviceSynchronize();
  destroy(p);
  return KGE_OK;
}

}  // extern "C"
