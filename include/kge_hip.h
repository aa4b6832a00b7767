/* kge_hip.h -- C ABI of the MI355X (gfx950) KGE hot path.
 *
 * This is the drop-in boundary: the reference (NIVA-Knowledge-Graph/KGE-Keras) has no FFI of its
 * own -- its hot path is the Python surface KGEkeras.models.* / KGEkeras.utils.* delegating to
 * TensorFlow ops.  Each entry point below names the reference interface it replaces (file:line,
 * relative to the reference repository).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C, no C++ exceptions cross the boundary; every function returns a kge_status
 *     (0 = ok, negative = error) and kge_last_error() returns a thread-local message.
 *   - all `float*` / `int32_t*` arguments are DEVICE pointers (HBM) unless the name ends in `_host`;
 *     the library borrows them (owned by the caller, e.g. torch tensors) and allocates nothing
 *     persistent: scratch is a caller-provided workspace sized by the *_workspace_bytes() query.
 *   - every launcher takes the hipStream_t to enqueue on (as void*), is asynchronous and
 *     graph-capturable (no allocation, no synchronisation inside).  Thread-compatible, not thread-safe.
 *   - tables are row-major fp32: E is (num_entities, d_e), Rel is (num_relations, d_r).
 *   - triples are int32 (n, 3) rows of (subject, predicate, object) ids.  Out-of-range ids never fault:
 *     the row scores NaN, contributes no gradient, and bit 0 of *err_flag (device word) is set.
 */
#ifndef KGE_HIP_H
#define KGE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KGE_ABI_VERSION 10

typedef void* kge_stream_t; /* hipStream_t */

typedef enum {
  KGE_OK = 0,
  KGE_EINVAL = -1,      /* bad argument (null pointer, negative size, unknown enum) */
  KGE_ESHAPE = -2,      /* widths that the reference would reject with a broadcast error */
  KGE_EHIP = -3,        /* a HIP runtime call failed */
  KGE_EWORKSPACE = -4,  /* workspace too small */
  KGE_EUNSUPPORTED = -5 /* valid in the reference, not built here yet */
} kge_status;

/* KGEkeras.models.* classes on the path (models.py:164-237). */
typedef enum { KGE_DISTMULT = 0, KGE_TRANSE = 1, KGE_COMPLEX = 2, KGE_HOLE = 3, KGE_CONVE = 4,
               KGE_ROTATE = 5 /* models.py:421-459 exactly as written: entity rows [re | im] of 2d floats, relation rows d phases;
                               * score = gamma - S if gamma > 0 else S, S = sum_k (re_s cos p - im_s sin p - re_o) + (re_s sin p + im_s cos p - im_o)
                               * (the signed sum: `norm` is never applied there).  d_e == 2 * d_r.  Training runs on the push path. */
               ,
               KGE_HAKE = 6   /* models.py:362-400 exactly as written: entity rows [phase | modulus] (d_e = 2h), relation rows
                               * [phase | modulus | bias] (d_r = 3h); the modulus part of the score is ONE norm over all rows of a call
                               * (tf.norm without an axis), so it has entry points of its own that take the rows of a call as a
                               * segment: kge_hake_score, kge_hake_train_grads (+ kge_opt_apply).  The generic entry points refuse it. */
} kge_model;

/* KGEkeras.utils.loss_function_lookup names (utils.py:321-330). */
typedef enum {
  KGE_POINTWIZE_HINGE = 0,    /* utils.py:288-289 */
  KGE_POINTWIZE_LOGISTIC = 1, /* utils.py:291-292 */
  KGE_PAIRWIZE_HINGE = 2,     /* utils.py:300-305 */
  KGE_PAIRWIZE_LOGISTIC = 3,  /* utils.py:307-312 */
  KGE_POINTWIZE_SQUARE = 4,   /* utils.py:294-295 */
  KGE_PAIRWIZE_SQUARE = 5,    /* utils.py:314-319 */
  KGE_POINTWIZE_CROSS_ENTROPY = 6 /* utils.py:297-298 (Keras binary_crossentropy on the raw scores) */
} kge_loss;

typedef enum { KGE_OPT_SGD = 0, KGE_OPT_ADAM_TF = 1 } kge_opt;

/* EmbeddingModel.__init__ arguments that shape the arithmetic (models.py:14-129, 176-190, 204-209). */
typedef struct {
  int32_t model;         /* kge_model */
  int32_t num_entities;  /* rows of E */
  int32_t num_relations; /* rows of Rel */
  int32_t d_e, d_r;      /* row widths in floats (ComplEx: 2*e_dim, models.py:208-209) */
  int32_t norm;          /* TransE: 1 or 2 (models.py:178,188); 3 = the phase distance of pRotatE (models.py:479-489):
                          * dist = modulus * sum_k |sin((s_k + p_k - o_k) / 2)|, same gamma convention (models.py:486-489);
                          * 5 = tf.norm(x, ord=p) for any other positive real p (models.py:188 `ord=self.norm`): dist = (sum_k |x_k|^p)^(1/p)
                          * with p in `modulus`; scoring, push-path training and the VALU distance sweep of the ranking */
  float gamma;           /* TransE (models.py:179,187-190): score = gamma - dist if gamma > 0 else +dist */
  float dropout;         /* Dropout(dp) on the three gathered rows, training only (models.py:122,144-148) */
  int32_t conv_filters;  /* ConvE: Conv2D filters (models.py:242) */
  float hidden_dropout;  /* ConvE: Dropout(hidden_dp) after each batch norm (models.py:241,257,262) */
  float modulus;         /* norm 3: models.py:476 modulus * embedding_range; norm 5: the exponent p */
  int32_t conv_kh, conv_kw; /* ConvE: Conv2D kernel (conv_size_w, conv_size_h) of models.py:243-244,254 = (rows, columns) over the
                             * (2w, h) image; 0 = the reference default 3.  8 filters of 3x3 take register-tiled kernels, every
                             * other shape the general ones (ABI version 6) */
} kge_model_desc;

/* loss_function(true, false, margin, negative_samples) as called at tests.py:80-81. */
typedef struct {
  int32_t loss;             /* kge_loss */
  int32_t negative_samples; /* reshape width of `false` for pairwize_* (utils.py:301) */
  float margin;
  float regularization;     /* l3_reg weight on the entity table (models.py:10-11,78-92); 0 = off */
} kge_loss_desc;

/* Keras optimizer state for both tables.  ADAM_TF: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); dense moment decay and
 * dense update over every row; eps outside the bias correction (tests.py:86 `optimizer='adam'`). */
#define KGE_STEP_STATE_BYTES 24
typedef struct {
  int32_t kind;  /* kge_opt */
  float lr, beta1, beta2, eps;
  int64_t step;  /* t of THIS update, 1-based (ignored when step_dev is set) */
  float *mE, *vE, *mRel, *vRel; /* Adam slots, same shapes as the tables (unused for SGD) */
  /* Optional device-resident step state, for steps replayed from a hipGraph where no host value can change between
   * launches.  KGE_STEP_STATE_BYTES = 24: {int64 t = updates done so far, float lr_t of update t+1, float unused, double running sum
   * of the step losses}.  kge_opt_state_init seeds it from `step`; every kge_opt_apply / kge_train_step* then reads lr_t from it
   * and, as its last kernel, advances t, prepares the next lr_t and adds *loss_out to the running sum.  NULL: `step` above is used. */
  int64_t* step_dev;
} kge_opt_desc;

const char* kge_last_error(void);
int kge_abi_version(void);
/* Fills props[0..3] = {compute units, wavefront size, LDS bytes per CU, HBM bytes}; needs a GPU. */
int kge_device_props(int64_t props[4]);

/* ---- scoring: EmbeddingModel.call with dropout inactive, i.e. model.predict (models.py:134-162;
 *      utils.py:211,231).  out[i] = func(E[s_i], Rel[p_i], E[o_i]).  */
int kge_score(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n,
              float* out, uint32_t* err_flag, kge_stream_t stream);

/* The same with the Dropout layer ACTIVE -- EmbeddingModel.call(inputs, training=True) (models.py:134-162: every lookup goes through
 * self.dp, which Keras switches on under training=True): row i takes the masks of batch row row_offset + i of step `seed`, the masks
 * the training kernels regenerate (a counter-based hash of (seed, batch row, role, element); kept entries scaled by 1 / (1 - rate),
 * rate = md->dropout; rate 0 = kge_score).  TensorFlow's own mask stream is not reproducible; the oracle regenerates these. */
int kge_score_training(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples, int64_t n, uint64_t seed,
                       int64_t row_offset, float* out, uint32_t* err_flag, kge_stream_t stream);

/* ---- forward-only loss of one [pos, neg] batch in INFERENCE mode: what Keras logs as val_loss when fit() is given validation_data
 *      (tests.py:130-134) / Model.evaluate over the wrapper's add_loss (tests.py:77-84) -- scores of both arrays on the given tables,
 *      loss_function(score, fscore, margin, negative_samples); no gradient lists, gradient rows or optimizer.  E may be the
 *      transformed table of the literal path (kge_lit_transform).  *loss_out = the loss; running_sum (optional device double) += it,
 *      so an epoch of validation batches is summed without a host read per batch.  kge_l3_value adds the l3_reg term
 *      weight * sum |E|^3 (models.py:10-11,78-92) to *loss_inout -- once per evaluation, the tables do not move.
 *      Workspace: kge_loss_workspace_bytes(m_pos, m_neg). */
int kge_loss_only(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const int32_t* pos, int64_t m_pos,
                  const int32_t* neg, int64_t m_neg, float* loss_out, double* running_sum, void* workspace, int64_t workspace_bytes,
                  uint32_t* err_flag, kge_stream_t stream);
int kge_l3_value(const float* E, int64_t n, float weight, float* loss_inout, void* workspace, int64_t workspace_bytes, kge_stream_t stream);

/* ---- one training batch: the default Keras train_step over the two-input functional wrapper
 *      (tests.py:63-89,130-134): score(pos), score(neg) on the same tables, loss, sparse gradients of both
 *      tables with duplicates summed, optimizer update.  pos and neg are (m_pos,3)/(m_neg,3) as produced by
 *      oversample_data (utils.py:123-147).  gradE/gradRel are dense fp32 accumulators of the tables' shapes
 *      that must be all-zero on entry and are all-zero again on return.  *loss_out receives the batch loss
 *      (what Keras logs).  scores_out (optional, may be NULL) receives the m_pos+m_neg forward scores.
 *      seed feeds the dropout mask generator when md->dropout > 0.
 *      The workspace must be ZERO-FILLED ONCE when it is allocated (hipMemset); the library then leaves its counters
 *      clean at the end of every step (no per-step memset launches).  */
int64_t kge_train_step_workspace_bytes(const kge_model_desc* md, const kge_loss_desc* ld, int64_t m_pos, int64_t m_neg);
int kge_train_step(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel,
                   float* gradE, float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                   float* loss_out, float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes,
                   uint32_t* err_flag, kge_stream_t stream);

/* The two halves of kge_train_step, exposed for DP training where a gradient all-reduce sits between them:
 *   kge_train_grads : forward + loss + gradient accumulation into gradE/gradRel (+= ; caller zeroes once)
 *   kge_opt_apply   : optimizer update of both tables from gradE/gradRel * grad_scale, then re-zeroes them.
 * loss_out of kge_train_grads excludes the L3 term, which kge_opt_apply adds (pass the same pointer). */
int kge_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE,
                    float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                    float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                    kge_stream_t stream);
int kge_opt_apply(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel,
                  float* gradE, float* gradRel, float grad_scale, float* loss_out, void* workspace,
                  int64_t workspace_bytes, kge_stream_t stream);

/* ---- filtered 1-vs-all ranking: `validate` (utils.py:199-264) = gen_tail_data/gen_head_data (:165-196) +
 *      model.predict (:211,231) + rankdata(res,'max') (:212,232).  For query i (row i of `triples`; side 0 ranks
 *      the tail o against (s,p,?), side 1 ranks the head s against (?,p,o)):
 *          counts[i] (+)= #{c in [cand_offset, cand_offset+cand_rows) : c not filtered, score(i,c) > score(i,target_i)}
 *      so that real_rank = 1 + counts[i] summed over the shards that cover all entities.  E is the table the ids in
 *      `triples` index (md->num_entities rows; query vectors and target scores are built from it): the full table, or --
 *      with the table row-sharded across GPUs and never gathered (SURVEY 8(e) schedule ii) -- just the rows the queries
 *      reference, `triples` then holding positions in it.  E_cand points at the candidate row of GLOBAL id cand_offset
 *      (E_cand = E + cand_offset*d_e for a local full table; a row shard otherwise; cand_rows may be 0).  The filter is a COO list
 *      sorted or not: filt_query[e] = query index, filt_ids[e] = global entity id to exclude (the reference's
 *      filter_t[(s,p)] / filter_h[(p,o)]); it must not contain the target (the reference raises ValueError).
 *      accumulate != 0 adds into counts (multi-shard loops); otherwise counts is overwritten.
 *      side 2 ranks BOTH sides of the n_queries test triples in one sweep of the candidate table (what `validate` does per test
 *      triple, utils.py:210-236): query i in [0, n) is the tail query of triple i, query n + i its head query; counts,
 *      target_scores_out and the query numbers in filt_query follow that numbering (2 n entries), and the workspace is
 *      kge_rank_workspace_bytes(md, 2 n).  One launch with fuller query tiles instead of two; same counts as side 0 + side 1.
 *      DistMult, ComplEx, HolE and TransE-L2 (ordered by |e|^2 - 2<q,e>; target_scores_out must be NULL for that form, otherwise
 *      the VALU distance sweep runs) use the fp32 matrix cores; score comparisons are made on bit-identical fmaf chains. */
int64_t kge_rank_workspace_bytes(const kge_model_desc* md, int64_t n_queries);
int kge_rank_counts(const kge_model_desc* md, const float* E, const float* Rel, const float* E_cand, int32_t cand_offset,
                    int32_t cand_rows, const int32_t* triples, int64_t n_queries, int32_t side, const int32_t* filt_query,
                    const int32_t* filt_ids, int64_t n_filt, int32_t* counts, int32_t accumulate, float* target_scores_out,
                    void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* Test hook: out[i*num_entities + c] = the score of (query i, candidate c) exactly as the ranking kernels compare it. */
int kge_rank_chain_scores(const kge_model_desc* md, const float* E, const float* Rel, const int32_t* triples,
                          int64_t n_queries, int32_t side, float* out, void* workspace, int64_t workspace_bytes,
                          uint32_t* err_flag, kge_stream_t stream);

/* ---- ConvE (models.py:239-275): the conv / batch-norm / dense parameters live in one flat fp32 vector
 *      theta = [K(3,3,C) | bc(C) | g1(C) | b1(C) | Wd(F,d) | bd(d) | g2(d) | b2(d)]  (F = (2w-2)(h-2)C, (w,h) = largest factor pair of d)
 *      and the non-trainable moving statistics in stats = [mean1(C) | var1(C) | mean2(d) | var2(d)].
 *      kge_conve_score / kge_conve_hidden run in inference mode (moving statistics, no dropout): scores, or the hidden
 *      vectors x(s,p) (n,d) with score = <x, E[o]> (what the 1-vs-all sweep consumes).  kge_conve_train_grads is the
 *      gradient half of a Keras train step (batch statistics per call of the wrapper: pos and neg separately; moving
 *      statistics updated; gradients accumulated into gradE / gradRel / gtheta, all zero on entry); the optimizer half is
 *      kge_opt_apply for the tables plus kge_opt_apply_vector for theta.  The dense layer is rocBLAS sgemm. */
int kge_conve_sizes(const kge_model_desc* md, int64_t* n_theta, int64_t* n_stats);
int64_t kge_conve_workspace_bytes(const kge_model_desc* md, int64_t m_pos, int64_t m_neg);
int64_t kge_conve_infer_workspace_bytes(const kge_model_desc* md, int64_t n);
int kge_conve_score(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* triples,
                    int64_t n, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
int kge_conve_hidden(const kge_model_desc* md, const float* E, const float* Rel, const float* theta, float* stats, const int32_t* triples,
                     int64_t n, float* hidden_out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
int kge_conve_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, const float* theta,
                          float* stats, float* gradE, float* gradRel, float* gtheta, const int32_t* pos, int64_t m_pos,
                          const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed, void* workspace,
                          int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* ---- HolE without dropout in the frequency domain (models.py:231-237: real(ifft(conj(fft(s)) * fft(o))) . p).  By Parseval the
 *      score is the ComplEx score (models.py:218-222) of the half spectra of the three rows, so a step transforms the two tables
 *      (kge_hole_spec_forward: out (rows, w) = X (rows, d) * W (d, w), w = kge_hole_spec_width(d) = 2 * roundup4(d/2 + 1), layout
 *      [real | imag]), runs kge_train_grads with model = KGE_COMPLEX, d_e = d_r = w on the spectra, and carries the gradients
 *      back (kge_hole_spec_adjoint: G (rows, d) += Ghat (rows, w) * W^T; zero_hat != 0 re-zeroes Ghat).  kge_hole_spec_basis
 *      fills the two bases: WE for the entity table, WR (columns weighted by the Hermitian multiplicity / d) for the relation
 *      table.  Dropout steps keep the direct correlation kernels (every triple masks its own rows). */
int32_t kge_hole_spec_width(int32_t d);
int kge_hole_spec_basis(int32_t d, float* WE, float* WR, kge_stream_t stream);
int kge_hole_spec_forward(const float* X, int64_t rows, int32_t d, const float* W, float* out, kge_stream_t stream);
int kge_hole_spec_adjoint(float* Ghat, int64_t rows, int32_t d, const float* W, float* G, int32_t zero_hat, kge_stream_t stream);
/* ---- LiteralE-style input path (models.py:114-120,152-158; tests.py:113,125-126): s_hat = act(Dense([dp(E[s]) ; dp(Lit[s])])),
 *      o_hat likewise with the same Dense, score = func(s_hat, dp(Rel[p]), o_hat).  Lit is the frozen (lit_rows, L) literal table,
 *      theta_l = [W (d_e + L, d_e) row-major | b (d_e)], act = 0 linear, 1 relu, 2 sigmoid, 3 tanh, 4 softplus, 5 softsign, 6 elu,
 *      7 selu, 8 exponential, 9 softmax (over the layer's d_e outputs).  kge_lit_transform builds the transformed entity table (rows, d_e) in inference mode: predict and
 *      the 1-vs-all ranking then run the ordinary kernels on it.  kge_lit_train_grads is the gradient half of a train step
 *      (gradients into gradE / gradRel / gtheta_l, zero on entry); built under DistMult, TransE (every norm id, i.e. pRotatE too), ComplEx,
 *      HolE and RotatE (d_e = 2 d_r: the Dense is d_e wide; the workspace must be zero-filled once, a padded relation-gradient table
 *      lives in it); ConvE and HAKE through kge_lit_conve_train_grads / kge_lit_hake_train_grads below. */
int64_t kge_lit_workspace_bytes(const kge_model_desc* md, int32_t L, int64_t m_pos, int64_t m_neg);
int64_t kge_lit_transform_workspace_bytes(const kge_model_desc* md, int32_t L, int32_t rows);
int kge_lit_transform(const kge_model_desc* md, int32_t act, const float* E, const float* Lit, int32_t L, int32_t lit_rows, const float* theta_l,
                      int32_t rows, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
int kge_lit_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, int32_t act, const float* E, const float* Rel, const float* Lit,
                        int32_t L, int32_t lit_rows, const float* theta_l, float* gradE, float* gradRel, float* gtheta_l, const int32_t* pos,
                        int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out, float* scores_out, uint64_t seed, void* workspace,
                        int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* The same gradient half with ConvE behind the transform (models.py:152-158 in front of ConvE.func, :264-275): the head runs on STAGED
 * tables -- entity rows = the batch's 2n transformed rows, relation rows = the batch's n relation rows with their dropout masks applied,
 * triples (t, t, n + t), its own input dropout off (the transform has masked the entity inputs) -- so its per-triple gradient rows are
 * dL/d(transformed rows), which then go back through the Dense.  theta_c / bn_stats / gtheta_c as in kge_conve_train_grads; the
 * workspace (kge_lit_workspace_bytes with the ConvE descriptor) must be zero-filled once. */
int kge_lit_conve_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, int32_t act, const float* E, const float* Rel, const float* Lit,
                              int32_t L, int32_t lit_rows, const float* theta_l, const float* theta_c, float* bn_stats, float* gradE, float* gradRel,
                              float* gtheta_l, float* gtheta_c, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                              float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* ---- HAKE (models.py:362-400; kge_model_desc.model = KGE_HAKE, gamma in the descriptor).  score_t = gamma - (phase_weight * sum_k
 *      |sin((phase_s + phase_p - phase_o) / 2)| + mod_weight * sqrt(sum over ALL rows of the call of M^2)) with phase = row / (embedding_range
 *      / pi), M = mod_s (mod_p + bias) - |mod_o| (1 - bias), bias = clip(bias_p, <= 1) raised to -|mod_p| -- the second term is tf.norm
 *      WITHOUT an axis: one scalar over the whole tensor handed to the call.
 *      kge_hake_score: rows [i * chunk, (i + 1) * chunk) are scored as one call each (chunk = 32: model.predict with Keras' default
 *      batch size, utils.py:211,231; chunk <= 0 or >= n: all n rows in one call = model(inputs)); or, seg_off != NULL, the n_seg calls
 *      [seg_off[i], seg_off[i+1]) of a device array of n_seg + 1 ascending offsets (the predict batches of many candidate lists in one
 *      launch; workspace kge_hake_workspace_bytes(n, -n_seg)).  kge_hake_rank_counts: counts[q] = #{rows of [q_off[q], q_off[q+1]) scoring
 *      above the first row of the range} -- the reference's rank of a candidate list whose row 0 is the target (utils.py:165-196,211-215)
 *      is 1 + that; n_q <= 2^31 - 1 queries, < 2^24 rows each.
 *      kge_hake_train_grads: the gradient half of the default Keras train step of the two-input wrapper (tests.py:63-89): the model is
 *      called once on pos and once on neg (one norm per side), loss by ld, gradients of the three gathered rows added into gradE / gradRel
 *      (float atomics; zero on entry like everywhere); kge_opt_apply completes the step.  Dropout by md->dropout / seed as in
 *      kge_train_grads.  At sqrt(sum M^2) = 0 TensorFlow's gradient is NaN; zero here.  Workspace: kge_hake_workspace_bytes(n, chunk)
 *      (chunk = 0 for training, n = m_pos + m_neg). */
typedef struct { float phase_weight, mod_weight, embedding_range; } kge_hake_params;
int64_t kge_hake_workspace_bytes(int64_t n, int64_t chunk);
int kge_hake_score(const kge_model_desc* md, const kge_hake_params* hp, const float* E, const float* Rel, const int32_t* triples, int64_t n,
                   int64_t chunk, const int64_t* seg_off, int64_t n_seg, float* out, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                   kge_stream_t stream);
int kge_hake_rank_counts(const float* scores, int64_t n, const int64_t* q_off, int64_t n_q, int32_t* counts, kge_stream_t stream);
int kge_hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, const float* E, const float* Rel,
                         float* gradE, float* gradRel, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                         float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* ... and with HAKE behind the transform (LiteralE-style input path above): the same staging -- one call of the model per side on the
 * staged rows, so the call-wide norm is the side's as in kge_hake_train_grads; its dense accumulators over the staged tables are the
 * per-row gradients; the relation rows (3h wide, entity rows 2h) are scattered into gradRel by float atomics. */
int kge_lit_hake_train_grads(const kge_model_desc* md, const kge_loss_desc* ld, const kge_hake_params* hp, int32_t act, const float* E,
                             const float* Rel, const float* Lit, int32_t L, int32_t lit_rows, const float* theta_l, float* gradE, float* gradRel,
                             float* gtheta_l, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg, float* loss_out,
                             float* scores_out, uint64_t seed, void* workspace, int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* Seeds the device-resident step state: od->step_dev := {od->step - 1, lr_t(od->step), 0} (see kge_opt_desc). */
int kge_opt_state_init(const kge_opt_desc* od, kge_stream_t stream);
/* Keras optimizer update of one flat fp32 vector (w -= ..., g re-zeroed); od->mE / od->vE are its Adam slots. */
int kge_opt_apply_vector(const kge_opt_desc* od, float* w, float* g, int64_t n, float grad_scale, kge_stream_t stream);
/* Filtered rank counts for any model whose score is a dot product <q_i, c_j>: Q is (n_queries, K), Cand holds rows
 * cand_offset .. cand_offset+cand_rows of the K-wide candidate matrix, target_ids[i] is the global candidate id of query i's
 * target (must lie in the range), filters as in kge_rank_counts.  Same MFMA sweep, same bit-exact comparisons. */
int64_t kge_rank_sweep_q_workspace_bytes(int64_t n_queries, int32_t K);
int kge_rank_sweep_q(const float* Q, int64_t n_queries, int32_t K, const float* Cand, int32_t cand_offset, int32_t cand_rows,
                     const int32_t* target_ids, const int32_t* filt_query, const int32_t* filt_ids, int64_t n_filt, int32_t* counts,
                     int32_t accumulate, void* workspace, int64_t workspace_bytes, kge_stream_t stream);

/* ---- batch plans: the part of a training step that depends on the batch ids only, built ahead of the step.
 *      The reference's Keras `fit` prefetches batches from its Sequence (tests.py:29-61,130-134) while the previous step
 *      runs; here the prefetch stage also folds duplicate rows and builds the per-relation / per-entity gradient lists, so
 *      that the step proper starts at the forward gather.  A plan is an opaque device buffer:
 *        kge_batch_plan_bytes   size for (md, ld, m_pos, m_neg); KGE_EUNSUPPORTED where no plan exists (pairwize losses, HolE,
 *                               ConvE, gradient mode 1) -- callers then use kge_train_step;
 *        kge_batch_plan_build   asynchronous on `stream` (any stream; the arrays pos / neg are not referenced once part 1 is
 *                               done).  parts: 1 = rows (duplicates folded, counters cleared) + the histograms of both index
 *                               halves, 2 = entity list offsets, 4 = relation list offsets and lists; 7 = everything.  2 and 4
 *                               need 1 and are independent of each other, so a caller may run them on two streams (fork after
 *                               1, join before the step); together they are one launch.  Adding 8 says that the plan buffer's
 *                               counters are known to be zero -- the buffer was zero-filled when it was allocated and every
 *                               build since has been consumed by a step (which leaves them zero) -- and saves the clearing launch;
 *        kge_train_step_planned = kge_train_step on the batch the plan was built from (same md, ld, m_pos, m_neg), without
 *                               scores_out.  CONSUMES the plan: one step per build.  The caller orders build -> step
 *                               (events or same stream).  Workspace as for kge_train_step. */
int64_t kge_batch_plan_bytes(const kge_model_desc* md, const kge_loss_desc* ld, int64_t m_pos, int64_t m_neg);
int kge_batch_plan_build(const kge_model_desc* md, const kge_loss_desc* ld, const int32_t* pos, int64_t m_pos, const int32_t* neg, int64_t m_neg,
                         void* plan, int64_t plan_bytes, int32_t parts, uint32_t* err_flag, kge_stream_t stream);
/* The same plan from a batch given in folded form: cand_rows ((n_pos_cand + n_neg_cand), 3), positives first, with their
 * multiplicities cand_counts laid out as (pos_lanes * n_pos_cand + n_neg_cand) int32 (count of positive a = sum over lanes l of
 * cand_counts[l * n_pos_cand + a]; pos_lanes = 1 for a plain count vector, KGE_CAND_POS_LANES for the sampler's output).  Rows
 * with count 0 are skipped; the counts of the positives sum to m_pos, those of the negatives to m_neg.  No hash table is
 * needed (every candidate becomes a row of the plan, multiplicity 0 where nobody picked it: no compaction either).  Buffer size as
 * for (m_pos, m_neg); requires n_pos_cand + n_neg_cand <= m_pos + m_neg. */
int kge_batch_plan_build_counted(const kge_model_desc* md, const kge_loss_desc* ld, const int32_t* cand_rows, const int32_t* cand_counts,
                                 int32_t pos_lanes, int64_t n_pos_cand, int64_t n_neg_cand, int64_t m_pos, int64_t m_neg, void* plan,
                                 int64_t plan_bytes, int32_t parts, uint32_t* err_flag, kge_stream_t stream);
int kge_train_grads_planned(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE, float* gradRel,
                            void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace,
                            int64_t workspace_bytes, kge_stream_t stream);
int kge_train_step_planned(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, float* E, float* Rel, float* gradE,
                           float* gradRel, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed,
                           void* workspace, int64_t workspace_bytes, kge_stream_t stream);
/* kge_train_grads_planned in two halves (data-parallel training, SURVEY 8(e): the gradient tables are exchanged before the optimizer
 * runs, and the entity table's 11.6 MB are the bulk of that exchange):
 *   parts 1  the entity gradient rows -> gradE.  One launch; it forms its own scores and needs nothing from the forward, so the
 *            caller can start the all-reduce of gradE behind it ...
 *   parts 2  ... while the relation-grouped forward (loss, relation gradient rows -> gradRel, *loss_out) still runs;
 *   parts 3  both, on `stream` (= kge_train_grads_planned).
 * Either order; one plan serves both halves of ONE step.  kge_train_grads_halves_supported: 1 where the halves come apart (rows
 * of at most 256 floats; elsewhere only parts = 3 is accepted), 0 otherwise, negative for an invalid descriptor. */
int kge_train_grads_halves_supported(const kge_model_desc* md, const kge_loss_desc* ld);
int kge_train_grads_planned_parts(const kge_model_desc* md, const kge_loss_desc* ld, const float* E, const float* Rel, float* gradE, float* gradRel,
                                  void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out, uint64_t seed, void* workspace,
                                  int64_t workspace_bytes, int32_t parts, kge_stream_t stream);
/* kge_train_step_planned between twin tables: the step reads the parameters from (E_in, Rel_in) and writes the updated tables to
 * (E_out, Rel_out) -- distinct buffers of the same shapes; the input tables are left as they were; the moments of `od` advance in
 * place.  Because no reader of this step can meet an updated row, the wave that has summed the gradient of a row applies the
 * optimizer to it on the spot: no gradient tables, no optimizer sweep (tests.py:63-89 + optimizer 'adam' / 'sgd', Keras semantics:
 * every row of both tables is updated).  The step has two independent halves (parts mask):
 *   1  entity half   one launch: per entity row gather, own scores, dL/dscore, gradient row, optimizer -> E_out; the plan's entity
 *                    counters are left zero;
 *   2  relation half relation-grouped forward (loss, partial relation rows), then relation rows + optimizer -> Rel_out, *loss_out,
 *                    the plan's relation counters left zero, the step state od->step_dev moved on (t += 1, loss sum);
 *   3  both, on `stream`.
 * A caller may issue the halves of one step on two streams (both read (E_in, Rel_in) and the plan; they share nothing they write);
 * the next step needs both.  parts = 3: lr_t of the update is read from the step state like everywhere else (od->step ignored; the step
 * may be captured into a hipGraph); parts = 1 | 2: od->step = the number of this update (1-based), lr_t then travels by value, because
 * the relation half moves the state on while the entity half may still be running.  The caller alternates the two
 * table pairs from step to step.  Needs od->step_dev (kge_opt_state_init), no L3 term, rows of at most 256 floats (per
 * half row for ComplEx); kge_train_step_swap_supported says so ahead of time (1 / 0; a negative status for an invalid descriptor),
 * the step itself returns KGE_EUNSUPPORTED otherwise.  Same results as kge_train_step_planned up to fp32 summation order. */
int kge_train_step_swap_supported(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od);
int kge_train_step_swap(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                        float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                        uint64_t seed, void* workspace, int64_t workspace_bytes, int32_t parts, kge_stream_t stream);

/* The same step as one rank's share of an OWNER-COMPUTES step over row-sharded tables ("row-sharded tables ... each GPU runs
 * Adam on its rows + all-gather of updated rows"; north star: "the entity table row-shards across the 8 GPUs").  Every rank holds the
 * SAME batch -- the samplers are bit-exact functions of their seeds, so every rank draws it itself, no communication -- and the same
 * input tables; rank r calls this with its window and then the ranks all-gather the window rows of E_out / Rel_out and sum their
 * *loss_out.  The N-rank tables then equal the one-GPU tables of kge_train_step_swap on that batch (tests.py:55-61: ONE batch per
 * step -- a data-parallel replica step trains on `ranks` batches instead), the two-launch step is kept, the exchange is half of a
 * gradient all-reduce's bytes, and a rank streams 1/ranks of the optimizer's moments.
 *   entity rows  [ent_lo, ent_hi): gradient from the row's list, optimizer -> E_out rows, moment rows of the window.  Moments are
 *                addressed by GLOBAL row id: a rank that stores only its window's moment rows passes od->mE / od->vE offset by
 *                -ent_lo * d_e floats (likewise mRel / vRel by -rel_lo * d_r);
 *   relation rows [rel_lo, rel_hi): the relation-grouped forward runs on the units of these relations only -> partial rows ->
 *                optimizer -> Rel_out rows; *loss_out = the loss terms of the batch rows whose relation is in the window (their sum over
 *                disjoint windows covering [0, R) is the batch loss, up to fp32 summation order); the step state advances by one update
 *                and adds this partial loss to its sum.
 * Rows of E_out / Rel_out outside the window are NOT written.  The plan is consumed (all its counters left zero, also those of rows
 * outside the window): every rank builds its own plan of the batch.  No dropout (masks are per batch row: fine, but not built). */
typedef struct { int32_t ent_lo, ent_hi, rel_lo, rel_hi; } kge_row_window;
int kge_train_step_swap_window(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                               float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                               uint64_t seed, void* workspace, int64_t workspace_bytes, const kge_row_window* win, kge_stream_t stream);

/* CHAINED steps between twin tables: kge_train_step_swap (parts = 3) as ONE launch, for runs of steps.  The closing launch of a step
 * (relation rows = sums of the forward's partial rows + optimizer, the loss, the step state: ~1 us of work) costs ~10 us plus a ~7 us
 * kernel boundary on the step stream; here R "closer" workgroups and one "state" workgroup sit late in the step's own grid, wait on
 * per-workgroup flags of the relation-grouped forward (write-through stores, cache-bypassing loads, no atomics) and do that work while
 * the rest of the entity role is still running.  Same tables, moments, loss and step state as the same steps through
 * kge_train_step_swap, bit for bit (the same sums in the same order), and complete after every call.
 *   seq = 0, 1, 2, ...: the number of the step in the chain; the caller alternates the table pairs exactly as with kge_train_step_swap.
 *     seq = 0 is preceded by a small launch that copies lr_t of the update from the step state into a ring inside `workspace` (a wave
 *     of the entity role may read lr_t after the state workgroup has moved the step state on) and clears the flags; a chain must not be
 *     interleaved with other steps on the same workspace and step state -- end it, then start a new one with seq = 0.
 *   err_flag (may be null): bit 32 is set if a flag wait gave up (a device fault -- never observed; the waits are bounded so that
 *     such a fault cannot hang the device).
 * No row window, no halves apart.  The workspace must have been zero-filled once (like the plan buffers). */
int kge_train_step_swap_chain(const kge_model_desc* md, const kge_loss_desc* ld, const kge_opt_desc* od, const float* E_in, const float* Rel_in,
                              float* E_out, float* Rel_out, void* plan, int64_t plan_bytes, int64_t m_pos, int64_t m_neg, float* loss_out,
                              uint64_t seed, void* workspace, int64_t workspace_bytes, int64_t seq, uint32_t* err_flag, kge_stream_t stream);

/* ---- the reference's loss functions as callables on two score vectors (utils.py:288-319, reduce_mean=True):
 *      *loss_out = loss(true, false, margin, negative_samples); d_true / d_false (optional, may be NULL) receive the
 *      derivatives.  Workspace: kge_train_step_workspace_bytes(NULL-free md of any shape, ld, m_true, m_false) suffices;
 *      kge_loss_workspace_bytes gives the exact need. */
int64_t kge_loss_workspace_bytes(int64_t m_true, int64_t m_false);
int kge_loss_value(const kge_loss_desc* ld, const float* true_scores, int64_t m_true, const float* false_scores, int64_t m_false,
                   float* loss_out, float* d_true, float* d_false, void* workspace, int64_t workspace_bytes, kge_stream_t stream);

/* reduce_mean=False of the pairwize losses (utils.py:303-305,310-312,317-319): the raw pair tensor of the NumPy/TF
 * broadcast of false.reshape(-1, ns) against true: out is (m_false/ns, max(ns, m_true)) row-major, out_elems its size.
 * KGE_ESHAPE where the reference would raise a broadcast error; KGE_EINVAL for pointwize losses (they always reduce). */
int kge_loss_pairs(const kge_loss_desc* ld, const float* true_scores, int64_t m_true, const float* false_scores, int64_t m_false,
                   float* out, int64_t out_elems, kge_stream_t stream);

/* Gradient path of kge_train_step / kge_train_grads: 0 (default) = "pull": per-destination-row gather-reduce, no float
 * atomics; 1 = "push": fused forward + no-return fp32 global atomics.  Same results to fp32 rounding. */
int kge_set_grad_mode(int mode);

/* Duplicate-row folding of the pull path (pointwise losses, no dropout): oversample_data (utils.py:123-147) draws rows
 * with replacement, identical rows have identical scores and gradient terms, so the step hashes the batch, runs on the
 * distinct rows and weights each by its multiplicity (same sums up to fp32 order).  0 = off, 1 (default) = batches of at
 * least 4096 rows, 2 = always. */
int kge_set_dedup_mode(int mode);

/* Run-to-run reproducibility of the planned training step ("only the former is run-to-run deterministic").  By default a row's
 * place in its relation list and in its entity lists is the arrival order of integer atomics, so two runs from the same seeds add a
 * row's gradient terms in different orders: equal to fp32 rounding, not bit for bit (like TensorFlow's unsorted segment sums).  With
 * the mode on -- also by KGE_DETERMINISTIC=1 in the environment -- the place is the rank BY BATCH ROW (two stable sorts per plan, on
 * the plan stage: off the step's critical path), duplicate folding by hash is off, and every sum of the planned pointwise step has ONE
 * order: two seeded epochs are bit-identical.  Set it BEFORE sizing plan buffers (kge_batch_plan_bytes grows by the sort scratch).
 * Not covered: the push path's float atomics, the sorted pairwize_hinge form (its running sum is a decoupled look-back scan). */
int kge_set_deterministic(int on);
int kge_get_deterministic(void);

/* The harness's all-pairs broadcast of the pairwize losses (utils.py:300-319 called with negative_samples = 1, tests.py:81): past
 * 2^27 pairs pairwize_hinge is evaluated from the two SORTED score vectors (counts + a running double sum, no pair formed), past
 * 2^22 pairs pairwize_square_loss from four sums; below, the all-pairs kernels run.  The forms agree to fp32 rounding, not bit for
 * bit (summation order changes with the algorithm, hence with the batch size).  -1 (default) = by pair count as above, 0 = always the
 * all-pairs kernels, 1 = always the sorted / four-sums forms; the environment variable KGE_PAIR_SORT sets the initial value. */
int kge_set_pair_sort_mode(int mode);

/* ---- in-library kernel timing (bench.py's roofline leg): when enabled, every launcher brackets its dominant
 *      kernel with hipEvents recorded on the launch stream.  kge_prof_read synchronises those events, returns the
 *      summed duration (ms) and launch count for one label and clears that label.  Off by default. */
typedef enum { KGE_PROF_TRAIN_GRADS = 0, KGE_PROF_OPT = 1, KGE_PROF_SCORE = 2, KGE_PROF_PAIRWISE = 3, KGE_PROF_RANK = 4,
               KGE_PROF_SAMPLER = 5, KGE_PROF_FWD = 6, KGE_PROF_INDEX = 7, KGE_PROF_DEDUP = 8,
               KGE_PROF_RANK_SWEEP = 9 /* the MFMA sweep kernel alone, inside KGE_PROF_RANK */, KGE_PROF_NLABELS = 10 } kge_prof_label;
int kge_prof_enable(int on);
int kge_prof_read(int label, double* total_ms, int64_t* launches);

/* ---- host-side bit-exact samplers (MT19937), replacing the NumPy / `random` calls of the batch builder:
 *      np.random.seed / randint (utils.py:118-119), np.random.shuffle (tests.py:50-53),
 *      random.seed / random.choices (utils.py:136).  State is an opaque 2504-byte blob owned by the caller. */
#define KGE_MT_STATE_BYTES 2504
int kge_mt_seed_numpy(void* state_host, uint32_t seed);                       /* np.random.seed(int) = init_genrand */
int kge_mt_seed_python(void* state_host, const uint32_t* key_host, int n);    /* random.seed(int) = init_by_array(|seed| words) */
int kge_mt_randint(void* state_host, int64_t low, int64_t high, int64_t* out_host, int64_t n); /* np.random.randint(low, high, n) */
int kge_mt_shuffle(void* state_host, int64_t* perm_host, int64_t n);          /* np.random.shuffle */
int kge_mt_choices(void* state_host, int64_t population, int64_t* idx_host, int64_t k); /* random.choices(range(pop), k=k) */
int kge_mt_raw(void* state_host, uint32_t* out_host, int64_t n);              /* n raw 32-bit outputs */
/* generate_negative + oversample_data fused on the host (tests.py:55-61): from the (B,3) batch, the NumPy state
 * and the Python state, writes pos_out (m,3) and neg_out (m,3) int32, m = B*ns. */
int kge_build_batch_host(void* np_state_host, void* py_state_host, const int32_t* batch_host, int64_t B, int64_t ns,
                         int64_t num_e, int32_t* pos_out_host, int32_t* neg_out_host);

/* ---- filter index of `validate` (utils.py:201-205 + the candidate removal of :165-196) on the host: from the test
 *      triples of one side and the (n_filt,3) filtering triples, the COO list (query index, excluded entity id) that
 *      kge_rank_counts consumes.  Two-pass: with out_q == NULL only *n_out is written.  KGE_EINVAL where the reference
 *      raises ValueError (evaluated triple inside its own filter; ids outside range(num_entities)). */
int kge_filter_coo_host(const int32_t* test_host, int64_t n_queries, int32_t side, const int32_t* filt_host, int64_t n_filt,
                        int64_t num_entities, int32_t* out_q_host, int32_t* out_id_host, int64_t* n_out);

/* ---- dataset front-end (host): `load_kg` (utils.py:106-112: every line stripped and split on whitespace, Python's str.split()
 *      separators) for all splits of a dataset at once, and the id mapping of the reference's script (tests.py:102-117) made
 *      deterministic: there the ids come from enumerating Python sets of strings (they change with PYTHONHASHSEED); here an id is
 *      the rank of the name among the sorted distinct names of all splits (entities: subjects and objects, relations: predicates;
 *      code-point order = byte order of the UTF-8 files).  KGE_EINVAL for a missing file and for a line that does not hold exactly
 *      three tokens (the script's `for a, b, c in train` raises ValueError there).
 *        kge_kg_open     reads the files, builds the maps, returns a handle;
 *        kge_kg_sizes    number of entities / relations, rows of every split;
 *        kge_kg_triples  the (rows, 3) int32 id triples of one split;
 *        kge_kg_names    the names in id order, concatenated, with offsets[n + 1] (buf_host NULL: offsets only -> offsets[n] bytes). */
int kge_kg_open(const char* const* paths_host, int32_t n_splits, void** handle);
int kge_kg_sizes(void* handle, int64_t* n_entities, int64_t* n_relations, int64_t* rows_per_split_host);
int kge_kg_triples(void* handle, int32_t split, int32_t* out_host);
int kge_kg_names(void* handle, int32_t which, char* buf_host, int64_t buf_bytes, int64_t* offsets_host);
int kge_kg_close(void* handle);

/* The same batch builder on the GPU: states are the same 2504-byte blobs resident in device memory, batch (B,3) int32 on
 * the device, pos_out / neg_out (B*ns,3) int32 on the device.  Bit-identical to kge_build_batch_host, including the
 * generator states left behind.  err_flag bit 1 is set if the rejection-sampling margin of the raw buffer ever ran
 * short (never observed; the states are then left at the end of the buffer).  The workspace must be zero-filled when it is
 * allocated and again whenever (B, ns, num_e) changes (its layout depends on them and it holds a batch epoch that tags the
 * in-flight tile counts); between calls of one shape the library keeps it consistent. */
/* plan (optional): MT19937's word stream is sequential, but the state J words ahead is a GF(2)-linear function of the
 * current one.  kge_sampler_plan_host computes, for this batch shape, the jump polynomials x^J mod phi of the starts of
 * 16 segments per stream (host, ~0.2-2 s once, cached); uploaded and passed as plan_dev, the generator then runs as 16
 * workgroups per stream after one short sequential pre-pass.  plan_dev == NULL: fully sequential generation.  Same bits. */
int64_t kge_sampler_workspace_bytes(int64_t B, int64_t ns, int64_t num_e);
int64_t kge_sampler_plan_bytes(void);
int kge_sampler_plan_host(int64_t B, int64_t ns, int64_t num_e, void* plan_host);
int kge_build_batch_device(void* np_state_dev, void* py_state_dev, const int32_t* batch_dev, int64_t B, int64_t ns,
                           int64_t num_e, int32_t* pos_out_dev, int32_t* neg_out_dev, const void* plan_dev, void* workspace,
                           int64_t workspace_bytes, uint32_t* err_flag, kge_stream_t stream);
/* Same, with the positives read in place: row i of the batch is kg[index[starts[*cursor] * B + i]] (kg: all training
 * triples (n,3) int32; index: the epoch permutation, int64; starts: batch numbers in drawing order, int64; cursor: one
 * int64 on the device) -- the slice DataGenerator.__getitem__ takes (tests.py:43-47,55-56).  The last kernel adds one to
 * *cursor, so consecutive calls (or replays of a hipGraph holding the call) walk through `starts`.
 * cand_rows / cand_counts (optional, both or neither): the batch BEFORE oversampling as ((B + B*ns), 3) int32 rows -- the B
 * positives, then the B*ns rows of generate_negative (utils.py:114-121) -- and how often random.choices picked each row
 * (utils.py:136): the oversampled batch in folded form, a by-product of drawing it (kge_batch_plan_build_counted).
 * cand_counts holds KGE_CAND_POS_LANES*B + B*ns int32: the count of positive a is the sum over lanes l of
 * cand_counts[l*B + a] (B bins take B*ns picks, so their counters are striped to spread the same-address atomics); the count
 * of generated negative j is cand_counts[KGE_CAND_POS_LANES*B + j].  With the folded form requested, pos_out / neg_out may
 * both be NULL (the oversampled arrays are then not materialised; the generator states advance all the same). */
#define KGE_CAND_POS_LANES 16
int kge_build_batch_device_indexed(void* np_state_dev, void* py_state_dev, const int32_t* kg_dev, const int64_t* index_dev,
                                   const int64_t* starts_dev, int64_t* cursor_dev, int64_t B, int64_t ns, int64_t num_e,
                                   int32_t* pos_out_dev, int32_t* neg_out_dev, int32_t* cand_rows_dev, int32_t* cand_counts_dev,
                                   const void* plan_dev, void* workspace, int64_t workspace_bytes, uint32_t* err_flag,
                                   kge_stream_t stream);
/* GF(2) jump-ahead primitives (host): out[312] uint64 = coefficients of x^J mod phi; kge_mt_jump_host replaces the 624
 * state words by raw words [J, J+624) of the stream (bits 0..30 of word 0 are not state and not guaranteed). */
int kge_mt_jump_poly(int64_t J, uint64_t* out_host);
int kge_mt_jump_host(void* state_host, int64_t J);

/* ---- the device batch builder in streamed form: word generation AHEAD of the batches that consume it.  An MT19937 stream is one
 *      fixed sequence of 624-word blocks; which of its words a batch consumes is data dependent (NumPy's masked rejection,
 *      utils.py:118-119), the words themselves are not.  Each generator gets a ring of raw state blocks in HBM:
 *        kge_sampler_stream_init     seeds both rings from the state blobs (block 0 = the state block, consumption from its position);
 *        kge_sampler_stream_produce  one production round per generator: R = kge_sampler_stream_round() batches' worth of words from
 *                                    16 jump-started segments (plan_dev: the polynomials of kge_sampler_stream_plan_host, uploaded;
 *                                    kge_sampler_plan_bytes() bytes; NULL: one sequential pass) -- skipped on the device while
 *                                    enough words lie ahead of the consumer.  Protocol: offer one round after init and one more in
 *                                    front of the draw of every batch k = 0, R, 2R, ... (more often is harmless), each ordered
 *                                    behind the draw of batch k-2 at the latest; the draw of batch k must be ordered behind the
 *                                    rounds offered before the one in front of it -- so a round can run on a stream of its own
 *                                    and has R batches' time to finish;
 *        kge_sampler_stream_draw     the consumer half of kge_build_batch_device[_indexed] (same arguments, same outputs, same
 *                                    generator states left in the blobs): select -> choices / assembly -> advance.
 *      Bit-identical to kge_build_batch_device.  The buffer also holds the scratch of the draw (no separate workspace). */
int64_t kge_sampler_stream_bytes(int64_t B, int64_t ns, int64_t num_e);
int32_t kge_sampler_stream_round(void);
int kge_sampler_stream_plan_host(int64_t B, int64_t ns, int64_t num_e, void* plan_host);
int kge_sampler_stream_init(void* stream_buf, int64_t stream_bytes, const void* np_state_dev, const void* py_state_dev, int64_t B, int64_t ns,
                            int64_t num_e, kge_stream_t stream);
int kge_sampler_stream_produce(void* stream_buf, int64_t stream_bytes, int64_t B, int64_t ns, int64_t num_e, const void* plan_dev,
                               kge_stream_t stream);
int kge_sampler_stream_draw(void* stream_buf, int64_t stream_bytes, void* np_state_dev, void* py_state_dev, const int32_t* batch_dev,
                            const int32_t* kg_dev, const int64_t* index_dev, const int64_t* starts_dev, int64_t* cursor_dev, int64_t B,
                            int64_t ns, int64_t num_e, int32_t* pos_out_dev, int32_t* neg_out_dev, int32_t* cand_rows_dev,
                            int32_t* cand_counts_dev, uint32_t* err_flag, kge_stream_t stream);

/* ---- RCCL collectives of the multi-GPU halves of the path (the reference is single-process and has no counterpart), on
 *      a communicator the CALLER made (ncclComm_t passed as void*), enqueued on `stream`.  RCCL is resolved at run time (from the
 *      process, else librccl.so.1); KGE_EUNSUPPORTED when it is not there.
 *        kge_allgather_rows    schedule (i) of the sharded evaluation: every rank contributes its (rows_per_rank, d) shard (padded to
 *                              equal size), `full` receives the (n_ranks * rows_per_rank, d) table;
 *        kge_allreduce_counts  schedule (ii): sum over ranks of the per-query int32 counts of kge_rank_counts over resident shards;
 *        kge_allreduce_f32     data-parallel training: sum over ranks of the flat gradient buffer (both tables + the loss word)
 *                              between kge_train_grads[_planned] and kge_opt_apply(grad_scale = 1 / ranks). */
int kge_allgather_rows(void* comm, const float* shard, int64_t rows_per_rank, int32_t d, float* full, kge_stream_t stream);
int kge_allreduce_counts(void* comm, int32_t* counts, int64_t n, kge_stream_t stream);
int kge_allreduce_f32(void* comm, float* buf, int64_t n, kge_stream_t stream);

/* ---- training pipeline: what Keras' fit() does for the reference with a prefetching Sequence (tests.py:29-61,130-134) as a native
 *      host loop over four device streams, one to two batches apart: word generation (kge_sampler_stream_produce), draw batch k
 *      (kge_sampler_stream_draw, folded output), plan batch k (kge_batch_plan_build_counted), train step k
 *      (kge_train_step_swap between the caller's tables and their twins where supported and E_twin / Rel_twin are given, else
 *      kge_train_step_planned; the device-resident optimizer step state either way).  Everything that changes from step to step lives on
 *      the device (generator rings and states, batch cursor, step state): the loop only launches.  By default it launches every
 *      stage's kernels itself (about 15 launches per step, ~0.05 ms of host time); with KGE_PIPE_GRAPHS=1 in the environment every
 *      stage is captured once per buffer parity into a hipGraph and replayed (slower on ROCm 7.2: back-to-back graph launches on
 *      one stream leave 20-30 us of idle time each).  When `stream` of kge_pipeline_create is not the default stream (0), the draw
 *      and plan streams are CU-masked streams over half of the compute units of every shader engine (blocking streams: they
 *      synchronise with the legacy default stream, which is why the caller must then stay off that one); KGE_PIPE_CUMASK=0 turns that
 *      off.  The descriptor's pointers are borrowed until kge_pipeline_destroy.
 *        kge_pipeline_start   (re)starts an epoch of n_batches batches: the caller has loaded index / starts / the generator states
 *                             and reset *cursor on `stream`; seeds the rings, issues rounds, draws and plans of batches 0 and 1
 *        kge_pipeline_run     n_steps training steps on `stream` (steps t .. t+n_steps-1 of the epoch), each followed by the round,
 *                             the draw and the plan of the batch two steps ahead; asynchronous (returns when everything is enqueued);
 *        kge_pipeline_finish  makes `stream` wait for the producer streams;
 *        kge_pipeline_state   {plan buffers in use, steps issued, table parity, batches}: what a caller needs to carry an epoch on
 *                             itself -- batch k sits planned in plan buffer k % out[0] (plan[0], plan[1], plan3), candidate buffer k & 1;
 *        kge_pipeline_set_seed  dropout only (md.dropout > 0 needs the pos / neg arrays of the descriptor): the dropout seed of the
 *                             step issued next is seed + 1, of the one after it seed + 2, ... (the masks of kge_train_step_planned).
 *      No hook between the halves of a step: data-parallel training drives the same library calls itself. */
typedef struct {
  kge_model_desc md;
  kge_loss_desc ld;
  kge_opt_desc od;            /* od.step_dev is required (kge_opt_state_init) */
  float *E, *Rel, *gradE, *gradRel;
  float* loss_out;
  void* train_ws;             /* kge_train_step_workspace_bytes(md, ld, B*ns, B*ns) */
  int64_t train_ws_bytes;
  uint32_t* err_flag;
  void *np_state, *py_state;  /* KGE_MT_STATE_BYTES each, device */
  const int32_t* kg;          /* (n,3) training triples */
  const int64_t* index;       /* epoch permutation */
  const int64_t* starts;      /* batch numbers in drawing order */
  int64_t* cursor;            /* one int64 */
  int64_t B, ns, num_e;
  const void* sampler_plan;   /* kge_sampler_plan_host output, uploaded (may be NULL: sequential generation) */
  void* sampler_ws;
  int64_t sampler_ws_bytes;
  uint32_t* sampler_err;
  void* sampler_stream;       /* kge_sampler_stream_bytes(B, ns, num_e), zero-filled when allocated: the generators' word rings, filled
                               * ahead of need on a stream of their own (NULL: every batch regenerates its words itself) */
  int64_t sampler_stream_bytes;
  const void* sampler_stream_plan; /* kge_sampler_stream_plan_host output, uploaded (may be NULL: sequential rounds) */
  int32_t* cand_rows[2];      /* ((B + B*ns), 3) per buffer parity */
  int32_t* cand_counts[2];    /* KGE_CAND_POS_LANES*B + B*ns per buffer parity */
  void* plan[2];              /* kge_batch_plan_bytes(md, ld, B*ns, B*ns) per buffer parity */
  int64_t plan_bytes;
  int32_t* pos[2];            /* optional, with neg (all four or none): (B*ns, 3) oversampled arrays per buffer parity.  With them the
                               * batches travel as plain arrays -- draw -> kge_batch_plan_build -> kge_train_step_planned with the
                               * dropout seed of the step -- which is what a model with dropout needs (every oversampled row has
                               * its own mask: no folded batches); cand_rows / cand_counts may then be NULL */
  int32_t* neg[2];
  float *E_twin, *Rel_twin;   /* optional (both or neither): second buffers of the two tables' shapes.  With them, and where
                               * kge_train_step_swap_supported, the steps alternate between (E, Rel) and the twins
                               * (kge_train_step_swap: two launches per step instead of three); every kge_pipeline_run leaves the
                               * current tables in (E, Rel) -- one device copy behind its last step when it issued an odd number,
                               * also when it fails half way (should even that copy fail, kge_pipeline_state reports parity 1) */
  void* plan3;                /* optional third plan buffer (plan_bytes, zero-filled like the others): the plan of batch k+2 then waits for
                               * step k-1 instead of step k -- no plan stage trails the last step of a run */
  void* plan_more[3];         /* optional further plan buffers behind plan3 (plan_bytes each, zero-filled, NULL-terminated).  With four or
                               * more buffers in all the loop records "plan consumed" on the step stream behind every (buffers - 2)-th
                               * step only (an experiment knob: at the bench shape three buffers measure fastest) */
  void* comm;                 /* optional: an RCCL communicator (ncclComm_t) the caller made over `world` ranks, one process per GPU.  With
                               * it every step is a data-parallel step: gradient half into gradE / gradRel, all-reduce(sum)
                               * of grad_flat over the ranks, kge_opt_apply(grad_scale = 1 / world) -- issued by the same native loop.
                               * Where the halves of the gradient computation come apart (kge_train_grads_halves_supported) the
                               * all-reduce of gradE runs on a stream of its own while the relation half still computes.  The twin
                               * tables are not used (the gradient tables must exist to be exchanged).  *loss_out and the step state's
                               * loss sum then hold the SUM of the ranks' losses */
  int32_t world;
  float* grad_flat;           /* with comm: the flat buffer one all-reduce covers -- gradE first ([gradE | gradRel | ... | loss word]) */
  int64_t grad_flat_n;        /* its length in floats */
} kge_pipeline_desc;
int kge_pipeline_create(const kge_pipeline_desc* desc, kge_stream_t stream, void** handle);
int kge_pipeline_start(void* handle, int64_t n_batches, kge_stream_t stream);
int kge_pipeline_set_seed(void* handle, uint64_t seed);
int kge_pipeline_run(void* handle, int64_t n_steps, kge_stream_t stream);
int kge_pipeline_finish(void* handle, kge_stream_t stream);
int kge_pipeline_state(void* handle, int64_t out[4]);
/* Host-side accounting of kge_pipeline_run since the last reset: out = {ms inside run, ms inside the stage launchers, stage launches;
 * graph mode only: ms waiting for a graph's previous launch to finish, launches that had to wait, that wait by stage} in the
 * order {run, wait, launch, launches, waited, wait draw, wait plan, wait step}. */
int kge_pipeline_stats(void* handle, double out[8], int32_t reset);
/* Stage timeline (a diagnostic): n > 0 arms hipEvent pairs around the next n stage launches; n == 0 synchronises and reads them back:
 * out receives up to cap records {stage (0 rounds, 1 draw, 2 plan, 3 step), batch / round number, begin ms, end ms}, relative to the
 * first begin; returns the number of records. */
int64_t kge_pipeline_trace(void* handle, int64_t n, double* out, int64_t cap);
int kge_pipeline_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* KGE_HIP_H */
