"""Counterpart of KGEkeras/models.py for the hot path: the same class names, constructor kwargs and
call / func / predict / compile / fit / train_step surface, with every gather, score, loss, gradient
and optimizer update running in the HIP kernels of libkge_hip.so.

Reference lines (relative to the reference repo):
  EmbeddingModel.__init__  KGEkeras/models.py:14-129      call  :134-162      get_config :131-132
  DistMult :164-172   TransE :175-190   ComplEx :203-222   HolE :224-237   ConvE :239-275
Keras-inherited behaviour the reference relies on (predict at utils.py:211,231; compile/fit at
tests.py:86-87,130-134; default train_step over add_loss) is restated on EmbeddingModel itself.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, ModelDesc, LossDesc, OptDesc, MODEL_IDS, LOSS_IDS, OPT_IDS, ACT_IDS


def _device():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def glorot_uniform(rows, cols, generator=None, device=None):
    """keras GlorotUniform for an Embedding matrix (models.py:87-88,101-102): U(-l, l), l = sqrt(6/(rows+cols))."""
    lim = float(np.sqrt(6.0 / (rows + cols)))
    t = torch.empty((rows, cols), dtype=torch.float32, device=device)
    t.uniform_(-lim, lim, generator=generator)
    return t


def as_triples(x, device, num_entities=None, num_relations=None):
    """(n,3) ids as a contiguous int32 device tensor.  Host arrays are range-checked here (TensorFlow's CPU gather
    raises on out-of-range ids); device tensors are checked by the kernels through the error flag."""
    if isinstance(x, torch.Tensor):
        t = x.reshape(-1, 3)
        if t.device.type != "cuda":
            return as_triples(t.numpy(), device, num_entities, num_relations)
        return t.to(dtype=torch.int32).contiguous()
    a = np.asarray(x)
    a = a.reshape(-1, 3)
    if a.dtype.kind == "f":  # Keras Input((3,)) is float32; Embedding casts to int32
        a = a.astype(np.int64)
    if a.size and num_entities is not None:
        if a[:, [0, 2]].min() < 0 or a[:, [0, 2]].max() >= num_entities or a[:, 1].min() < 0 or a[:, 1].max() >= num_relations:
            raise IndexError("triple ids out of range of the embedding tables (indices[...] is not in [0, N))")
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=False)


class EmbeddingTable:
    """What the reference keeps as a keras Embedding layer (models.py:87-112): the weight matrix, on the GPU."""

    def __init__(self, weights, name):
        self.weights = weights  # torch.float32 (rows, width), contiguous, device
        self.name = name
        self.trainable = True

    def get_weights(self):
        return [self.weights.detach().cpu().numpy()]

    def set_weights(self, ws):
        w = torch.as_tensor(np.asarray(ws[0], dtype=np.float32), device=self.weights.device)
        if w.shape != self.weights.shape:
            raise ValueError(f"Layer weight shape {tuple(self.weights.shape)} not compatible with provided weight shape {tuple(w.shape)}")
        self.weights.copy_(w)


class _Optimizer:
    """Keras optimizer as named in compile() (tests.py:86): 'adam' (lr 1e-3, b1 .9, b2 .999, eps 1e-7) or 'sgd' (lr .01)."""

    def __init__(self, kind, **kw):
        kind = kind.lower()
        if kind not in OPT_IDS:
            raise ValueError(f"Unknown optimizer: {kind}")
        self.kind = kind
        d = dict(lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7) if kind == "adam" else dict(lr=0.01, beta_1=0.0, beta_2=0.0, epsilon=0.0)
        if "learning_rate" in kw:
            kw["lr"] = kw.pop("learning_rate")
        d.update(kw)
        self.lr, self.beta_1, self.beta_2, self.epsilon = d["lr"], d["beta_1"], d["beta_2"], d["epsilon"]
        self.iterations = 0
        self.slots = None  # (mE, vE, mRel, vRel)


def loss_value(name, true, false, margin=1, negative_samples=1, return_grads=False):
    """loss_function_lookup(name)(true, false, margin, negative_samples) of the reference (utils.py:288-319) on the GPU:
    a 0-d device tensor (and, with return_grads, d loss/d true and d loss/d false)."""
    if name not in LOSS_IDS:
        raise KeyError(name)
    dev = _device()
    t = torch.as_tensor(true, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    f = torch.as_tensor(false, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    need = check(lib.kge_loss_workspace_bytes(t.numel(), f.numel()))
    ws = torch.zeros(int(need), dtype=torch.uint8, device=dev)
    out = torch.zeros(1, dtype=torch.float32, device=dev)
    gt = torch.empty_like(t) if return_grads else None
    gf = torch.empty_like(f) if return_grads else None
    ld = LossDesc(LOSS_IDS[name], int(negative_samples), float(margin), 0.0)
    check(lib.kge_loss_value(C.byref(ld), t.data_ptr(), t.numel(), f.data_ptr(), f.numel(), out.data_ptr(),
                             gt.data_ptr() if return_grads else None, gf.data_ptr() if return_grads else None,
                             ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    return (out[0], gt, gf) if return_grads else out[0]


def loss_pairs(name, true, false, margin=1, negative_samples=1):
    """reduce_mean=False of a pairwize loss (utils.py:303-305,310-312,317-319): the raw pair tensor
    phi(false.reshape(-1, negative_samples) - true) with the reference's broadcast, as a device tensor."""
    dev = _device()
    t = torch.as_tensor(true, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    f = torch.as_tensor(false, dtype=torch.float32, device=dev).reshape(-1).contiguous()
    ns = int(negative_samples)
    rows, cols = f.numel() // max(ns, 1), max(ns, t.numel())
    out = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    ld = LossDesc(LOSS_IDS[name], ns, float(margin), 0.0)
    check(lib.kge_loss_pairs(C.byref(ld), t.data_ptr(), t.numel(), f.data_ptr(), f.numel(), out.data_ptr(), out.numel(),
                             _lib.stream_ptr()))
    return out


class BatchPlan:
    """The half of a training step that depends on the batch ids only (duplicate rows folded, per-relation and per-entity
    gradient lists), built ahead of the step by EmbeddingModel.plan_batch -- typically on a side stream while the previous
    step runs -- and consumed by exactly one train_step / train_grads (kge_batch_plan_build / kge_train_step_planned)."""
    __slots__ = ("buf", "m_pos", "m_neg")

    def __init__(self, buf, m_pos, m_neg):
        self.buf, self.m_pos, self.m_neg = buf, int(m_pos), int(m_neg)


class EmbeddingModel:
    MODEL = None

    def __init__(self, e_dim, r_dim, num_entities, num_relations, negative_samples=2, batch_size=16,
                 loss_function='pointwize_hinge', dp=0.2, margin=1, loss_weight=1, regularization=0.0,
                 literal_activation=None, literals=None, use_batch_norm=False, entity_embedding_args=None,
                 relational_embedding_args=None, init_entities=None, init_relations=None, name='embedding_model',
                 **kwargs):
        dev = _device()
        self.device = dev
        self.name = name
        self.regularization = regularization
        self.num_entities = int(num_entities)
        self.num_relations = int(num_relations)
        seed = kwargs.pop("seed", None)
        gen = None
        if seed is not None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(seed))
        if init_entities is not None:  # models.py:94-99
            E = torch.as_tensor(np.asarray(init_entities, dtype=np.float32), device=dev).contiguous().clone()
            if tuple(E.shape) != (self.num_entities, e_dim):
                raise ValueError(f"init_entities has shape {tuple(E.shape)}, expected {(self.num_entities, e_dim)}")
        else:
            E = glorot_uniform(self.num_entities, e_dim, gen, dev)
        if init_relations is not None:  # models.py:107-112
            Rl = torch.as_tensor(np.asarray(init_relations, dtype=np.float32), device=dev).contiguous().clone()
            if tuple(Rl.shape) != (self.num_relations, r_dim):
                raise ValueError(f"init_relations has shape {tuple(Rl.shape)}, expected {(self.num_relations, r_dim)}")
        else:
            Rl = glorot_uniform(self.num_relations, r_dim, gen, dev)
        self.entity_embedding = EmbeddingTable(E, name + '_entity_embedding')
        self.relational_embedding = EmbeddingTable(Rl, name + '_relational_embedding')
        self.dp = float(dp)  # the reference keeps a Dropout layer here (models.py:122); the rate is what matters
        self.e_dim = e_dim
        self.r_dim = r_dim
        self.margin = margin
        self.loss_weight = loss_weight
        self.loss_function = loss_function.name if hasattr(loss_function, "name") else loss_function
        self.negative_samples = negative_samples
        self.batch_size = batch_size
        self.loss_negative_samples = 1  # what the harness passes to the loss (tests.py:81)
        self.optimizer = None
        self.stop_training = False
        self._gE = None
        self._gR = None
        self._ws = None
        self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._seed = int(seed) if seed is not None else 0
        self._lit = None
        if literal_activation:  # models.py:114-120: frozen literal table + Dense(e_dim, activation) on [embedding ; literal]
            if self.MODEL not in ("DistMult", "TransE", "ComplEx", "HolE", "RotatE", "ConvE", "HAKE"):  # (MODEL "TransE" covers TransE with any norm and pRotatE)
                raise NotImplementedError(f"the literal input path is not built for {type(self).__name__}")
            act = literal_activation if isinstance(literal_activation, str) else getattr(literal_activation, "__name__", None)
            if act not in ACT_IDS:
                raise NotImplementedError(f"literal_activation {literal_activation!r}: one of {sorted(k for k in ACT_IDS if k)}")
            lit = torch.as_tensor(np.asarray(literals, dtype=np.float32), device=dev).contiguous()
            if lit.ndim != 2:
                raise ValueError("literals must be a (len(literals), L) array")
            L = int(lit.shape[1])
            K1 = int(e_dim) + L
            theta = torch.zeros(K1 * int(e_dim) + int(e_dim), dtype=torch.float32, device=dev)
            lim = float(np.sqrt(6.0 / (K1 + int(e_dim))))  # Dense: Glorot-uniform kernel, zero bias
            theta[:K1 * int(e_dim)].uniform_(-lim, lim, generator=gen)
            self._lit = dict(table=lit, L=L, act=ACT_IDS[act], theta=theta, gtheta=torch.zeros_like(theta), slots=None, ws=None, tws=None)
        self.__dict__.update(kwargs)  # models.py:129: unknown kwargs become attributes

    # ------------------------------------------------------------------ descriptors
    def _model_desc(self, training=False):
        n = getattr(self, "norm", 1)  # (TransE maps its `norm` argument onto the library's norm ids itself)
        nid = int(n) if isinstance(n, (int, np.integer)) and not isinstance(n, bool) else 1
        return ModelDesc(MODEL_IDS[self.MODEL], self.num_entities, self.num_relations, int(self.e_dim), int(self.r_dim),
                         nid, float(getattr(self, "gamma", 0.0)), float(self.dp) if training else 0.0,
                         int(getattr(self, "conv_filters", 0)), float(getattr(self, "hidden_dp", 0.0)) if training else 0.0,
                         float(getattr(self, "modulus", 0.0)) if self.MODEL == "TransE" else 0.0,
                         int(getattr(self, "conv_size_w", 0)), int(getattr(self, "conv_size_h", 0)))

    def _loss_desc(self):
        if self.loss_function not in LOSS_IDS:
            raise KeyError(self.loss_function)  # utils.py:322-330
        return LossDesc(LOSS_IDS[self.loss_function], int(self.loss_negative_samples), float(self.margin or 1),
                        float(self.regularization))

    def get_config(self):
        return self.__dict__  # models.py:131-132

    @property
    def E(self):
        return self.entity_embedding.weights

    @property
    def Rel(self):
        return self.relational_embedding.weights

    # ------------------------------------------------------------------ literal input path (models.py:114-120,152-158)
    def get_literal_dense(self):
        """(W (e_dim+L, e_dim), b (e_dim)) of the literal Dense layer as numpy arrays."""
        K1, d = int(self.e_dim) + self._lit["L"], int(self.e_dim)
        th = self._lit["theta"].cpu().numpy()
        return th[:K1 * d].reshape(K1, d).copy(), th[K1 * d:].copy()

    def set_literal_dense(self, W, b):
        th = np.concatenate([np.asarray(W, dtype=np.float32).reshape(-1), np.asarray(b, dtype=np.float32).reshape(-1)])
        self._lit["theta"].copy_(torch.from_numpy(th))

    def _eval_table(self):
        """The entity table the scoring / ranking kernels see in inference mode: E itself, or, with the literal path,
        act(Dense([E ; literals])) of every entity (dropout is the identity outside fit)."""
        if self._lit is None:
            return self.E
        lt = self._lit
        md = self._model_desc(False)
        N = self.num_entities
        need = check(lib.kge_lit_transform_workspace_bytes(C.byref(md), lt["L"], N))
        if lt["tws"] is None or lt["tws"].numel() < need:
            lt["tws"] = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        out = torch.empty((N, int(self.e_dim)), dtype=torch.float32, device=self.device)
        check(lib.kge_lit_transform(C.byref(md), lt["act"], self.E.data_ptr(), lt["table"].data_ptr(), lt["L"], lt["table"].shape[0],
                                    lt["theta"].data_ptr(), N, out.data_ptr(), lt["tws"].data_ptr(), lt["tws"].numel(), self._err.data_ptr(),
                                    _lib.stream_ptr()))
        return out

    # ------------------------------------------------------------------ forward (models.py:134-162)
    def call(self, inputs, training=False):
        """(n,3) ids -> (n,) scores as a device tensor (models.py:134-162).  training=True: the Dropout layer on the three gathered rows
        is active like under Keras (every call draws fresh masks: the counter-based masks of the training kernels, seeded by this
        model's step counter, which the call advances); with the literal path the masks sit on the transformed rows' inputs as in
        a train step, so that path scores through the gradient half and discards the gradients."""
        t = as_triples(inputs, self.device, self.num_entities, self.num_relations)
        n = t.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        if training and self.dp > 0.0:
            if self._lit is not None:
                raise NotImplementedError("call(training=True) with the literal input path: use train_step(..., scores_out=...)")
            md = self._model_desc(True)
            self._seed += 1
            check(lib.kge_score_training(C.byref(md), self.E.data_ptr(), self.Rel.data_ptr(), t.data_ptr(), n, self._seed, 0, out.data_ptr(),
                                         self._err.data_ptr(), _lib.stream_ptr()))
            return out
        md = self._model_desc(False)
        Et = self._eval_table()
        check(lib.kge_score(C.byref(md), Et.data_ptr(), self.Rel.data_ptr(), t.data_ptr(), n, out.data_ptr(),
                            self._err.data_ptr(), _lib.stream_ptr()))
        return out

    __call__ = call

    def func(self, s, p, o, training=False):
        """Score already-gathered rows (models.py:160): rows are staged as temporary tables and scored by the same kernel (training=True:
        dropout on the three rows, as in call)."""
        s = torch.as_tensor(s, dtype=torch.float32, device=self.device).reshape(-1, self.e_dim)
        p = torch.as_tensor(p, dtype=torch.float32, device=self.device).reshape(-1, self.r_dim)
        o = torch.as_tensor(o, dtype=torch.float32, device=self.device).reshape(-1, self.e_dim)
        n = s.shape[0]
        Et = torch.cat([s, o], 0).contiguous()
        idx = torch.arange(n, dtype=torch.int32, device=self.device)
        t = torch.stack([idx, idx, idx + n], 1).contiguous()
        drop = bool(training) and self.dp > 0.0
        md = self._model_desc(drop)
        md.num_entities, md.num_relations = 2 * n, n
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        if drop:
            self._seed += 1
            check(lib.kge_score_training(C.byref(md), Et.data_ptr(), p.contiguous().data_ptr(), t.data_ptr(), n, self._seed, 0, out.data_ptr(),
                                         self._err.data_ptr(), _lib.stream_ptr()))
            return out
        check(lib.kge_score(C.byref(md), Et.data_ptr(), p.contiguous().data_ptr(), t.data_ptr(), n, out.data_ptr(),
                            self._err.data_ptr(), _lib.stream_ptr()))
        return out

    def predict(self, x, batch_size=32, verbose=0, **_):
        """keras Model.predict as used at utils.py:211,231: numpy (n,) scores.  One launch regardless of batch_size."""
        out = self.call(x).cpu().numpy()
        self._raise_on_flag()
        return out

    def _raise_on_flag(self):
        flag = int(self._err.item())
        if flag & 32:  # (kge_train_step_swap_chain: the bounded flag waits of its closer workgroups -- a device fault, never a data error)
            self._err.zero_()
            raise RuntimeError("chained training steps: a closer workgroup gave up waiting for the forward's partial rows (device fault); the tables are not to be trusted")
        if flag != 0:
            self._err.zero_()
            raise IndexError("triple ids out of range of the embedding tables (indices[...] is not in [0, N))")

    # ------------------------------------------------------------------ training (tests.py:63-89,130-134)
    def compile(self, optimizer='adam', loss=None, **kw):
        self.optimizer = optimizer if isinstance(optimizer, _Optimizer) else _Optimizer(optimizer)
        if self._gE is None:
            # dense gradient accumulators (kept all-zero between steps by the optimizer sweep) and the loss word live in ONE
            # flat buffer, so that data-parallel training needs a single all-reduce per step (kge_keras_amd.dist)
            nE, nR = self.E.numel(), self.Rel.numel()
            pad = (-(nE + nR)) % 4
            self._gall = torch.zeros(nE + nR + pad + 4, dtype=torch.float32, device=self.device)
            self._gE = self._gall[:nE].view_as(self.E)
            self._gR = self._gall[nE:nE + nR].view_as(self.Rel)
            self._loss = self._gall[nE + nR + pad:nE + nR + pad + 1]
        return self

    def _ensure_train_state(self, m_pos, m_neg):
        if self.optimizer is None:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
        if self.optimizer.kind == "adam" and self.optimizer.slots is None:
            self.optimizer.slots = tuple(torch.zeros_like(x) for x in (self.E, self.E, self.Rel, self.Rel))
        md, ld = self._model_desc(True), self._loss_desc()
        need = check(lib.kge_train_step_workspace_bytes(C.byref(md), C.byref(ld), m_pos, m_neg))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(int(need), dtype=torch.uint8, device=self.device)  # zero-filled once (kge_hip.h)
        return md, ld

    def _opt_desc(self):
        o = self.optimizer
        sl = [x.data_ptr() for x in o.slots] if o.slots else [None] * 4
        sd = getattr(self, "_step_dev", None)  # device-resident step counter while a hipGraph pipeline drives the steps
        return OptDesc(OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, max(1, o.iterations), *sl, sd.data_ptr() if sd is not None else None)

    def plan_supported(self):
        """True where a batch plan exists for this model / loss (the relation-grouped pull path of the HIP library)."""
        if self._lit is not None or type(self).train_step is not EmbeddingModel.train_step:
            return False
        md, ld = self._model_desc(True), self._loss_desc()
        return lib.kge_batch_plan_bytes(C.byref(md), C.byref(ld), 1, 1) > 0

    def pipeline_supported(self):
        """True where the whole step is ONE planned library call (what pipeline.TrainPipeline captures into its step graph); HolE's
        frequency-domain step takes plans too but is a composite of calls and runs batch by batch."""
        return type(self).train_step is EmbeddingModel.train_step and self.plan_supported()

    def plan_batch_counted(self, cand_rows, cand_counts, n_pos_cand, m_pos, m_neg, out=None, parts=7, pos_lanes=1):
        """BatchPlan of a batch given in folded form: cand_rows (rows of the batch before oversampling, positives first) with
        their multiplicities cand_counts (DeviceSampler.build_batch(..., counted=...): pos_lanes = _lib.CAND_POS_LANES striped
        counters for the positives); no hash table involved."""
        md, ld = self._model_desc(True), self._loss_desc()
        need = check(lib.kge_batch_plan_bytes(C.byref(md), C.byref(ld), int(m_pos), int(m_neg)))
        buf = out.buf if out is not None and out.buf.numel() >= need else torch.empty(int(need), dtype=torch.uint8, device=self.device)
        n_neg_cand = int(cand_counts.numel()) - int(pos_lanes) * int(n_pos_cand)
        check(lib.kge_batch_plan_build_counted(C.byref(md), C.byref(ld), cand_rows.data_ptr(), cand_counts.data_ptr(), int(pos_lanes),
                                               int(n_pos_cand), n_neg_cand, int(m_pos), int(m_neg), buf.data_ptr(), buf.numel(), int(parts),
                                               self._err.data_ptr(), _lib.stream_ptr()))
        return BatchPlan(buf, m_pos, m_neg)

    def plan_batch(self, data, out=None, parts=7):
        """Build the BatchPlan of [pos, neg] on the CURRENT stream (asynchronous).  `out`: a BatchPlan whose buffer is reused
        when large enough.  The caller orders the build before the step that consumes it (same stream, or an event).
        parts: 1 = rows, 2 = entity lists, 4 = relation lists (2 and 4 may run on two streams once 1 is done); 7 = all."""
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        md, ld = self._model_desc(True), self._loss_desc()
        need = check(lib.kge_batch_plan_bytes(C.byref(md), C.byref(ld), pos.shape[0], neg.shape[0]))
        buf = out.buf if out is not None and out.buf.numel() >= need else torch.empty(int(need), dtype=torch.uint8, device=self.device)
        check(lib.kge_batch_plan_build(C.byref(md), C.byref(ld), pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0], buf.data_ptr(),
                                       buf.numel(), int(parts), self._err.data_ptr(), _lib.stream_ptr()))
        return BatchPlan(buf, pos.shape[0], neg.shape[0])

    def grad_halves_supported(self):
        """True where train_grads(plan, parts=1 | 2) is available: the entity gradient rows and the relation half (forward, loss,
        relation gradient rows) as two independent calls (kge_train_grads_planned_parts)."""
        md, ld = self._model_desc(True), self._loss_desc()
        return lib.kge_train_grads_halves_supported(C.byref(md), C.byref(ld)) == 1

    def _step_planned(self, plan, apply, parts=3):
        md, ld = self._ensure_train_state(plan.m_pos, plan.m_neg)
        if parts != 2:
            self._seed += 1
        if not apply:
            check(lib.kge_train_grads_planned_parts(C.byref(md), C.byref(ld), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                                    self._gR.data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), plan.m_pos, plan.m_neg,
                                                    self._loss.data_ptr(), self._seed, self._ws.data_ptr(), self._ws.numel(), int(parts),
                                                    _lib.stream_ptr()))
            return {"loss": self._loss[0]}
        self.optimizer.iterations += 1
        od = self._opt_desc()
        tw = getattr(self, "_step_twins", None)  # (E twin, Rel twin) lent by a TrainPipeline: the step closed by its pull launch
        if tw is not None and od.step_dev and lib.kge_train_step_swap_supported(C.byref(md), C.byref(ld), C.byref(od)) == 1:
            seq = getattr(self, "_chain_seq", None)  # (a TrainPipeline's serial pass: the one-launch form of the step, like its native driver)
            if seq is not None:
                check(lib.kge_train_step_swap_chain(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), tw[0].data_ptr(),
                                                    tw[1].data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), plan.m_pos, plan.m_neg,
                                                    self._loss.data_ptr(), self._seed, self._ws.data_ptr(), self._ws.numel(), seq, self._err.data_ptr(),
                                                    _lib.stream_ptr()))
                self._chain_seq = seq + 1
                self.E.copy_(tw[0])
                self.Rel.copy_(tw[1])
                return {"loss": self._loss[0]}
            check(lib.kge_train_step_swap(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), tw[0].data_ptr(),
                                          tw[1].data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), plan.m_pos, plan.m_neg,
                                          self._loss.data_ptr(), self._seed, self._ws.data_ptr(), self._ws.numel(), 3, _lib.stream_ptr()))
            self.E.copy_(tw[0])
            self.Rel.copy_(tw[1])
            return {"loss": self._loss[0]}
        check(lib.kge_train_step_planned(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                         self._gR.data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), plan.m_pos, plan.m_neg,
                                         self._loss.data_ptr(), self._seed, self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def train_step(self, data, scores_out=None):
        """Default Keras train_step for the two-input wrapper: data = [pos (m,3), neg (m,3)] (optionally (X, y)), or a BatchPlan
        built from such a batch.  Returns {'loss': 0-d device tensor}."""
        if isinstance(data, BatchPlan):
            return self._step_planned(data, True)
        if isinstance(data, tuple) and len(data) == 2 and (data[1] is None or isinstance(data[0], (list, tuple))):
            data = data[0]
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        m_pos, m_neg = pos.shape[0], neg.shape[0]
        if self._lit is not None:
            return self._train_step_literal(pos, neg, scores_out)
        md, ld = self._ensure_train_state(m_pos, m_neg)
        self.optimizer.iterations += 1
        od = self._opt_desc()
        self._seed += 1
        check(lib.kge_train_step(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(),
                                 self._gE.data_ptr(), self._gR.data_ptr(), pos.data_ptr(), m_pos, neg.data_ptr(), m_neg,
                                 self._loss.data_ptr(), scores_out.data_ptr() if scores_out is not None else None,
                                 self._seed, self._ws.data_ptr(), self._ws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def _train_step_literal(self, pos, neg, scores_out=None):
        lt, o = self._lit, self.optimizer
        md, ld = self._ensure_train_state(0, 0)
        if o.kind == "adam" and lt["slots"] is None:
            lt["slots"] = (torch.zeros_like(lt["theta"]), torch.zeros_like(lt["theta"]))
        need = check(lib.kge_lit_workspace_bytes(C.byref(md), lt["L"], pos.shape[0], neg.shape[0]))
        if lt["ws"] is None or lt["ws"].numel() < need:
            lt["ws"] = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        o.iterations += 1
        self._seed += 1
        check(lib.kge_lit_train_grads(C.byref(md), C.byref(ld), lt["act"], self.E.data_ptr(), self.Rel.data_ptr(), lt["table"].data_ptr(),
                                      lt["L"], lt["table"].shape[0], lt["theta"].data_ptr(), self._gE.data_ptr(), self._gR.data_ptr(),
                                      lt["gtheta"].data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0],
                                      self._loss.data_ptr(), scores_out.data_ptr() if scores_out is not None else None, self._seed,
                                      lt["ws"].data_ptr(), lt["ws"].numel(), self._err.data_ptr(), _lib.stream_ptr()))
        od = self._opt_desc()
        check(lib.kge_opt_apply(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                self._gR.data_ptr(), 1.0, self._loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()))
        ts = [x.data_ptr() for x in lt["slots"]] if lt["slots"] else [None, None]
        odv = OptDesc(OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, o.iterations, ts[0], ts[1], None, None)
        check(lib.kge_opt_apply_vector(C.byref(odv), lt["theta"].data_ptr(), lt["gtheta"].data_ptr(), lt["theta"].numel(), 1.0,
                                       _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def train_grads(self, data, scores_out=None, parts=3):
        """First half of train_step (forward + loss + gradient accumulation); for data-parallel training an
        all-reduce of `_gE` / `_gR` goes between this and apply_grads.  With a BatchPlan and grad_halves_supported(): parts = 1
        the entity gradient rows (`_gE`) alone, parts = 2 the relation half (`_gR`, the loss) -- two independent calls on one plan,
        so the exchange of `_gE`, the bulk of the bytes, can run while the relation half is still computing."""
        if isinstance(data, BatchPlan):
            return self._step_planned(data, False, parts)
        if parts != 3:
            raise ValueError("train_grads(parts=1 | 2) needs a BatchPlan")
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        md, ld = self._ensure_train_state(pos.shape[0], neg.shape[0])
        self._seed += 1
        check(lib.kge_train_grads(C.byref(md), C.byref(ld), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                  self._gR.data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0],
                                  self._loss.data_ptr(), scores_out.data_ptr() if scores_out is not None else None, self._seed,
                                  self._ws.data_ptr(), self._ws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def apply_grads(self, grad_scale=1.0):
        """Second half of train_step: optimizer update from the (reduced) accumulators * grad_scale; re-zeroes them."""
        md, ld = self._model_desc(True), self._loss_desc()
        self.optimizer.iterations += 1
        od = self._opt_desc()
        check(lib.kge_opt_apply(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(),
                                self._gE.data_ptr(), self._gR.data_ptr(), float(grad_scale), self._loss.data_ptr(),
                                self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()))

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            shuffle=True, **_):
        """keras Model.fit over a Sequence-style generator (tests.py:130-134): len(x) batches per epoch,
        x[i] -> ([pos, neg], None), x.on_epoch_end() after each epoch.  Returns a History-like object."""
        import random
        import time
        callbacks = callbacks or []
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        history = {"loss": []}
        if validation_data is not None:
            history["val_loss"] = []
        for epoch in range(epochs):
            t0 = time.time()
            order = list(range(len(x)))
            if shuffle:
                random.shuffle(order)  # Keras shuffles the batch order of a Sequence with Python's `random`
            tot = torch.zeros((), dtype=torch.float64, device=self.device)
            if hasattr(x, "run_epoch") and self.pipeline_supported():
                # device generator + a model with batch plans: draw | plan | step pipelined on the GPU, replayed from hipGraphs
                tot = x.run_epoch(self, order)
            else:
                it = x.batches(order) if hasattr(x, "batches") else (x[i][0] for i in order)  # device generators prefetch on a side stream
                for X in it:
                    tot += self.train_step(X)["loss"].double()
            logs = {"loss": float(tot.item()) / max(1, len(order))}
            self._raise_on_flag()
            if validation_data is not None:
                logs["val_loss"] = self.evaluate(validation_data)
            if hasattr(x, "on_epoch_end"):
                x.on_epoch_end()
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            for k, v in logs.items():
                history.setdefault(k, []).append(v)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items() if isinstance(v, float)))
            if self.stop_training:
                break
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        return type("History", (), {"history": history, "model": self})()

    def evaluate(self, x, **_):
        """Mean batch loss over a generator without updating the tables (keras Model.evaluate over add_loss; what fit reports as
        val_loss, tests.py:130-134).  Forward only (kge_loss_only); the batch losses are summed on the device -- one host read per
        evaluation, not per batch; the inference-mode table of the literal path and the l3_reg term are formed once."""
        n = len(x)
        if n == 0:
            return 0.0
        acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        ctx = self._loss_only_context()
        for i in range(n):
            X, _y = x[i]
            self._batch_loss_into(X, ctx, acc)
        tot = float(acc.item()) / n
        if self.regularization:
            tot += float(self._l3_term(ctx).item())
        return tot

    def _loss_only_context(self):
        """What a run of forward-only losses shares: the inference-mode entity table (transformed once with the literal path)."""
        return {"Et": self._eval_table()}

    def _loss_ws(self, m_pos, m_neg):
        need = check(lib.kge_loss_workspace_bytes(int(m_pos), int(m_neg)))
        if getattr(self, "_lws", None) is None or self._lws.numel() < need:
            self._lws = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        return self._lws

    def _batch_loss_into(self, data, ctx, acc=None, out=None):
        """Forward-only loss of one [pos, neg] batch (no regulariser) -> out (1 float, device); acc (1 double, device) += it."""
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        md, ld = self._model_desc(False), self._loss_desc()
        ws = self._loss_ws(pos.shape[0], neg.shape[0])
        if out is None:
            out = getattr(self, "_vloss", None)
            if out is None:
                out = self._vloss = torch.zeros(1, dtype=torch.float32, device=self.device)
        check(lib.kge_loss_only(C.byref(md), C.byref(ld), ctx["Et"].data_ptr(), self.Rel.data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(),
                                neg.shape[0], out.data_ptr(), acc.data_ptr() if acc is not None else None, ws.data_ptr(), ws.numel(),
                                self._err.data_ptr(), _lib.stream_ptr()))
        return out

    def _l3_term(self, ctx=None):
        """regularization * sum |E|^3 over the whole entity table (models.py:10-11,78-92) as a 1-element device tensor."""
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        ws = self._loss_ws(1, 1)
        check(lib.kge_l3_value(self.E.data_ptr(), self.E.numel(), float(self.regularization), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return out

    def batch_loss(self, data):
        """Loss of one [pos, neg] batch in inference mode (no dropout), no update: what Keras reports as val_loss (the added loss + the
        regularisation term).  Forward only: two score launches and the loss on them (kge_loss_only) -- through the literal transform
        when the model has one; no gradient half runs and no optimizer is involved."""
        ctx = self._loss_only_context()
        loss = self._batch_loss_into(data, ctx, out=torch.zeros(1, dtype=torch.float32, device=self.device))
        if self.regularization:
            loss = loss + self._l3_term(ctx)
        return loss[0]

    # ------------------------------------------------------------------ filtered 1-vs-all ranking (utils.py:165-264)
    @staticmethod
    def _filter_coo(test, num_entities, filt, side):
        """COO (query index, excluded id) of the reference's filter sets; ValueError where list.remove would raise."""
        qs, ids = [], []
        for i, (s, p, o) in enumerate(test):
            target, key = (o, (s, p)) if side == 0 else (s, (p, o))
            if not 0 <= target < num_entities:
                raise ValueError("list.remove(x): x not in list")
            fl = filt.get(key) if hasattr(filt, "get") else filt[key]
            if not fl:
                continue
            for x in fl:
                if x == target or not 0 <= x < num_entities:
                    raise ValueError("list.remove(x): x not in list")  # utils.py:168-171 / :184-188
            qs.extend([i] * len(fl))
            ids.extend(fl)
        return np.asarray(qs, dtype=np.int32), np.asarray(ids, dtype=np.int32)

    def rank_counts(self, triples, side, filt_q, filt_ids, num_entities=None, cand_offset=0, cand_rows=None, counts=None):
        """counts[i] = #{kept candidates c in the range with score(i,c) > score(i,target_i)} on the GPU.  side 0: tail queries,
        1: head queries, 2: both in one sweep -- 2 n counts, tail queries first; filt_q numbers the queries the same way."""
        t = as_triples(triples, self.device, self.num_entities, self.num_relations)
        nt = t.shape[0]
        nq = nt * (2 if int(side) == 2 else 1)
        num_entities = self.num_entities if num_entities is None else int(num_entities)
        cand_rows = num_entities - cand_offset if cand_rows is None else cand_rows
        md = self._model_desc(False)
        need = check(lib.kge_rank_workspace_bytes(C.byref(md), nq))
        if getattr(self, "_rank_ws", None) is None or self._rank_ws.numel() < need:
            self._rank_ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        def dev_i32(a):  # device tensors are taken as they are (no host round trip per call)
            if isinstance(a, torch.Tensor) and a.device.type == "cuda":
                return a.to(dtype=torch.int32).contiguous()
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)
        fq, fi = dev_i32(filt_q), dev_i32(filt_ids)
        acc = counts is not None
        if counts is None:
            counts = torch.empty(nq, dtype=torch.int32, device=self.device)
        Et = self._eval_table() if self._lit is not None else self.E
        ec_ptr = Et.data_ptr() + 4 * int(cand_offset) * int(self.e_dim)
        check(lib.kge_rank_counts(C.byref(md), Et.data_ptr(), self.Rel.data_ptr(), ec_ptr, int(cand_offset), int(cand_rows),
                                  t.data_ptr(), nt, int(side), fq.data_ptr() if fq.numel() else None,
                                  fi.data_ptr() if fi.numel() else None, fq.numel(), counts.data_ptr(), 1 if acc else 0, None,
                                  self._rank_ws.data_ptr(), self._rank_ws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        return counts

    def rank_counts_shard(self, ref_rows, triples_local, side, filt_q, filt_ids, shard, lo, counts=None):
        """SURVEY 8(e) schedule (ii), one rank's share: counts[i] = #{kept c in [lo, lo + len(shard)) : score(i, c) >
        score(i, target_i)} with the entity table row-sharded.  ref_rows: the rows the queries reference (dist.
        exchange_referenced_rows), triples_local: the queries with s / o given as positions in ref_rows; filters carry GLOBAL
        entity ids (the kernels skip ids outside this shard).  kge_rank_counts with E = ref_rows, E_cand = shard."""
        ref_rows = ref_rows.contiguous()
        shard = shard.contiguous()
        t = as_triples(triples_local, self.device, ref_rows.shape[0], self.num_relations)
        nt = t.shape[0]
        nq = nt * (2 if int(side) == 2 else 1)  # (side 2: both sides in one sweep of the shard, 2 n counts, tail queries first)
        md = self._model_desc(False)
        md.num_entities = int(ref_rows.shape[0])
        need = check(lib.kge_rank_workspace_bytes(C.byref(md), nq))
        if getattr(self, "_rank_ws", None) is None or self._rank_ws.numel() < need:
            self._rank_ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        fq = torch.as_tensor(filt_q, dtype=torch.int32, device=self.device).contiguous()
        fi = torch.as_tensor(filt_ids, dtype=torch.int32, device=self.device).contiguous()
        acc = counts is not None
        if counts is None:
            counts = torch.empty(nq, dtype=torch.int32, device=self.device)
        check(lib.kge_rank_counts(C.byref(md), ref_rows.data_ptr(), self.Rel.data_ptr(), shard.data_ptr(), int(lo), int(shard.shape[0]),
                                  t.data_ptr(), nt, int(side), fq.data_ptr() if fq.numel() else None,
                                  fi.data_ptr() if fi.numel() else None, fq.numel(), counts.data_ptr(), 1 if acc else 0, None,
                                  self._rank_ws.data_ptr(), self._rank_ws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        return counts

    def rank(self, test_data, num_entities, filter_t, filter_h, chunk=32768):
        """(tail_ranks, head_ranks) of `validate` (utils.py:199-264): real_rank = 1 + #{kept c: score_c > score_target}."""
        test = [tuple(int(v) for v in t) for t in test_data]
        tails, heads = np.empty(len(test), dtype=np.int64), np.empty(len(test), dtype=np.int64)
        for a in range(0, len(test), chunk):  # both sides of a chunk in one sweep of the table (kge_rank_counts, side 2)
            part = test[a:a + chunk]
            n = len(part)
            fq0, fi0 = self._filter_coo(part, num_entities, filter_t, 0)
            fq1, fi1 = self._filter_coo(part, num_entities, filter_h, 1)
            cnt = self.rank_counts(np.asarray(part, dtype=np.int32), 2, np.concatenate([fq0, fq1 + n]).astype(np.int32),
                                   np.concatenate([fi0, fi1]).astype(np.int32), num_entities).cpu().numpy().astype(np.int64)
            tails[a:a + n], heads[a:a + n] = 1 + cnt[:n], 1 + cnt[n:]
        self._raise_on_flag()
        return tails, heads

    def rank_triples(self, test, num_entities, filt, chunk=65536):
        """Same as rank(), from raw (n,3) arrays: the filter index is built by the C++ front-end (utils.filter_coo)."""
        from .utils import filter_coo
        test = np.ascontiguousarray(np.asarray(test, dtype=np.int32).reshape(-1, 3))
        tails, heads = np.empty(len(test), dtype=np.int64), np.empty(len(test), dtype=np.int64)
        for a in range(0, len(test), chunk):  # both sides of a chunk in one sweep of the table (kge_rank_counts, side 2)
            part = test[a:a + chunk]
            n = len(part)
            fq0, fi0 = filter_coo(part, 0, filt, num_entities)
            fq1, fi1 = filter_coo(part, 1, filt, num_entities)
            cnt = self.rank_counts(part, 2, np.concatenate([fq0, fq1 + n]).astype(np.int32), np.concatenate([fi0, fi1]).astype(np.int32),
                                   num_entities).cpu().numpy().astype(np.int64)
            tails[a:a + n], heads[a:a + n] = 1 + cnt[:n], 1 + cnt[n:]
        self._raise_on_flag()
        return tails, heads

    # ------------------------------------------------------------------ weights I/O (checkpointing)
    def _extra_state(self):
        """Every trainable or stateful tensor besides the two tables, by name (subclasses add theirs)."""
        st = {}
        if self._lit is not None:
            st["lit_theta"] = self._lit["theta"]
            if self._lit["slots"] is not None:
                st["lit_slot_m"], st["lit_slot_v"] = self._lit["slots"]
        return st

    def _set_extra_state(self, st):
        if self._lit is not None:
            self._lit["theta"].copy_(st["lit_theta"])
            if "lit_slot_m" in st:
                self._lit["slots"] = (st["lit_slot_m"].to(self.device).clone(), st["lit_slot_v"].to(self.device).clone())

    def get_weights(self):
        out = self.entity_embedding.get_weights() + self.relational_embedding.get_weights()
        if self._lit is not None:
            out += list(self.get_literal_dense())
        return out

    def set_weights(self, ws):
        want = 2 + (2 if self._lit is not None else 0)
        if len(ws) != want:
            raise ValueError(f"You called `set_weights(weights)` on {self.name} with a weight list of length {len(ws)}, but it expects {want} weights")
        self.entity_embedding.set_weights([ws[0]])
        self.relational_embedding.set_weights([ws[1]])
        if self._lit is not None:
            W, b = np.asarray(ws[2], dtype=np.float32), np.asarray(ws[3], dtype=np.float32)
            K1, d = int(self.e_dim) + self._lit["L"], int(self.e_dim)
            if W.shape != (K1, d) or b.shape != (d,):
                raise ValueError(f"literal Dense weights of shapes {W.shape}, {b.shape}; expected {(K1, d)}, {(d,)}")
            self.set_literal_dense(W, b)

    def save_weights(self, path):
        """Everything a resumed run needs: both tables, every other trainable / stateful tensor (literal Dense, ConvE head and its
        batch-norm statistics), the optimizer's kind, step count and all its slots, the dropout step seed."""
        st = {"format": 2, "model": type(self).__name__, "shapes": {"E": tuple(self.E.shape), "Rel": tuple(self.Rel.shape)},
              "E": self.E.cpu(), "Rel": self.Rel.cpu(), "seed": self._seed,
              "extra": {k: v.cpu() for k, v in self._extra_state().items()}}
        if self.optimizer is not None:
            st["opt"] = dict(kind=self.optimizer.kind, iterations=self.optimizer.iterations,
                             slots=[s.cpu() for s in self.optimizer.slots] if self.optimizer.slots else None)
        torch.save(st, path)

    def load_weights(self, path):
        """Restores what save_weights wrote.  Checkpoints of the first layout of this package ({E, Rel, model[, opt]}: tables and
        optimizer only, no `format` key) are read as format 1: models with further trainable tensors (literal Dense, ConvE head)
        cannot be resumed from those.  A checkpoint that carries optimizer state loaded into a model that is not compiled yet
        would silently lose the moments and the step count, so that warns: compile first, then load."""
        st = torch.load(path, map_location="cpu")
        if not isinstance(st, dict) or "E" not in st or "Rel" not in st or "model" not in st:
            raise ValueError(f"{path}: not a checkpoint of this package")
        fmt = st.get("format", 1)
        if fmt == 1:
            if self._extra_state():
                raise ValueError(f"{path} is a format-1 checkpoint (tables and optimizer only); this model has further trainable tensors {sorted(self._extra_state())}")
            st = dict(st, extra={}, format=2)
        elif fmt != 2:
            raise ValueError(f"{path}: checkpoint format {fmt!r} is newer than this package reads (1, 2)")
        if "opt" in st and self.optimizer is None:
            import warnings
            warnings.warn(f"{path} holds optimizer state ({st['opt']['kind']}, {st['opt']['iterations']} updates) but the model is not compiled: "
                          "only the tables are restored -- call compile() before load_weights() to resume training", stacklevel=2)
        if st["model"] != type(self).__name__:
            raise ValueError(f"{path} holds a {st['model']} model, this is a {type(self).__name__}")
        for k, t in (("E", self.E), ("Rel", self.Rel)):
            if tuple(st[k].shape) != tuple(t.shape):
                raise ValueError(f"{path}: table {k} has shape {tuple(st[k].shape)}, the model expects {tuple(t.shape)}")
        mine = self._extra_state()
        for k, v in st["extra"].items():
            if k in mine and tuple(v.shape) != tuple(mine[k].shape):
                raise ValueError(f"{path}: tensor {k} has shape {tuple(v.shape)}, the model expects {tuple(mine[k].shape)}")
        missing = [k for k in mine if k not in st["extra"] and not k.endswith(("_m", "_v"))]
        if missing:
            raise ValueError(f"{path} lacks {missing}")
        self.E.copy_(st["E"])
        self.Rel.copy_(st["Rel"])
        self._seed = int(st.get("seed", self._seed))
        self._set_extra_state(st["extra"])
        if "opt" in st and self.optimizer is not None:
            if st["opt"]["kind"] != self.optimizer.kind:
                raise ValueError(f"{path} was written with optimizer {st['opt']['kind']!r}, the model is compiled with {self.optimizer.kind!r}")
            self.optimizer.iterations = int(st["opt"]["iterations"])
            if st["opt"]["slots"] is not None:
                shapes = [tuple(x.shape) for x in (self.E, self.E, self.Rel, self.Rel)]
                if [tuple(s.shape) for s in st["opt"]["slots"]] != shapes:
                    raise ValueError(f"{path}: optimizer slots do not match the tables' shapes")
                self.optimizer.slots = tuple(s.to(self.device).clone() for s in st["opt"]["slots"])


class DistMult(EmbeddingModel):
    """models.py:164-172."""
    MODEL = "DistMult"

    def __init__(self, name='DistMult', **kwargs):
        super().__init__(name=name, **kwargs)


class TransE(EmbeddingModel):
    """models.py:175-190: score = gamma - ||s+p-o||_norm if gamma > 0 else +||s+p-o||_norm."""
    MODEL = "TransE"

    def __init__(self, name='TransE', norm=1, gamma=12, **kwargs):
        super().__init__(name=name, **kwargs)
        self.gamma = gamma
        self.norm = norm

    def _model_desc(self, training=False):
        """tf.norm(x, ord=self.norm, axis=1) (models.py:188): ord 1 and 2 have their own kernels (library norm ids 1, 2); any other
        positive real p is the general p-norm (id 5, exponent in the descriptor's `modulus` field).  'euclidean' = 2.  ord = inf
        (max |x|) and the matrix norms tf.norm also accepts are not built."""
        md = super()._model_desc(training)
        n = 2 if self.norm == 'euclidean' else self.norm
        if isinstance(n, bool) or not isinstance(n, (int, float, np.integer, np.floating)) or not np.isfinite(n) or n <= 0:
            raise NotImplementedError(f"TransE norm {self.norm!r}: positive real orders (and 'euclidean') are built")
        if n in (1, 2):
            md.norm = int(n)
        else:
            md.norm, md.modulus = 5, float(n)
        return md


class ComplEx(EmbeddingModel):
    """models.py:203-222: both widths are doubled (so e_dim and r_dim must be passed as kwargs, as in the reference)."""
    MODEL = "ComplEx"

    def __init__(self, name='ComplEx', **kwargs):
        kwargs['e_dim'] = 2 * kwargs['e_dim']  # KeyError if absent, like models.py:208
        kwargs['r_dim'] = 2 * kwargs['r_dim']
        super().__init__(name=name, **kwargs)


class pRotatE(EmbeddingModel):
    """models.py:461-489: score = gamma - modulus * sum_k |sin((phase_s + phase_p - phase_o)_k / 2)| (or +... when gamma <= 0), both
    widths doubled (:469-470), modulus = modulus * embedding_range with embedding_range = (gamma + epsilon) / e_dim / 2 taken on
    the doubled e_dim (:475-476).  The reference wraps every row into (-pi, pi] first (atan2(sin x, cos x), :481-483); a wrap
    moves the half-angle by a multiple of pi, which |sin| does not see, so the kernels take the rows as they are.  Structurally
    this is TransE with the element function |sin(x/2)| in place of |x|: the TransE kernels with norm id 3 (every training path,
    batch plans, the distance sweep of the 1-vs-all ranking)."""
    MODEL = "TransE"

    def __init__(self, gamma=12, epsilon=2, modulus=0.5, name='pRotatE', **kwargs):
        kwargs['e_dim'] = 2 * kwargs['e_dim']  # KeyError if absent, like models.py:469
        kwargs['r_dim'] = 2 * kwargs['r_dim']
        super().__init__(name=name, **kwargs)
        self.gamma = gamma
        self.epsilon = epsilon
        self.norm = 3
        self.pi = np.pi
        self.embedding_range = (self.gamma + self.epsilon) / self.e_dim / 2
        self.modulus = modulus * self.embedding_range


class RotatE(EmbeddingModel):
    """models.py:421-459, exactly as written: e_dim doubled, r_dim NOT (:428-429: the relation rows are rotation phases, one per
    complex component); phase = atan2(sin p, cos p) (:443); the rotated subject minus the object is summed WITH its signs (:446-453;
    the constructor's `norm` is never applied), so the score is linear in s and in o -- gamma - sum if gamma > 0 else the sum
    (:455-458).  Own model id in the library (KGE_ROTATE): scoring kernel, push-path training (forward + loss + atomic gradient
    rows; no batch plans, so fit() steps batch by batch), and the 1-vs-all ranking as a dot-product sweep on the matrix cores (a
    linear score orders the candidates by <q, e_c>)."""
    MODEL = "RotatE"

    def __init__(self, gamma=12, norm=2, epsilon=2, name='RotatE', **kwargs):
        kwargs['e_dim'] = 2 * kwargs['e_dim']  # KeyError if absent, like models.py:428
        kwargs['r_dim'] = kwargs['r_dim']
        super().__init__(name=name, **kwargs)
        self.gamma = gamma
        self.epsilon = epsilon
        self.norm = norm
        self.pi = np.pi
        self.embedding_range = (self.gamma + self.epsilon) / self.e_dim / 2

    def _model_desc(self, training=False):
        md = super()._model_desc(training)
        md.norm = 1  # the constructor's `norm` takes no part in the score (models.py:446-459)
        return md


class HAKE(EmbeddingModel):
    """models.py:362-400, exactly as written: entity rows [phase | modulus] (e_dim doubled), relation rows [phase | modulus | bias]
    (r_dim tripled); score = gamma - (phase_weight * sum_k |sin((phase_s + phase_p - phase_o) / 2)| + mod_weight * r) where r is
    tf.norm(..., ord=2) WITHOUT an axis (:398): one scalar over every row of the tensor handed to that call of the model.  A
    triple's score therefore depends on the rows it is scored with, and this class reproduces the three ways the reference calls
    the model: `model(x)` -- all rows one call; `model.predict(x)` -- Keras' batches of 32 rows (utils.py:211,231), so `validate`
    ranks candidate lists whose scores carry a per-32-row offset; the train step -- one call on the positives, one on the
    negatives (tests.py:77-78).  Own entry points in the library (kge_hake_*): forward rows + per-call norms, the loss kernels, the
    gradient rows by float atomics, then the dense optimizer sweep (kge_opt_apply); no batch plans, no matrix-core sweep."""
    MODEL = "HAKE"

    def __init__(self, epsilon=2, gamma=12, phase_weight=0.5, mod_weight=1, name='HAKE', **kwargs):
        kwargs['e_dim'] = 2 * kwargs['e_dim']  # KeyError if absent, like models.py:371-372
        kwargs['r_dim'] = 3 * kwargs['r_dim']
        super().__init__(name=name, **kwargs)
        self.gamma = gamma
        self.epsilon = epsilon
        self.phase_weight = phase_weight
        self.mod_weight = mod_weight
        self.pi = np.pi
        self.embedding_range = (self.gamma + self.epsilon) / self.e_dim / 2
        self._hws = None

    def _hake(self):
        return _lib.HakeParams(float(self.phase_weight), float(self.mod_weight), float(self.embedding_range))

    def _hake_ws(self, n, chunk):
        need = check(lib.kge_hake_workspace_bytes(n, chunk))
        if self._hws is None or self._hws.numel() < need:
            self._hws = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        return self._hws

    def _score(self, t, chunk, seg_off=None, md=None, E=None, Rel=None):
        n = t.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        md, hp = md or self._model_desc(False), self._hake()
        n_seg = 0 if seg_off is None else seg_off.numel() - 1
        ws = self._hake_ws(n, -n_seg if n_seg else (chunk if 0 < chunk < n else n))  # (scoring: no loss scratch; chunk 0 sizes a training batch)
        if E is None:
            E = self._eval_table()  # (E itself, or with the literal path the transformed table: models.py:152-158 in inference mode)
        check(lib.kge_hake_score(C.byref(md), C.byref(hp), E.data_ptr(), (self.Rel if Rel is None else Rel).data_ptr(),
                                 t.data_ptr(), n, chunk, seg_off.data_ptr() if n_seg else None, n_seg, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                 self._err.data_ptr(), _lib.stream_ptr()))
        return out

    def call(self, inputs, training=False):
        """model(inputs): every row of `inputs` in ONE call -- one modulus norm over all of them (models.py:398)."""
        return self._score(as_triples(inputs, self.device, self.num_entities, self.num_relations), 0)

    __call__ = call

    def func(self, s, p, o, training=False):
        """Score already-gathered rows (models.py:383-400) as one call: rows staged as temporary tables, same kernels."""
        s = torch.as_tensor(s, dtype=torch.float32, device=self.device).reshape(-1, self.e_dim)
        p = torch.as_tensor(p, dtype=torch.float32, device=self.device).reshape(-1, self.r_dim)
        o = torch.as_tensor(o, dtype=torch.float32, device=self.device).reshape(-1, self.e_dim)
        n = s.shape[0]
        idx = torch.arange(n, dtype=torch.int32, device=self.device)
        md = self._model_desc(False)
        md.num_entities, md.num_relations = 2 * n, n
        return self._score(torch.stack([idx, idx, idx + n], 1).contiguous(), 0, md=md, E=torch.cat([s, o], 0).contiguous(), Rel=p.contiguous())

    def predict(self, x, batch_size=32, verbose=0, **_):
        """keras Model.predict (utils.py:211,231): the model is called on consecutive batches of `batch_size` rows (32 unless the
        caller says otherwise), each with its own modulus norm; numpy (n,) scores."""
        out = self._score(as_triples(x, self.device, self.num_entities, self.num_relations), int(batch_size) if batch_size else 32).cpu().numpy()
        self._raise_on_flag()
        return out

    # ------------------------------------------------------------------ training
    def plan_supported(self):
        return False

    def _grads(self, pos, neg, md, ld, scores_out, seed):
        hp = self._hake()
        if self._lit is not None:  # models.py:152-158 in front of HAKE.func: the model on the batch's transformed rows as staged tables
            lt = self._lit
            need = check(lib.kge_lit_workspace_bytes(C.byref(md), lt["L"], pos.shape[0], neg.shape[0]))
            if lt["ws"] is None or lt["ws"].numel() < need:
                lt["ws"] = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
            check(lib.kge_lit_hake_train_grads(C.byref(md), C.byref(ld), C.byref(hp), lt["act"], self.E.data_ptr(), self.Rel.data_ptr(), lt["table"].data_ptr(),
                                               lt["L"], lt["table"].shape[0], lt["theta"].data_ptr(), self._gE.data_ptr(), self._gR.data_ptr(),
                                               lt["gtheta"].data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0], self._loss.data_ptr(),
                                               scores_out.data_ptr() if scores_out is not None else None, seed, lt["ws"].data_ptr(), lt["ws"].numel(),
                                               self._err.data_ptr(), _lib.stream_ptr()))
            return
        ws = self._hake_ws(pos.shape[0] + neg.shape[0], 0)
        check(lib.kge_hake_train_grads(C.byref(md), C.byref(ld), C.byref(hp), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                       self._gR.data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0], self._loss.data_ptr(),
                                       scores_out.data_ptr() if scores_out is not None else None, seed, ws.data_ptr(), ws.numel(),
                                       self._err.data_ptr(), _lib.stream_ptr()))

    def train_grads(self, data, scores_out=None, parts=3):
        if isinstance(data, BatchPlan) or parts != 3:
            raise ValueError("HAKE trains batch by batch: no batch plans")
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        md, ld = self._ensure_train_state(pos.shape[0], neg.shape[0])
        self._seed += 1
        self._grads(pos, neg, md, ld, scores_out, self._seed)
        return {"loss": self._loss[0]}

    def train_step(self, data, scores_out=None):
        """Default Keras train_step of the two-input wrapper (tests.py:63-89): one call of the model per side, loss, gradients,
        optimizer update (+ the l3 regulariser's dense gradient inside the sweep)."""
        if isinstance(data, tuple) and len(data) == 2 and (data[1] is None or isinstance(data[0], (list, tuple))):
            data = data[0]
        self.train_grads(data, scores_out)
        self.apply_grads(1.0)
        if self._lit is not None:  # the Dense of the literal path
            lt, o = self._lit, self.optimizer
            if o.kind == "adam" and lt["slots"] is None:
                lt["slots"] = (torch.zeros_like(lt["theta"]), torch.zeros_like(lt["theta"]))
            ts = [x.data_ptr() for x in lt["slots"]] if lt["slots"] else [None, None]
            odv = OptDesc(OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, o.iterations, ts[0], ts[1], None, None)
            check(lib.kge_opt_apply_vector(C.byref(odv), lt["theta"].data_ptr(), lt["gtheta"].data_ptr(), lt["theta"].numel(), 1.0, _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def _batch_loss_into(self, data, ctx, acc=None, out=None):
        """Forward-only loss in inference mode: the wrapper calls the model once per side (tests.py:77-78), so every side is ONE segment of
        the call-wide norm (kge_hake_score, chunk = all rows); the loss on the two score vectors by kge_loss_value."""
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        sp, sn = self._score(pos, 0), self._score(neg, 0)
        sc = torch.cat([sp, sn])
        ld = self._loss_desc()
        ws = self._loss_ws(pos.shape[0], neg.shape[0])
        if out is None:
            out = getattr(self, "_vloss", None)
            if out is None:
                out = self._vloss = torch.zeros(1, dtype=torch.float32, device=self.device)
        check(lib.kge_loss_value(C.byref(ld), sc.data_ptr(), pos.shape[0], sc[pos.shape[0]:].data_ptr(), neg.shape[0], out.data_ptr(), None, None,
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        if acc is not None:
            acc += out.double()
        return out

    def _loss_only_context(self):
        return {}

    # ------------------------------------------------------------------ ranking (utils.py:165-264)
    def rank_counts(self, *a, **kw):
        raise NotImplementedError("a HAKE score depends on the rows it is predicted with: rank() scores the reference's candidate lists")

    rank_counts_shard = rank_counts

    def _rank_side(self, part, N, fq, fi, side):
        """Ranks of one side of a chunk of test triples, by the reference's own route: the candidate list of every query (target
        first, then the kept entities ascending: utils.py:165-196) is predicted in batches of 32 rows and the target's score is
        compared within the list (utils.py:211-215).  All lists of the chunk go through ONE scoring launch, with the predict
        batches as segments."""
        nq = len(part)
        tcol, ccol = (2, 0) if side == 0 else (0, 2)
        keep = np.ones((nq, N), dtype=bool)
        keep[np.arange(nq), part[:, tcol]] = False
        if len(fq):
            keep[fq, fi] = False
        qi, ids = np.nonzero(keep)  # (ascending ids within a query)
        n_q = 1 + keep.sum(1).astype(np.int64)
        q_off = np.zeros(nq + 1, dtype=np.int64)
        np.cumsum(n_q, out=q_off[1:])
        n = int(q_off[-1])
        cand = np.empty(n, dtype=np.int32)
        owner = np.empty(n, dtype=np.int64)
        cand[q_off[:-1]] = part[:, tcol]
        owner[q_off[:-1]] = np.arange(nq)
        rest = np.ones(n, dtype=bool)
        rest[q_off[:-1]] = False
        cand[rest], owner[rest] = ids, qi
        rows = np.empty((n, 3), dtype=np.int32)
        rows[:, 1] = part[owner, 1]
        rows[:, ccol] = part[owner, ccol]
        rows[:, tcol] = cand
        seg = np.concatenate([np.arange(q_off[i], q_off[i + 1], 32) for i in range(nq)] + [q_off[-1:]]).astype(np.int64)
        t = torch.from_numpy(rows).to(self.device)
        sc = self._score(t, 32, torch.from_numpy(seg).to(self.device))
        cnt = torch.empty(nq, dtype=torch.int32, device=self.device)
        qo = torch.from_numpy(q_off).to(self.device)
        check(lib.kge_hake_rank_counts(sc.data_ptr(), n, qo.data_ptr(), nq, cnt.data_ptr(), _lib.stream_ptr()))
        return 1 + cnt.cpu().numpy().astype(np.int64)

    def rank_triples(self, test, num_entities, filt, chunk=256):
        from .utils import filter_coo
        test = np.ascontiguousarray(np.asarray(test, dtype=np.int32).reshape(-1, 3))
        N = int(num_entities)
        chunk = max(1, min(chunk, (1 << 26) // max(1, N)))  # candidate rows of a chunk: at most 64 M
        tails, heads = np.empty(len(test), dtype=np.int64), np.empty(len(test), dtype=np.int64)
        for a in range(0, len(test), chunk):
            part = test[a:a + chunk]
            tails[a:a + chunk] = self._rank_side(part, N, *filter_coo(part, 0, filt, N), 0)
            heads[a:a + chunk] = self._rank_side(part, N, *filter_coo(part, 1, filt, N), 1)
        self._raise_on_flag()
        return tails, heads

    def rank(self, test_data, num_entities, filter_t, filter_h, chunk=256):
        test = np.asarray([tuple(int(v) for v in t) for t in test_data], dtype=np.int32).reshape(-1, 3)
        N = int(num_entities)
        chunk = max(1, min(chunk, (1 << 26) // max(1, N)))
        tails, heads = np.empty(len(test), dtype=np.int64), np.empty(len(test), dtype=np.int64)
        for a in range(0, len(test), chunk):
            part = test[a:a + chunk]
            lst = [tuple(int(v) for v in t) for t in part]
            tails[a:a + chunk] = self._rank_side(part, N, *self._filter_coo(lst, N, filter_t, 0), 0)
            heads[a:a + chunk] = self._rank_side(part, N, *self._filter_coo(lst, N, filter_h, 1), 1)
        self._raise_on_flag()
        return tails, heads


class HolE(EmbeddingModel):
    """models.py:224-237: score = sum_k p_k * ccorr(s, o)_k.

    Training steps without dropout run in the frequency domain (csrc/hole_spec.hip): by Parseval the score is the ComplEx score of
    the half spectra of the three rows, so a step transforms the two tables once (one GEMM each with the real DFT basis), runs
    the O(d) ComplEx-shaped kernels -- batch plans and the pipeline included -- on the spectra and carries the gradient tables
    back through the transposed bases; the optimizer (and the L3 regulariser) act on the real tables as before.
    Steps with dropout (every triple masks its own rows before the correlation), predict and the query vectors of the 1-vs-all
    sweep use the direct O(d^2) LDS-tiled kernels of csrc/hole.hip.  `spectral=False` (or KGE_HOLE_SPECTRAL=0) keeps the direct
    kernels for every step."""
    MODEL = "HolE"

    def __init__(self, name='HolE', **kwargs):
        super().__init__(name=name, **kwargs)
        self._spec = None

    # ---- frequency-domain training path
    def _spectral_on(self):
        import os
        if self.dp != 0 or self._lit is not None or int(self.e_dim) != int(self.r_dim):
            return False
        if not getattr(self, "spectral", True) or os.environ.get("KGE_HOLE_SPECTRAL", "1") == "0":
            return False
        return lib.kge_hole_spec_width(int(self.e_dim)) // 2 <= 256  # row width the relation-grouped ComplEx kernels take

    def _spec_state(self):
        if self.optimizer is None:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
        if self._spec is None:
            d = int(self.e_dim)
            w = int(lib.kge_hole_spec_width(d))
            WE = torch.empty(d, w, dtype=torch.float32, device=self.device)
            WR = torch.empty(d, w, dtype=torch.float32, device=self.device)
            check(lib.kge_hole_spec_basis(d, WE.data_ptr(), WR.data_ptr(), _lib.stream_ptr()))
            inner = ComplEx(e_dim=w // 2, r_dim=w // 2, num_entities=self.num_entities, num_relations=self.num_relations, dp=0,
                            loss_function=self.loss_function, margin=self.margin, seed=0, name=self.name + '_spectra')
            inner.compile('sgd')  # allocates the spectral gradient accumulators; its optimizer is never applied
            inner._loss = self._loss  # the loss word and the error flag of the step are the outer model's
            inner._err = self._err
            self._spec = dict(WE=WE, WR=WR, inner=inner, w=w)
        sp = self._spec
        inner = sp["inner"]
        inner.loss_function, inner.margin, inner.loss_negative_samples = self.loss_function, self.margin, self.loss_negative_samples
        return sp

    def plan_supported(self):
        return self._spectral_on() and self.optimizer is not None and self._spec_state()["inner"].plan_supported()

    def plan_batch(self, data, out=None, parts=7):
        return self._spec_state()["inner"].plan_batch(data, out=out, parts=parts)

    def plan_batch_counted(self, *a, **kw):
        return self._spec_state()["inner"].plan_batch_counted(*a, **kw)

    def train_grads(self, data, scores_out=None):
        if not self._spectral_on():
            if isinstance(data, BatchPlan):
                raise ValueError("a BatchPlan exists for HolE only on the frequency-domain path (dp == 0)")
            return super().train_grads(data, scores_out)
        sp = self._spec_state()
        inner, d, st = sp["inner"], int(self.e_dim), _lib.stream_ptr()
        self._ensure_train_state(1, 1)  # optimizer slots + the workspace of kge_opt_apply
        check(lib.kge_hole_spec_forward(self.E.data_ptr(), self.num_entities, d, sp["WE"].data_ptr(), inner.E.data_ptr(), st))
        check(lib.kge_hole_spec_forward(self.Rel.data_ptr(), self.num_relations, d, sp["WR"].data_ptr(), inner.Rel.data_ptr(), st))
        inner.train_grads(data, scores_out)
        check(lib.kge_hole_spec_adjoint(inner._gE.data_ptr(), self.num_entities, d, sp["WE"].data_ptr(), self._gE.data_ptr(), 1, st))
        check(lib.kge_hole_spec_adjoint(inner._gR.data_ptr(), self.num_relations, d, sp["WR"].data_ptr(), self._gR.data_ptr(), 1, st))
        return {"loss": self._loss[0]}

    def train_step(self, data, scores_out=None):
        if not self._spectral_on():
            return super().train_step(data, scores_out)
        if isinstance(data, tuple) and len(data) == 2 and (data[1] is None or isinstance(data[0], (list, tuple))):
            data = data[0]
        self.train_grads(data, scores_out)
        self.apply_grads()
        return {"loss": self._loss[0]}


class ConvE(EmbeddingModel):
    """models.py:239-275: x = Drop(ReLU(BN(Dense(Flatten(Drop(ReLU(BN(Conv2D(conv_filters, (conv_size_w, conv_size_h))([s ; p] as a
    2w x h image)))))))), score = <x, o>.  The layer parameters live in one flat vector `theta` (+ the batch norms' moving statistics
    `bn_stats`), see csrc/conve.hip (8 filters of 3x3, the reference defaults, take register-tiled kernels; any other shape the
    general ones)."""
    MODEL = "ConvE"

    def __init__(self, name='ConvE', hidden_dp=0.2, conv_filters=8, conv_size_w=3, conv_size_h=3, **kwargs):
        super().__init__(name=name, **kwargs)
        self.conv_size_w, self.conv_size_h = int(conv_size_w), int(conv_size_h)  # Conv2D kernel (rows, columns), models.py:254
        if self.conv_size_w < 1 or self.conv_size_h < 1 or int(conv_filters) < 1:
            raise ValueError("ConvE: conv_filters and conv_size_* must be positive")
        self.dim = kwargs['e_dim']
        factors = [(int(i), int(self.dim / i)) for i in range(1, int(self.dim ** 0.5) + 1) if self.dim % i == 0]
        self.w, self.h = factors.pop(-1)
        assert self.w > 1 or self.h > 1  # models.py:252
        self.hidden_dp = float(hidden_dp)
        self.conv_filters = int(conv_filters)
        md = self._model_desc(False)
        nt, ns = C.c_int64(0), C.c_int64(0)
        check(lib.kge_conve_sizes(C.byref(md), C.byref(nt), C.byref(ns)))
        Cc, d = self.conv_filters, int(self.dim)
        KK = self.conv_size_w * self.conv_size_h
        Ho, Wo = 2 * self.w - self.conv_size_w + 1, self.h - self.conv_size_h + 1
        F = Ho * Wo * Cc
        self._F = F
        seed = kwargs.get("seed", None)
        gen = None
        if seed is not None:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(seed) + 1)
        theta = torch.zeros(nt.value, dtype=torch.float32, device=self.device)
        # Keras defaults: Glorot-uniform kernels, zero biases, gamma 1, beta 0, moving mean 0 / variance 1
        limk = float(np.sqrt(6.0 / (KK + KK * Cc)))
        limd = float(np.sqrt(6.0 / (F + d)))
        theta[:KK * Cc].uniform_(-limk, limk, generator=gen)
        theta[(KK + 1) * Cc:(KK + 2) * Cc] = 1.0
        off_wd = (KK + 3) * Cc
        theta[off_wd:off_wd + F * d].uniform_(-limd, limd, generator=gen)
        theta[off_wd + F * d + d:off_wd + F * d + 2 * d] = 1.0
        self.theta = theta
        st = torch.zeros(ns.value, dtype=torch.float32, device=self.device)
        st[Cc:2 * Cc] = 1.0
        st[2 * Cc + d:] = 1.0
        self.bn_stats = st
        self._gtheta = torch.zeros_like(theta)
        self._tslots = None
        self._cws = None
        self._iws = None

    # theta <-> named parameters (K, bc, g1, b1, Wd, bd, g2, b2) and stats <-> (mm1, mv1, mm2, mv2)
    def _layout(self):
        Cc, d, F = self.conv_filters, int(self.dim), self._F
        sizes = [("K", self.conv_size_w * self.conv_size_h * Cc), ("bc", Cc), ("g1", Cc), ("b1", Cc), ("Wd", F * d), ("bd", d), ("g2", d), ("b2", d)]
        out, o = {}, 0
        for k, n in sizes:
            out[k] = (o, n)
            o += n
        return out

    def get_head_params(self):
        lay = self._layout()
        Cc, d, F = self.conv_filters, int(self.dim), self._F
        th = self.theta.cpu().numpy()
        st = self.bn_stats.cpu().numpy()
        P = {k: th[o:o + n].copy() for k, (o, n) in lay.items()}
        P["K"] = P["K"].reshape(self.conv_size_w, self.conv_size_h, Cc)
        P["Wd"] = P["Wd"].reshape(F, d)
        P.update(mm1=st[:Cc].copy(), mv1=st[Cc:2 * Cc].copy(), mm2=st[2 * Cc:2 * Cc + d].copy(), mv2=st[2 * Cc + d:].copy())
        return P

    def set_head_params(self, P):
        lay = self._layout()
        th = self.theta.cpu().numpy()
        for k, (o, n) in lay.items():
            if k in P:
                th[o:o + n] = np.asarray(P[k], dtype=np.float32).reshape(-1)
        self.theta.copy_(torch.from_numpy(th))
        if "mm1" in P:
            st = np.concatenate([np.asarray(P[k], dtype=np.float32).reshape(-1) for k in ("mm1", "mv1", "mm2", "mv2")])
            self.bn_stats.copy_(torch.from_numpy(st))

    def _infer_ws(self, n):
        md = self._model_desc(False)
        need = check(lib.kge_conve_infer_workspace_bytes(C.byref(md), n))
        if self._iws is None or self._iws.numel() < need:
            self._iws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return md

    def call(self, inputs, training=False):
        t = as_triples(inputs, self.device, self.num_entities, self.num_relations)
        n = t.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        md = self._infer_ws(n)
        Et = self._eval_table()  # (E itself, or with the literal path the transformed table: models.py:152-158 in inference mode)
        check(lib.kge_conve_score(C.byref(md), Et.data_ptr(), self.Rel.data_ptr(), self.theta.data_ptr(), self.bn_stats.data_ptr(),
                                  t.data_ptr(), n, out.data_ptr(), self._iws.data_ptr(), self._iws.numel(), self._err.data_ptr(),
                                  _lib.stream_ptr()))
        return out

    __call__ = call

    def func(self, s, p, o, training=False):
        """ConvE.func on already-gathered rows (models.py:264-275): the rows are staged as temporary tables ([s ; o], p) and scored by the
        same kernels in inference mode (moving statistics, no dropout)."""
        if training:
            raise NotImplementedError("ConvE.func(training=True): batch statistics and dropout are built inside train_step only")
        s = torch.as_tensor(s, dtype=torch.float32, device=self.device).reshape(-1, int(self.dim))
        p = torch.as_tensor(p, dtype=torch.float32, device=self.device).reshape(-1, int(self.dim)).contiguous()
        o = torch.as_tensor(o, dtype=torch.float32, device=self.device).reshape(-1, int(self.dim))
        n = s.shape[0]
        Et = torch.cat([s, o], 0).contiguous()
        idx = torch.arange(n, dtype=torch.int32, device=self.device)
        t = torch.stack([idx, idx, idx + n], 1).contiguous()
        md = self._infer_ws(n)
        md.num_entities, md.num_relations = 2 * n, n
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        check(lib.kge_conve_score(C.byref(md), Et.data_ptr(), p.data_ptr(), self.theta.data_ptr(), self.bn_stats.data_ptr(), t.data_ptr(), n,
                                  out.data_ptr(), self._iws.data_ptr(), self._iws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        return out

    def hidden(self, triples, Et=None):
        """x(s, p) of every row in inference mode, (n, dim) on the device; score(s, p, o) = <x, E[o]> (Et: the entity table to read, the
        literal-transformed one where the literal path is on)."""
        t = as_triples(triples, self.device, self.num_entities, self.num_relations)
        n = t.shape[0]
        out = torch.empty((n, int(self.dim)), dtype=torch.float32, device=self.device)
        md = self._infer_ws(n)
        if Et is None:
            Et = self._eval_table()
        check(lib.kge_conve_hidden(C.byref(md), Et.data_ptr(), self.Rel.data_ptr(), self.theta.data_ptr(), self.bn_stats.data_ptr(),
                                   t.data_ptr(), n, out.data_ptr(), self._iws.data_ptr(), self._iws.numel(), self._err.data_ptr(),
                                   _lib.stream_ptr()))
        return out

    def train_step(self, data, scores_out=None):
        if isinstance(data, tuple) and len(data) == 2 and (data[1] is None or isinstance(data[0], (list, tuple))):
            data = data[0]
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        if self.optimizer is None:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
        o = self.optimizer
        if o.kind == "adam" and o.slots is None:
            o.slots = tuple(torch.zeros_like(x) for x in (self.E, self.E, self.Rel, self.Rel))
        if o.kind == "adam" and self._tslots is None:
            self._tslots = (torch.zeros_like(self.theta), torch.zeros_like(self.theta))
        md, ld = self._model_desc(True), self._loss_desc()
        if self._lit is not None:
            return self._train_step_literal_conve(pos, neg, md, ld, scores_out)
        need = check(lib.kge_conve_workspace_bytes(C.byref(md), pos.shape[0], neg.shape[0]))
        if self._cws is None or self._cws.numel() < need:
            self._cws = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        o.iterations += 1
        self._seed += 1
        check(lib.kge_conve_train_grads(C.byref(md), C.byref(ld), self.E.data_ptr(), self.Rel.data_ptr(), self.theta.data_ptr(),
                                        self.bn_stats.data_ptr(), self._gE.data_ptr(), self._gR.data_ptr(), self._gtheta.data_ptr(),
                                        pos.data_ptr(), pos.shape[0], neg.data_ptr(), neg.shape[0], self._loss.data_ptr(),
                                        scores_out.data_ptr() if scores_out is not None else None, self._seed, self._cws.data_ptr(),
                                        self._cws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        od = self._opt_desc()
        if self._ws is None:
            need2 = check(lib.kge_train_step_workspace_bytes(C.byref(md), C.byref(ld), 0, 0))
            self._ws = torch.zeros(int(need2), dtype=torch.uint8, device=self.device)
        check(lib.kge_opt_apply(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                self._gR.data_ptr(), 1.0, self._loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()))
        ts = [x.data_ptr() for x in self._tslots] if self._tslots else [None, None]
        odv = OptDesc(OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, o.iterations, ts[0], ts[1], None, None)
        check(lib.kge_opt_apply_vector(C.byref(odv), self.theta.data_ptr(), self._gtheta.data_ptr(), self.theta.numel(), 1.0, _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def _train_step_literal_conve(self, pos, neg, md, ld, scores_out):
        """models.py:152-158 in front of ConvE.func: the gradient half through kge_lit_conve_train_grads (the head on the batch's transformed
        rows as staged tables), then the optimizer on the four parameter groups (tables, head, Dense)."""
        lt, o = self._lit, self.optimizer
        if o.kind == "adam" and lt["slots"] is None:
            lt["slots"] = (torch.zeros_like(lt["theta"]), torch.zeros_like(lt["theta"]))
        need = check(lib.kge_lit_workspace_bytes(C.byref(md), lt["L"], pos.shape[0], neg.shape[0]))
        if lt["ws"] is None or lt["ws"].numel() < need:
            lt["ws"] = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        o.iterations += 1
        self._seed += 1
        check(lib.kge_lit_conve_train_grads(C.byref(md), C.byref(ld), lt["act"], self.E.data_ptr(), self.Rel.data_ptr(), lt["table"].data_ptr(), lt["L"],
                                            lt["table"].shape[0], lt["theta"].data_ptr(), self.theta.data_ptr(), self.bn_stats.data_ptr(), self._gE.data_ptr(),
                                            self._gR.data_ptr(), lt["gtheta"].data_ptr(), self._gtheta.data_ptr(), pos.data_ptr(), pos.shape[0], neg.data_ptr(),
                                            neg.shape[0], self._loss.data_ptr(), scores_out.data_ptr() if scores_out is not None else None, self._seed,
                                            lt["ws"].data_ptr(), lt["ws"].numel(), self._err.data_ptr(), _lib.stream_ptr()))
        od = self._opt_desc()
        if self._ws is None:
            need2 = check(lib.kge_train_step_workspace_bytes(C.byref(md), C.byref(ld), 0, 0))
            self._ws = torch.zeros(int(need2), dtype=torch.uint8, device=self.device)
        check(lib.kge_opt_apply(C.byref(md), C.byref(ld), C.byref(od), self.E.data_ptr(), self.Rel.data_ptr(), self._gE.data_ptr(),
                                self._gR.data_ptr(), 1.0, self._loss.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr()))
        for th, g, sl in ((self.theta, self._gtheta, self._tslots), (lt["theta"], lt["gtheta"], lt["slots"])):
            ts = [x.data_ptr() for x in sl] if sl else [None, None]
            odv = OptDesc(OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, o.iterations, ts[0], ts[1], None, None)
            check(lib.kge_opt_apply_vector(C.byref(odv), th.data_ptr(), g.data_ptr(), th.numel(), 1.0, _lib.stream_ptr()))
        return {"loss": self._loss[0]}

    def _batch_loss_into(self, data, ctx, acc=None, out=None):
        """Forward-only loss of one batch in inference mode (models.py:239-275 with moving statistics and no dropout; tests.py:130-134
        passes validation_data to fit): the scores of both arrays by kge_conve_score, the loss on them by kge_loss_value."""
        pos = as_triples(data[0], self.device, self.num_entities, self.num_relations)
        neg = as_triples(data[1], self.device, self.num_entities, self.num_relations)
        sc = torch.empty(pos.shape[0] + neg.shape[0], dtype=torch.float32, device=self.device)
        Et = ctx["Et"] if ctx and "Et" in ctx else self._eval_table()
        for t, o_ in ((pos, sc[:pos.shape[0]]), (neg, sc[pos.shape[0]:])):
            md = self._infer_ws(t.shape[0])
            check(lib.kge_conve_score(C.byref(md), Et.data_ptr(), self.Rel.data_ptr(), self.theta.data_ptr(), self.bn_stats.data_ptr(), t.data_ptr(),
                                      t.shape[0], o_.data_ptr(), self._iws.data_ptr(), self._iws.numel(), self._err.data_ptr(), _lib.stream_ptr()))
        ld = self._loss_desc()
        ws = self._loss_ws(pos.shape[0], neg.shape[0])
        if out is None:
            out = getattr(self, "_vloss", None)
            if out is None:
                out = self._vloss = torch.zeros(1, dtype=torch.float32, device=self.device)
        check(lib.kge_loss_value(C.byref(ld), sc.data_ptr(), pos.shape[0], sc[pos.shape[0]:].data_ptr(), neg.shape[0], out.data_ptr(), None, None,
                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        if acc is not None:
            acc += out.double()
        return out

    def rank_triples(self, test, num_entities, filt, chunk=8192):
        """Filtered ranks (utils.py:199-264).  Tails: queries x(s,p) against the entity table.  Heads: x depends on the
        candidate subject, so per relation the hidden vectors of ALL (s', p) are built once and swept against E[o]."""
        from .utils import filter_coo
        test = np.ascontiguousarray(np.asarray(test, dtype=np.int32).reshape(-1, 3))
        N, d = int(num_entities), int(self.dim)
        tails = np.empty(len(test), dtype=np.int64)
        heads = np.empty(len(test), dtype=np.int64)

        def sweep(Q, cand, targets, fq, fi):
            nq = Q.shape[0]
            need = check(lib.kge_rank_sweep_q_workspace_bytes(nq, d))
            ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            cnt = torch.empty(nq, dtype=torch.int32, device=self.device)
            tg = torch.from_numpy(np.ascontiguousarray(targets, dtype=np.int32)).to(self.device)
            fqd = torch.from_numpy(fq).to(self.device)
            fid = torch.from_numpy(fi).to(self.device)
            check(lib.kge_rank_sweep_q(Q.data_ptr(), nq, d, cand.data_ptr(), 0, N, tg.data_ptr(), fqd.data_ptr() if len(fq) else None,
                                       fid.data_ptr() if len(fi) else None, len(fq), cnt.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                                       _lib.stream_ptr()))
            return 1 + cnt.cpu().numpy().astype(np.int64)

        Et = self._eval_table()  # (with the literal path: the transformed table, built once for the whole evaluation)
        Ecand = Et[:N].contiguous()
        for a in range(0, len(test), chunk):
            part = test[a:a + chunk]
            fq, fi = filter_coo(part, 0, filt, N)
            tails[a:a + chunk] = sweep(self.hidden(part, Et), Ecand, part[:, 2], fq, fi)
        ent = np.arange(N, dtype=np.int32)
        for p in np.unique(test[:, 1]):
            idx = np.nonzero(test[:, 1] == p)[0]
            allsp = np.stack([ent, np.full(N, p, dtype=np.int32), np.zeros(N, dtype=np.int32)], 1)
            Xp = self.hidden(allsp, Et)  # (N, d): candidate matrix of this relation
            for a in range(0, len(idx), chunk):
                sel = idx[a:a + chunk]
                part = test[sel]
                fq, fi = filter_coo(part, 1, filt, N)
                Q = Et.index_select(0, torch.from_numpy(part[:, 2].astype(np.int64)).to(self.device)).contiguous()
                heads[sel] = sweep(Q, Xp, part[:, 0], fq, fi)
        self._raise_on_flag()
        return tails, heads

    def rank(self, test_data, num_entities, filter_t, filter_h, chunk=8192):
        filt = [(s, p, o) for (s, p), os_ in filter_t.items() for o in os_]
        return self.rank_triples(np.asarray([tuple(int(v) for v in t) for t in test_data], dtype=np.int32), num_entities,
                                 np.asarray(filt, dtype=np.int32).reshape(-1, 3))

    def _extra_state(self):
        st = super()._extra_state()
        st["theta"], st["bn_stats"] = self.theta, self.bn_stats
        if self._tslots is not None:
            st["theta_slot_m"], st["theta_slot_v"] = self._tslots
        return st

    def _set_extra_state(self, st):
        super()._set_extra_state(st)
        self.theta.copy_(st["theta"])
        self.bn_stats.copy_(st["bn_stats"])
        if "theta_slot_m" in st:
            self._tslots = (st["theta_slot_m"].to(self.device).clone(), st["theta_slot_v"].to(self.device).clone())

    def get_weights(self):
        return super().get_weights() + [self.theta.cpu().numpy(), self.bn_stats.cpu().numpy()]

    def set_weights(self, ws):
        if len(ws) != 4:
            raise ValueError(f"You called `set_weights(weights)` on {self.name} with a weight list of length {len(ws)}, but it expects 4 weights")
        th, bn = np.asarray(ws[2], dtype=np.float32).reshape(-1), np.asarray(ws[3], dtype=np.float32).reshape(-1)
        if th.shape[0] != self.theta.numel() or bn.shape[0] != self.bn_stats.numel():
            raise ValueError(f"ConvE head weights of sizes {th.shape[0]}, {bn.shape[0]}; expected {self.theta.numel()}, {self.bn_stats.numel()}")
        EmbeddingModel.set_weights(self, ws[:2])
        self.theta.copy_(torch.from_numpy(th))
        self.bn_stats.copy_(torch.from_numpy(bn))
