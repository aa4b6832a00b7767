"""The training hot path as a three-stage software pipeline on the GPU, replayed from hipGraphs.

What Keras does for the reference (tests.py:130-134: `fit` over a `Sequence`, batches prefetched by worker threads while the
previous step runs) becomes, per tick t:

    stage S (stream 1)  draw batch t+2 on the device: bit-exact NumPy / CPython MT19937 streams (utils.py:114-147)
    stage P (stream 2)  plan of batch t+1: fold duplicate rows, per-relation / per-entity gradient lists (ids only)
    stage T (main)      train step t on the plan of batch t: forward, loss, gradients, optimizer (tests.py:63-89)

The three chains are independent inside a tick (double-buffered batches and plans) and meet at its end.  A tick is ~35 short
kernels, so launching it kernel by kernel is bound by the host; the steady-state tick is therefore captured once per buffer
parity into a hipGraph and replayed with one launch.  Everything that changes from step to step lives on the device: the
generator states, the batch cursor, the Adam step counter (kge_opt_desc.step_dev).  The last two ticks of an epoch draw no
further batches (the random streams must stop where the reference's would), so they are launched stage by stage.
"""
import torch

from . import _lib


class TrainPipeline:
    def __init__(self, model, kg, n_rows, batch_size, ns, num_e, sampler, graphs=True, between=None):
        """kg: (n,3) int32 device rows; n_rows: length of the epoch permutation; sampler: DeviceSampler whose states are loaded.
        `between`: optional callable run between the gradient half and the optimizer half of every step (the data-parallel
        all-reduce); the optimizer half then stays outside the graphs."""
        if not model.plan_supported():
            raise NotImplementedError("TrainPipeline needs a model / loss with batch plans (EmbeddingModel.plan_supported)")
        self.model, self.kg, self.B, self.ns, self.num_e, self.sampler = model, kg, int(batch_size), int(ns), int(num_e), sampler
        self.dev = kg.device
        self.between = between
        # a captured step cannot change its dropout seed: graphs only without dropout
        self.use_graphs = bool(graphs) and float(getattr(model, "dp", 0.0) or 0.0) == 0.0
        m = self.B * self.ns
        self.bufs = [tuple(torch.empty((m, 3), dtype=torch.int32, device=self.dev) for _ in range(2)) for _ in range(2)]
        md, ld = model._model_desc(True), model._loss_desc()
        need = _lib.check(_lib.lib.kge_batch_plan_bytes(_lib.C.byref(md), _lib.C.byref(ld), m, m))
        from .models import BatchPlan
        self.plans = [BatchPlan(torch.empty(int(need), dtype=torch.uint8, device=self.dev), m, m) for _ in range(2)]
        self.index = torch.zeros(int(n_rows), dtype=torch.int64, device=self.dev)   # epoch permutation (tests.py:50-53)
        self.starts = None                                                          # batch numbers in drawing order
        self.cursor = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.ar = torch.arange(self.B, dtype=torch.int64, device=self.dev)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=self.dev)
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=self.dev)          # kge_opt_desc.step_dev: {t, lr_t, sum of losses}
        self.s1, self.s2 = torch.cuda.Stream(), torch.cuda.Stream()
        self.graphs = [None, None]
        self._sig = None

    # ------------------------------------------------------------------ stages (each on the current stream)
    def _stage_S(self, slot):
        b = self.starts.index_select(0, self.cursor)
        rows = self.index.index_select(0, b * self.B + self.ar)
        batch = self.kg.index_select(0, rows)
        self.sampler.build_batch(batch, self.ns, self.num_e, *self.bufs[slot])
        self.cursor.add_(1)

    def _stage_P(self, slot):
        self.plans[slot] = self.model.plan_batch(list(self.bufs[slot]), out=self.plans[slot])

    def _stage_T(self, slot):
        if self.between is None:
            self.model.train_step(self.plans[slot])  # the step's last kernel adds its loss to the device step state
        else:
            self.model.train_grads(self.plans[slot])

    def _tick(self, j, do_S, do_P):
        cur = torch.cuda.current_stream()
        if do_S:
            self.s1.wait_stream(cur)
            with torch.cuda.stream(self.s1):
                self._stage_S(j)
        if do_P:
            self.s2.wait_stream(cur)
            with torch.cuda.stream(self.s2):
                self._stage_P(1 - j)
        self._stage_T(j)
        if do_S:
            cur.wait_stream(self.s1)
        if do_P:
            cur.wait_stream(self.s2)

    def _after(self):
        if self.between is not None:
            scale = self.between()  # e.g. the data-parallel all-reduce; returns the gradient scale (1 / world size)
            self.model.apply_grads(1.0 if scale is None else float(scale))
            self.loss_sum += self.model._loss[0].double()

    # ------------------------------------------------------------------ graphs
    def _signature(self):
        mo, sa = self.model, self.sampler
        t = [mo.E, mo.Rel, mo._gall, mo._ws, mo._err, sa.np_state, sa.py_state, sa._ws, self.plans[0].buf, self.plans[1].buf, self.starts]
        if mo.optimizer.slots:
            t += list(mo.optimizer.slots)
        return tuple(x.data_ptr() for x in t if x is not None) + (mo.optimizer.kind, mo.optimizer.lr, mo.loss_function, float(mo.margin or 1))

    def _capture(self):
        """Capture the steady-state tick for both buffer parities.  Capturing enqueues nothing; host-side counters the stages
        bump (dropout seed, iterations) are restored."""
        mo = self.model
        seed, it = mo._seed, mo.optimizer.iterations
        for j in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._tick(j, True, True)
            self.graphs[j] = g
        mo._seed, mo.optimizer.iterations = seed, it
        self._sig = self._signature()

    # ------------------------------------------------------------------ driving
    def load_epoch(self, index, order):
        """index: the epoch's row permutation (device or host); order: batch numbers in the order Keras would draw them."""
        self.index.copy_(torch.as_tensor(index, dtype=torch.int64))
        order = torch.as_tensor(list(order), dtype=torch.int64)
        self.n = int(order.numel())
        if self.starts is None or self.starts.numel() < self.n:  # persistent (the graphs hold its address)
            self.starts = torch.zeros(max(self.n, 16), dtype=torch.int64, device=self.dev)
        self.starts[:self.n].copy_(order)
        self.cursor.zero_()
        self.loss_sum.zero_()
        self.t = 0

    def start(self):
        """Prologue: batch 0 drawn and planned, batch 1 drawn.  Switches the optimizer to the device step counter."""
        mo = self.model
        mo._ensure_train_state(self.B * self.ns, self.B * self.ns)
        if self.between is None:
            mo._step_dev = self.step_dev
            o = mo.optimizer
            od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, int(o.iterations) + 1, None, None, None, None,
                              self.step_dev.data_ptr())
            _lib.check(_lib.lib.kge_opt_state_init(_lib.C.byref(od), _lib.stream_ptr()))
        if self.n >= 1:
            self._stage_S(0)
            self._stage_P(0)
        if self.n >= 2:
            self._stage_S(1)

    def tick(self):
        """One training step (and the stages of the next two batches that run beside it)."""
        t, j = self.t, self.t % 2
        do_P, do_S = t + 1 < self.n, t + 2 < self.n
        if self.use_graphs and do_S and do_P:
            if self.graphs[j] is None or self._sig != self._signature():
                self._capture()
            self.graphs[j].replay()
        else:
            self._tick(j, do_S, do_P)
        self._after()
        self.t += 1

    def finish(self):
        mo = self.model
        if getattr(mo, "_step_dev", None) is not None:
            mo.optimizer.iterations = int(self.step_dev[0].item())
            mo._step_dev = None
            return self.step_dev.view(torch.float32)[3].double()  # running sum kept by the library (kge_opt_desc.step_dev)
        return self.loss_sum

    def run_epoch(self, index, order):
        """All batches of one epoch; returns the summed loss (0-d float64 device tensor)."""
        self.load_epoch(index, order)
        self.start()
        for _ in range(self.n):
            self.tick()
        return self.finish()
