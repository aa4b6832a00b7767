"""The training hot path as a software pipeline on the GPU: generator rounds | draw | plan | step, one to two batches apart.

What Keras does for the reference (tests.py:130-134: `fit` over a `Sequence`, batches prefetched by worker threads while the
previous step runs) becomes streams that run next to the training stream:

    rounds G            the word streams of NumPy's / CPython's MT19937 produced ahead into rings (native driver only)
    stage S (stream 1)  draw batch k: bit-exact NumPy / CPython draws on the device (utils.py:114-147)
    stage P (stream 2)  plan of batch k: fold duplicate rows, per-relation / per-entity gradient lists (ids only)
    stage T (main)      train step k on the plan of batch k: forward, loss, gradients, optimizer (tests.py:63-89)

with double-buffered batches and plans and events per buffer pair (drawn / arrays free / planned / plan free).  Everything that
changes from step to step lives on the device: the generator states, the batch cursor, the optimizer step state
(kge_opt_desc.step_dev).

Without dropout and without a hook between the halves of a step the whole loop is native (csrc/pipeline.cpp): kge_pipeline_run(n)
issues n steps from C, kernel by kernel on four streams -- one ctypes call per `run`, no per-step Python; the step runs between twin
tables (kge_train_step_swap: two launches), the draw and plan streams are confined to half of the compute units, and the loop is
issued from a stream of this object's own (see _on_native).  With an RCCL communicator (`comm`) the native loop runs the
data-parallel step instead: gradient half, all-reduce of the flat gradient buffer on the communicator (the entity table's share beside
the relation half), optimizer half.  The Python loop below remains for a `between` hook (data-parallel training over
torch.distributed: the all-reduce sits between the halves of the step), for `serial` profiling runs and for epochs taken over from
the native driver; it replays S and P from hipGraphs captured once per buffer parity where it can (one graph holding all stages as
branches does not overlap them on ROCm 7.2; separate graphs on separate streams do).  (Dropout is no obstacle to the native loop: it
launches kernel by kernel and hands every step its mask seed.)
"""
import os
import time

import torch

from . import _lib


class TrainPipeline:
    def __init__(self, model, kg, n_rows, batch_size, ns, num_e, sampler, graphs=True, between=None, between_halves=None, comm=None, world=1,
                 row_shard=None):
        """kg: (n,3) int32 device rows; n_rows: length of the epoch permutation; sampler: DeviceSampler whose states are loaded.
        `between`: optional callable run between the gradient half and the optimizer half of every step (the data-parallel
        all-reduce); the optimizer half then stays outside the graphs.
        `comm`, `world`: an RCCL communicator (ncclComm_t as an integer / c_void_p: dist.make_rccl_comm) over `world` ranks -- the
        data-parallel step then runs inside the native driver (gradient half, all-reduce of the flat gradient buffer with the entity
        table's share beside the relation half, optimizer half with scale 1 / world); excludes `between`."""
        if comm is not None and between is not None:
            raise ValueError("TrainPipeline: either an RCCL communicator (native data-parallel steps) or a `between` hook, not both")
        # row_shard: a dist.RowShardTrainer -- owner-computes steps over row-sharded tables: every rank feeds this pipeline the SAME
        # batches (same sampler seeds), the step stage is the trainer's windowed step + its exchange (collectives: never captured)
        if row_shard is not None and (comm is not None or between is not None):
            raise ValueError("TrainPipeline: row_shard excludes comm / between (those are the data-parallel replica step)")
        self.row_shard = row_shard
        self.comm, self.world = comm, int(world)
        if not model.pipeline_supported():
            # (HolE's frequency-domain step takes plans too, but is a composite of library calls, not one captured step)
            raise NotImplementedError("TrainPipeline needs a model / loss with batch plans (EmbeddingModel.plan_supported)")
        self.model, self.kg, self.B, self.ns, self.num_e, self.sampler = model, kg, int(batch_size), int(ns), int(num_e), sampler
        self.dev = kg.device
        self.between = between
        # optional callable run between the entity half and the relation half of the gradient computation (data-parallel training:
        # start the exchange of the entity gradient table there); only with `between` and where the model's halves come apart
        self.between_halves = between_halves if between is not None else None
        # a captured step cannot change its dropout seed: graphs only without dropout.  With a collective between the halves of
        # the step (one process per GPU) the draw and plan stages are still graph replays; the step itself (gradient half,
        # collective, optimizer half) is launched kernel by kernel, so that no collective ever sits inside a captured graph
        self.use_graphs = bool(graphs) and float(getattr(model, "dp", 0.0) or 0.0) == 0.0
        self.native_wanted = bool(graphs)  # the native driver launches kernel by kernel: dropout (a seed per step) is no obstacle
        # (a collective never sits inside a captured graph: with a `between` hook OR an RCCL communicator the step stage -- gradient half,
        #  all-reduce, optimizer half -- is launched kernel by kernel; only the draw and plan stages are replayed)
        self.graph_stages = ("S", "P", "T") if between is None and comm is None and row_shard is None else ("S", "P")
        m = self.B * self.ns
        # the batch before oversampling + how often each row was picked: the sampler's by-product the plans are built from when
        # there is no dropout (with dropout every oversampled row has its own mask and the plan takes the arrays as they are)
        self.counted = float(getattr(model, "dp", 0.0) or 0.0) == 0.0
        self.bufs = [None, None] if self.counted else [tuple(torch.empty((m, 3), dtype=torch.int32, device=self.dev) for _ in range(2))
                                                        for _ in range(2)]
        self.cand = [(torch.empty((self.B + m, 3), dtype=torch.int32, device=self.dev), torch.zeros(_lib.CAND_POS_LANES * self.B + m, dtype=torch.int32, device=self.dev))
                     for _ in range(2)] if self.counted else [None, None]
        md, ld = model._model_desc(True), model._loss_desc()
        need = _lib.check(_lib.lib.kge_batch_plan_bytes(_lib.C.byref(md), _lib.C.byref(ld), m, m))
        from .models import BatchPlan
        # plans[0], plans[1]: one per buffer parity; plans[2]: the native driver's third buffer (kge_pipeline_desc.plan3), which the
        # Python loop also uses when it takes an epoch over from the driver (batch k sits in plans[k % self._np])
        # (four to six buffers, with "plan consumed" recorded behind every (buffers - 2)-th step only, measured slower at the bench shape:
        # 0.0999 ms per step with 3, 0.1005 with 4, 0.102 with 5, 0.104 with 6 -- profiles/r03_experiments.txt -- so three it is)
        n_plans = 4  # (the step records "plan consumed" every second step then: pipeline.cpp rec_every; 3 -> 4 measured 0.1017 -> 0.0987 ms per step with chained steps)
        self.plans = [BatchPlan(torch.zeros(int(need), dtype=torch.uint8, device=self.dev), m, m) for _ in range(n_plans)]
        self._np = 2
        self.index = torch.zeros(int(n_rows), dtype=torch.int64, device=self.dev)   # epoch permutation (tests.py:50-53)
        self.starts = None                                                          # batch numbers in drawing order
        self.cursor = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=self.dev)
        self.step_dev = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device=self.dev)  # kge_opt_desc.step_dev: {t, lr_t, sum of losses}
        self.s1, self.s2, self.s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        self.ready = [torch.cuda.Event() for _ in range(2)]     # batch arrays drawn
        self.freed = [torch.cuda.Event() for _ in range(2)]     # batch arrays no longer needed (their plan is built)
        self.planned = [torch.cuda.Event() for _ in range(len(self.plans))]   # plan built (per plan buffer)
        self.pfreed = [torch.cuda.Event() for _ in range(len(self.plans))]    # plan consumed
        self.graphs = {}
        self._done = {}  # (stage, slot) -> event recorded behind the last replay of that graph
        self._sig = None
        self.serial = False  # True: every stage on the caller's stream, nothing overlaps (clean per-kernel timings)
        self._h = None          # native driver (kge_pipeline_create) and the signature it was captured for
        self._h_sig = None
        self._native_active = False
        self._twins = None
        self._nstream = None    # the stream the native driver's steps run on (see _on_native)
        self._halves = None     # cached model.grad_halves_supported()
        self._taken_over = False  # this epoch was begun by the native driver and is finished by the Python loop (kernel by kernel)

    # ------------------------------------------------------------------ stages (each on the current stream)
    def _stage_S(self, slot):
        # positives read in place (kg[index[starts[cursor] * B + i]]); the sampler's last kernel moves the cursor on
        # (with the folded form the oversampled arrays themselves are not needed and not written)
        pos, neg = (False, None) if self.counted else self.bufs[slot]
        self.sampler.build_batch(None, self.ns, self.num_e, pos, neg, indexed=(self.kg, self.index, self.starts, self.cursor, self.B),
                                 counted=self.cand[slot])

    def _stage_P(self, slot, fork=None, pslot=None):
        """rows first, then the entity lists and the relation lists -- side by side on `fork` (a stream that already follows
        the caller's origin stream and is joined by the caller) when given: they only share the rows.  slot: candidate / array
        buffer; pslot: plan buffer (default: the same number)."""
        cur, pl = torch.cuda.current_stream(), self.plans[slot if pslot is None else pslot]
        b = None if self.counted else list(self.bufs[slot])
        m = self.B * self.ns
        if self.counted:
            build = lambda parts: self.model.plan_batch_counted(*self.cand[slot], self.B, m, m, out=pl, parts=parts, pos_lanes=_lib.CAND_POS_LANES)  # noqa: E731
        else:
            build = lambda parts: self.model.plan_batch(b, out=pl, parts=parts)  # noqa: E731
        if fork is None:
            build(7 | 8)  # (8: the plan buffers' counters are zero: zero-filled in start(), left zero by every step that consumes a plan)
            return
        build(1)
        fork.wait_stream(cur)
        with torch.cuda.stream(fork):
            build(2)
        build(4)

    def _stage_T(self, slot):
        if self.row_shard is not None:
            self.row_shard.step(self.plans[slot])
        elif self.comm is not None:  # (the Python loop's form of the native data-parallel step: serial / profiling runs)
            self.model.train_grads(self.plans[slot])
            g = self.model._gall
            _lib.check(_lib.lib.kge_allreduce_f32(self.comm if isinstance(self.comm, int) else self.comm.value, g.data_ptr(), g.numel(), _lib.stream_ptr()))
            self.model.apply_grads(1.0 / self.world)
        elif self.between is None:
            self.model.train_step(self.plans[slot])  # the step's last kernel adds its loss to the device step state
        elif self.between_halves is not None and self._halves_ok():
            self.model.train_grads(self.plans[slot], parts=1)
            self.between_halves()
            self.model.train_grads(self.plans[slot], parts=2)
        else:
            self.model.train_grads(self.plans[slot])

    def _halves_ok(self):
        if self._halves is None:
            self._halves = bool(self.model.grad_halves_supported())
        return self._halves

    def _after(self):
        if self.between is not None:
            scale = self.between()  # e.g. the data-parallel all-reduce; returns the gradient scale (1 / world size)
            self.model.apply_grads(1.0 if scale is None else float(scale))
            self.loss_sum += self.model._loss[0].double()

    # ------------------------------------------------------------------ graphs
    def _signature(self):
        mo, sa = self.model, self.sampler
        t = [mo.E, mo.Rel, mo._gall, mo._ws, mo._err, sa.np_state, sa.py_state, sa._ws] + [pl.buf for pl in self.plans] + [self.starts,
             getattr(sa, "_stream", None)]
        t += [x for c in self.cand if c is not None for x in c]
        t += [x for b in self.bufs if b is not None for x in b]
        if self._twins is not None:
            t += list(self._twins)
        if mo.optimizer.slots:
            t += list(mo.optimizer.slots)
        return tuple(x.data_ptr() for x in t if x is not None) + (mo.optimizer.kind, mo.optimizer.lr, mo.loss_function, float(mo.margin or 1),
                                                                  None if self.comm is None else (self.comm if isinstance(self.comm, int) else self.comm.value), self.world)

    def _capture(self):
        """Capture the three stages for both buffer parities.  Capturing enqueues nothing; host-side counters the stages bump
        (dropout seed, iterations) are restored."""
        mo = self.model
        seed, it = mo._seed, mo.optimizer.iterations
        self.graphs = {}
        try:
            for j in (0, 1):
                for name, fn in (("S", lambda j=j: self._stage_S(j)), ("P", lambda j=j: self._stage_P(j)), ("T", lambda j=j: self._stage_T(j))):
                    if name not in self.graph_stages:
                        continue
                    g = torch.cuda.CUDAGraph()
                    # thread_local: other threads of the process (e.g. the watchdog of an RCCL process group) may keep calling
                    # into the runtime while this thread captures
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        fn()
                    self.graphs[name, j] = g
        except Exception as e:  # a runtime that cannot capture these stages: launch them kernel by kernel instead
            import warnings
            warnings.warn(f"TrainPipeline: hipGraph capture failed ({e!r}); falling back to stream launches")
            self.graphs, self.use_graphs = {}, False
            torch.cuda.synchronize()
        mo._seed, mo.optimizer.iterations = seed, it
        self._sig = self._signature()

    def _stage_P_forked(self, slot, pslot=None):
        """Stage P with its two list builds side by side on a forked stream -- kernel-by-kernel launches only, never inside a
        capture.  (Round 1 saw the ROCm 7.2 runtime crash once with this fork inside a captured stage; the failing call and its
        error were not kept, and the two notes written then disagree on whether it died at hipStreamEndCapture or at the first
        replay, so the cause is NOT established.  What is established: every capture this package makes is flat -- one stream, no
        fork -- and this method refuses to run under capture.)"""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("TrainPipeline._stage_P_forked must not be captured into a hipGraph (see its docstring)")
        cur = torch.cuda.current_stream()
        self.s3.wait_stream(cur)
        self._stage_P(slot, fork=self.s3, pslot=pslot)
        cur.wait_stream(self.s3)

    def _run(self, name, slot, pslot=None):
        """slot: candidate / array buffer of the batch (stage S, P) or its plan buffer (stage T); pslot: plan buffer of stage P.  The
        captured stages exist per buffer parity (plan buffer == candidate buffer); an epoch taken over from the native driver, which
        plans into three buffers, is launched kernel by kernel."""
        graphs = self.use_graphs and not self._taken_over
        if graphs and not self.graphs:  # (start() re-captures when a tensor the graphs hold has moved)
            self._capture()  # may switch use_graphs off (capture not possible on this runtime)
            graphs = self.use_graphs
        if graphs and name in self.graph_stages:
            # A graph exec cannot run twice at once: launching it again while its previous launch is still in flight blocks
            # inside the runtime, and that wait sleeps (milliseconds on a busy host).  Spin on an event instead.
            ev = self._done.get((name, slot))
            if ev is not None:
                t0 = None
                while not ev.query():  # bounded: a stage that never finishes (a fault on the device) becomes an error, not a hang
                    if t0 is None:
                        t0 = time.monotonic()
                    elif time.monotonic() - t0 > 20.0:
                        raise RuntimeError(f"TrainPipeline: stage {name} of slot {slot} did not finish within 20 s")
            else:
                ev = self._done[name, slot] = torch.cuda.Event()
            self.graphs[name, slot].replay()
            ev.record(torch.cuda.current_stream())
        elif name == "S":
            self._stage_S(slot)
        elif name == "P":
            # kernel-by-kernel launches are bound by the host: the forked variant (shorter chain, more launches and event
            # traffic) only pays where nothing else sits on the step's critical path
            self._stage_P(slot, pslot=pslot) if (self.serial or self.between is not None) else self._stage_P_forked(slot, pslot)
        else:
            self._stage_T(slot)

    # ------------------------------------------------------------------ native driver (csrc/pipeline.cpp)
    def native_ok(self):
        return self.native_wanted and self.between is None and self.row_shard is None and not self.serial

    def _on_native(self, fn, join=True):
        """Run fn with the native driver's own stream current, ordered after the caller's stream and (join) joined back into it.  The
        driver confines its producer streams to half of the compute units (csrc/pipeline.cpp); such CU-masked streams are blocking
        streams, which synchronise with the legacy default stream -- PyTorch's default -- so the steps must not be issued from that
        one.  join=False leaves the caller's stream alone: a stream-to-stream wait that stays pending behind hundreds of enqueued
        steps slows the whole pipeline down on ROCm 7.2 (measured: 0.104 -> 0.17-0.21 ms per step), so long runs end with a host
        synchronisation instead (finish(), torch.cuda.synchronize())."""
        cur = torch.cuda.current_stream()
        if self._nstream is None:
            self._nstream = torch.cuda.Stream(device=self.dev)
        self._nstream.wait_stream(cur)
        with torch.cuda.stream(self._nstream):
            out = fn()
        if join:
            cur.wait_stream(self._nstream)
        return out

    def _ensure_native(self):
        mo, sa, C = self.model, self.sampler, _lib.C
        sig = self._signature()
        if self._h is not None and self._h_sig == sig:
            return
        self.close()
        o = mo.optimizer
        sl = [x.data_ptr() for x in o.slots] if o.slots else [None] * 4
        m = self.B * self.ns
        d = _lib.PipelineDesc()
        d.md, d.ld = mo._model_desc(True), mo._loss_desc()
        d.od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, 1, *sl, self.step_dev.data_ptr())
        d.E, d.Rel, d.gradE, d.gradRel, d.loss_out = mo.E.data_ptr(), mo.Rel.data_ptr(), mo._gE.data_ptr(), mo._gR.data_ptr(), mo._loss.data_ptr()
        d.train_ws, d.train_ws_bytes, d.err_flag = mo._ws.data_ptr(), mo._ws.numel(), mo._err.data_ptr()
        d.np_state, d.py_state = sa.np_state.data_ptr(), sa.py_state.data_ptr()
        d.kg, d.index, d.starts, d.cursor = self.kg.data_ptr(), self.index.data_ptr(), self.starts.data_ptr(), self.cursor.data_ptr()
        d.B, d.ns, d.num_e = self.B, self.ns, self.num_e
        plan = sa._plans.get((self.B, self.ns, self.num_e)) if sa.segmented else None
        d.sampler_plan = plan.data_ptr() if plan is not None else None
        d.sampler_ws, d.sampler_ws_bytes, d.sampler_err = sa._ws.data_ptr(), sa._ws.numel(), sa.err.data_ptr()
        ring = sa.stream_buffer(self.B, self.ns, self.num_e)  # word rings of the two generators, filled one batch ahead
        d.sampler_stream, d.sampler_stream_bytes = ring.data_ptr(), ring.numel()
        splan = sa.stream_plan(self.B, self.ns, self.num_e)
        d.sampler_stream_plan = splan.data_ptr() if splan is not None else None
        for j in (0, 1):
            if self.counted:
                d.cand_rows[j], d.cand_counts[j] = self.cand[j][0].data_ptr(), self.cand[j][1].data_ptr()
            else:  # dropout: the oversampled arrays themselves (a mask per row: no folded form)
                d.pos[j], d.neg[j] = self.bufs[j][0].data_ptr(), self.bufs[j][1].data_ptr()
            d.plan[j] = self.plans[j].buf.data_ptr()
        d.plan_bytes = min(pl.buf.numel() for pl in self.plans)
        d.plan3 = self.plans[2].buf.data_ptr()  # third plan buffer of the native driver, and the further ones
        for i, pl in enumerate(self.plans[3:6]):
            d.plan_more[i] = pl.buf.data_ptr()
        # twin tables: steps alternate between (E, Rel) and these (kge_train_step_swap); run() leaves the current ones in (E, Rel)
        if self._twins is not None:
            d.E_twin, d.Rel_twin = self._twins[0].data_ptr(), self._twins[1].data_ptr()
        if self.comm is not None:  # native data-parallel steps: one flat buffer [gradE | gradRel | pad | loss word] to all-reduce
            d.comm = self.comm if isinstance(self.comm, int) else self.comm.value
            d.world, d.grad_flat, d.grad_flat_n = self.world, mo._gall.data_ptr(), mo._gall.numel()
        h = C.c_void_p()
        _lib.check(_lib.lib.kge_pipeline_create(C.byref(d), _lib.stream_ptr(), C.byref(h)))
        self._h, self._h_sig = h, sig

    def host_stats(self, reset=True):
        """kge_pipeline_stats as a dict (ms): where the thread that issues the steps spends its time."""
        if self._h is None:
            return None
        out = (_lib.C.c_double * 8)()
        _lib.check(_lib.lib.kge_pipeline_stats(self._h, out, 1 if reset else 0))
        return dict(run_ms=out[0], wait_prev_launch_ms=out[1], graph_launch_ms=out[2], graph_launches=int(out[3]), launches_that_waited=int(out[4]),
                    wait_by_stage_ms=dict(draw=out[5], plan=out[6], step=out[7]))

    def trace(self, n=0):
        """Stage timeline of the native driver (kge_pipeline_trace): trace(n) arms timing events around the next n stage launches,
        trace() reads them back as rows (stage name, batch, begin us, end us)."""
        if self._h is None:
            return []
        if n > 0:
            _lib.check(_lib.lib.kge_pipeline_trace(self._h, int(n), None, 0))
            return []
        cap = 4096
        out = (_lib.C.c_double * (4 * cap))()
        k = int(_lib.check(_lib.lib.kge_pipeline_trace(self._h, 0, out, cap)))
        names = ("rounds", "draw", "plan", "step")
        return [(names[int(out[4 * i])], int(out[4 * i + 1]), out[4 * i + 2] * 1e3, out[4 * i + 3] * 1e3) for i in range(k)]

    def close(self):
        """Release the native driver (graphs, streams, events); the next start() captures again."""
        if self._h is not None:
            _lib.lib.kge_pipeline_destroy(self._h)
            self._h, self._h_sig, self._native_active = None, None, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _leave_native(self):
        """The flags changed mid-epoch (e.g. a profiling pass switches to serial launches): hand over to the Python loop.  The
        caller's stream then waits for the producer streams; batches t and t+1 are already drawn and planned -- batch k into plan
        buffer k % np of the driver (three buffers by default), candidate buffer k & 1 -- and the Python loop carries on with the
        driver's buffer numbering, kernel by kernel (its captured stages are per buffer parity).  Step count and learning rate live
        in the device step state, which both loops advance; model.optimizer.iterations is brought up to it here."""
        if self._nstream is not None:
            self._nstream.synchronize()  # (host wait, then a join that is satisfied at once: no stream-to-stream wait left pending)
        self._on_native(lambda: _lib.check(_lib.lib.kge_pipeline_finish(self._h, _lib.stream_ptr())))
        st4 = (_lib.C.c_int64 * 4)()
        _lib.check(_lib.lib.kge_pipeline_state(self._h, st4))
        if int(st4[2]):
            raise RuntimeError("TrainPipeline: the native driver left the current tables in the twins (a failed run); the epoch cannot be carried on")
        if int(st4[1]) != self.t:
            raise RuntimeError(f"TrainPipeline: the native driver issued {int(st4[1])} steps of this epoch, this object counted {self.t}")
        self._np, self._taken_over = int(st4[0]), True
        self._native_active = False
        main = torch.cuda.current_stream()
        main.synchronize()
        self.model.optimizer.iterations = int(self.step_dev[0].item())
        for st in (self.s1, self.s2, self.s3):
            st.wait_stream(main)

    def run(self, n, join=True):
        """n training steps (each followed by the draw and the plan of the batch two steps ahead); asynchronous.  join=False (native
        driver only): the caller's stream is NOT made to wait for the steps -- the caller synchronises with the device (or calls
        finish()) before it touches the tables; what long runs should use (see _on_native)."""
        n = int(n)
        if self._native_active and self.native_ok():
            if not self.counted:  # dropout: the steps take the seeds model._seed + 1, + 2, ... like n calls of train_step would
                _lib.check(_lib.lib.kge_pipeline_set_seed(self._h, int(self.model._seed)))
                self.model._seed += n
            self._on_native(lambda: _lib.check(_lib.lib.kge_pipeline_run(self._h, n, _lib.stream_ptr())), join)
            self.t += n
            return
        # Kernel-by-kernel launches (serial = a profiling pass) of the step between twin tables: as chained steps, i.e. the SAME one-launch
        # form the native driver issues (closers inside step1_kernel), so that what such a pass times is the kernel of the timed region
        chain = self.serial and not self.use_graphs and getattr(self.model, "_step_twins", None) is not None
        if chain:
            self.model._chain_seq = 0
        try:
            for _ in range(n):
                self.tick()
        finally:
            if chain:
                self.model._chain_seq = None

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

    def _produce(self, k):
        """Issue the draw and the plan of batch k on their streams."""
        if k >= self.n:
            return
        j, pj = k % 2, k % self._np
        s1, s2 = (self.s1, self.s2) if not self.serial else (torch.cuda.current_stream(),) * 2
        with torch.cuda.stream(s1):
            s1.wait_event(self.freed[j])
            self._run("S", j)
            self.ready[j].record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(self.ready[j])
            s2.wait_event(self.pfreed[pj])
            self._run("P", j, pj)
            self.planned[pj].record(s2)
            self.freed[j].record(s2)

    def start(self):
        """Prologue: batches 0 and 1 on their way.  Switches the optimizer to the device step state."""
        mo = self.model
        main = torch.cuda.current_stream()
        if self.row_shard is None:  # (a row-shard trainer keeps its own workspace and only this rank's share of the moments)
            mo._ensure_train_state(self.B * self.ns, self.B * self.ns)
        for pl in self.plans:  # counters of a plan that was built but never consumed (an epoch cut short) would poison the next build
            pl.buf.zero_()
        self._np, self._taken_over = 2, False
        if self.between is None:
            mo._step_dev = self.step_dev
            # twin tables (kge_train_step_swap: lr_t of every update comes from the device step state, so the step may also be one
            # of this object's captured stages); with dropout the three-launch step is the faster one
            if os.environ.get("KGE_PIPE_SWAP", "1") != "0" and float(getattr(mo, "dp", 0.0) or 0.0) == 0.0 and self.comm is None and self.row_shard is None:
                if self._twins is None or self._twins[0].shape != mo.E.shape or self._twins[1].shape != mo.Rel.shape:
                    self._twins = (torch.empty_like(mo.E), torch.empty_like(mo.Rel))
                mo._step_twins = self._twins
            o = mo.optimizer
            od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, int(o.iterations) + 1, None, None, None, None,
                              self.step_dev.data_ptr())
            _lib.check(_lib.lib.kge_opt_state_init(_lib.C.byref(od), _lib.stream_ptr()))
        self.sampler.prepare(self.B, self.ns, self.num_e)  # workspace + jump polynomials: nothing is allocated inside a capture
        if self.native_ok():
            def go():
                self._ensure_native()
                _lib.check(_lib.lib.kge_pipeline_start(self._h, self.n, _lib.stream_ptr()))
            self._on_native(go)
            self._native_active = True
            return
        self._native_active = False
        if self.use_graphs and (not self.graphs or self._sig != self._signature()):
            self._capture()
        for st in (self.s1, self.s2, self.s3):
            st.wait_stream(main)
        self._produce(0)
        self._produce(1)

    def tick(self):
        """One training step; then the draw and the plan of the batch two steps ahead are issued."""
        if self._native_active:
            if self.native_ok():
                return self.run(1)
            self._leave_native()
        main = torch.cuda.current_stream()
        j = self.t % self._np
        main.wait_event(self.planned[j])
        self._run("T", j)
        self._after()
        self.pfreed[j].record(main)
        self._produce(self.t + 2)
        self.t += 1

    def finish(self):
        mo = self.model
        main = torch.cuda.current_stream()
        if self._native_active:
            self._leave_native()
        for st in (self.s1, self.s2, self.s3):
            main.wait_stream(st)
        err = int(self.sampler.err.item())  # (synchronises: the epoch is over)
        if err:
            self.sampler.err.zero_()
            mo._step_dev = mo._step_twins = None
            raise RuntimeError(f"device sampler error flag {err}: the masked-rejection buffer ran short; the batches drawn since are not the "
                               "reference's (kge_build_batch_device: err_flag bit 1)")
        if getattr(mo, "_step_dev", None) is not None:
            mo.optimizer.iterations = int(self.step_dev[0].item())
            mo._step_dev = mo._step_twins = None
            tot = self.step_dev.view(torch.float64)[2].clone()  # running sum kept by the library (kge_opt_desc.step_dev), a double
            if self.row_shard is not None and self.row_shard.tables.ws > 1:  # (every rank's state summed ITS partial losses)
                torch.distributed.all_reduce(tot, group=self.row_shard.tables.group)
            return tot
        return self.loss_sum

    def run_epoch(self, index, order):
        """All batches of one epoch; returns the summed loss (0-d float64 device tensor)."""
        self.load_epoch(index, order)
        try:
            self.start()
            self.run(self.n, join=False)  # (finish() synchronises)
        except BaseException:
            self.model._step_dev = self.model._step_twins = None  # never leave the model on the device step state
            raise
        return self.finish()
