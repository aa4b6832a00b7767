"""Multi-GPU orchestration of the two halves of the path that shard; one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* training  : triples are independent units -> data-parallel replicas.  The one exchange per step is the
              all-reduce of the dense gradient tables between kge_train_grads and kge_opt_apply; every rank
              then applies the identical update (tables stay bit-identical across ranks).
              `RowShardedTables` / `row_sharded_step` is the other split SURVEY 8(e) names -- row-sharded tables, OWNER COMPUTES: every
              rank draws the SAME batch (the samplers are bit-exact functions of their seeds: no communication), updates only the
              entity rows and relation rows of its own window (gradient from the row's list + optimizer, moments sharded), then the
              ranks all-gather the updated window rows and sum their partial losses.  One batch per step like the reference
              (tests.py:55-61), N-rank tables == one-GPU tables, half the bytes of a gradient all-reduce.
* evaluation: the entity table is row-sharded at rest.  Schedule (i) (`sharded_eval`): the shards are all-gathered
              into a full table on every GPU (YAGO3-10: 98.5 MB, one collective per evaluation, not per query), the
              queries are split across ranks and the integer ranks are all-gathered back.  Schedule (ii)
              (`resident_shard_eval`): the table STAYS sharded; only the rows the queries reference (query entity +
              target, <= 2 rows per test triple) are exchanged, every rank counts `score > target` over its own rows
              for ALL queries (kge_rank_counts with a candidate range) and the int32 counts are all-reduced -- traffic
              independent of the table size, the schedule that scales past one GPU's memory.

The functions here only move tensors; the compute is injected (the HIP-backed model methods in production, so the
same orchestration can be exercised on CPU tensors with gloo).  The reference has no counterpart (single process).
"""
import torch
import torch.distributed as dist


def world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_range(n, rank, world_size):
    """Contiguous row shard [lo, hi) of n rows for `rank`; all shards but the last hold ceil(n/world) rows."""
    per = (n + world_size - 1) // world_size
    return min(n, rank * per), min(n, (rank + 1) * per)


def allgather_rows(shard, n_rows, group=None):
    """Row-sharded table -> full (n_rows, d) table on every rank (one all_gather_into_tensor of equal-size padded shards)."""
    rank, ws = world(group)
    if ws == 1:
        return shard[:n_rows]
    per = (n_rows + ws - 1) // ws
    d = shard.shape[1]
    mine = shard
    if shard.shape[0] != per:
        mine = torch.zeros((per, d), dtype=shard.dtype, device=shard.device)
        mine[:shard.shape[0]] = shard
    full = torch.empty((per * ws, d), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(full, mine.contiguous(), group=group)
    return full[:n_rows]


def split_queries(n_queries, rank, world_size):
    """Indices of the queries rank `rank` evaluates (strided, so every rank gets the same mix of heads and tails)."""
    return torch.arange(rank, n_queries, world_size)


_GATHER_PERM = {}


def _gather_perm(n_queries, ws, per, device):
    """Position in the all-gathered (ws * per) buffer of every query's rank, for the strided split of split_queries (cached: an
    evaluation then costs one collective and one gather kernel instead of a loop over the ranks)."""
    key = (n_queries, ws, str(device))
    if key not in _GATHER_PERM:
        q = torch.arange(n_queries)
        _GATHER_PERM[key] = ((q % ws) * per + q // ws).to(device)
    return _GATHER_PERM[key]


def gather_ranks(local_ranks, n_queries, group=None):
    """Inverse of split_queries: int64 ranks of all n_queries on every rank."""
    rank, ws = world(group)
    if ws == 1:
        return local_ranks
    per = (n_queries + ws - 1) // ws
    buf = torch.zeros(per, dtype=torch.int64, device=local_ranks.device)
    buf[:local_ranks.numel()] = local_ranks
    out = torch.empty(per * ws, dtype=torch.int64, device=local_ranks.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return out.index_select(0, _gather_perm(n_queries, ws, per, local_ranks.device))


def allreduce_sum_(tensors, group=None):
    """In-place sum over ranks of every tensor in the list (dense gradient tables, loss scalars, count vectors)."""
    _, ws = world(group)
    if ws > 1:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


def dp_train_step(grads_fn, apply_fn, grad_tensors, loss_tensor=None, group=None):
    """One data-parallel step: grads_fn() accumulates this rank's gradients into grad_tensors (and the loss into
    loss_tensor), the tables are summed over ranks, apply_fn(1/world) applies the mean gradient and re-zeroes.
    With equal per-rank batch sizes this equals one step on the concatenated batch (reduce_mean over it)."""
    _, ws = world(group)
    grads_fn()
    allreduce_sum_(list(grad_tensors) + ([loss_tensor] if loss_tensor is not None and ws > 1 else []), group)
    if loss_tensor is not None and ws > 1:
        loss_tensor /= ws
    apply_fn(1.0 / ws)


class RowShardedTables:
    """Both parameter tables of an owner-computes training run as TWIN buffers padded to `world` equal row shards (so that the exchange
    of the updated rows is one in-place all-gather per table): a step reads the current pair and writes the other one
    (kge_train_step_swap_window), the exchange completes it, the pairs swap.  Rank r owns entity rows [r * per_e, (r + 1) * per_e) and
    relation rows [r * per_r, (r + 1) * per_r), clipped to the tables (= shard_range)."""

    def __init__(self, E, Rel, group=None):
        self.group = group
        self.rank, self.ws = world(group)
        self.N, self.R = int(E.shape[0]), int(Rel.shape[0])
        self.per_e, self.per_r = (self.N + self.ws - 1) // self.ws, (self.R + self.ws - 1) // self.ws
        mk = lambda t, per: [torch.zeros((per * self.ws, t.shape[1]), dtype=t.dtype, device=t.device) for _ in range(2)]  # noqa: E731
        self.bufE, self.bufR = mk(E, self.per_e), mk(Rel, self.per_r)
        self.bufE[0][:self.N].copy_(E)
        self.bufR[0][:self.R].copy_(Rel)
        self.cur = 0
        self.ent = shard_range(self.N, self.rank, self.ws)
        self.rel = shard_range(self.R, self.rank, self.ws)

    @property
    def E(self):
        return self.bufE[self.cur][:self.N]

    @property
    def Rel(self):
        return self.bufR[self.cur][:self.R]

    @property
    def E_out(self):
        return self.bufE[self.cur ^ 1][:self.N]

    @property
    def Rel_out(self):
        return self.bufR[self.cur ^ 1][:self.R]

    def exchange(self, loss=None):
        """The window rows every rank wrote into the out buffers -> complete tables on every rank (one all-gather per table, equal
        padded shards, in place); `loss` (this rank's partial loss, a tensor) is summed over the ranks in place; the pairs swap."""
        if self.ws > 1:
            for buf, per in ((self.bufE[self.cur ^ 1], self.per_e), (self.bufR[self.cur ^ 1], self.per_r)):
                mine = buf[self.rank * per:(self.rank + 1) * per]
                if dist.get_backend(self.group) != "nccl":
                    # gloo (rehearsals, CPU tests): its all-gathers of device tensors are pathologically slow (190 ms for 5.9 MB between two
                    # ranks on one GPU, tools/rows_probe.py; its all-reduce moves twice the bytes in 4 ms), so the other ranks' slices are
                    # zeroed and the table is all-reduced: x + 0 is exact
                    for r in range(self.ws):
                        if r != self.rank:
                            buf[r * per:(r + 1) * per].zero_()
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.all_gather_into_tensor(buf, mine, group=self.group)
            if loss is not None:
                dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        self.cur ^= 1
        return loss

    def exchanged_bytes(self):
        """Bytes one rank receives per step."""
        return 4 * ((self.ws - 1) * (self.per_e * self.bufE[0].shape[1] + self.per_r * self.bufR[0].shape[1]))


def row_sharded_step(tables, window_step, loss=None):
    """One owner-computes step: window_step(E_in, Rel_in, E_out, Rel_out, (ent_lo, ent_hi), (rel_lo, rel_hi)) updates this rank's rows
    of the out tables from the in tables (the HIP step kge_train_step_swap_window in production; injected, so that the same
    orchestration runs on CPU tensors under gloo) and leaves this rank's partial loss in `loss`; then the exchange."""
    window_step(tables.E, tables.Rel, tables.E_out, tables.Rel_out, tables.ent, tables.rel)
    return tables.exchange(loss)


class RowShardTrainer:
    """Owner-computes training of an EmbeddingModel over a process group (one process per GPU): the model's tables move into the padded
    twin buffers of a RowShardedTables, this rank keeps the Adam moments of ITS rows only (1 / world of the optimizer's state and
    streams), and step(plan) = kge_train_step_swap_window on the rank's window + the exchange.  Every rank must feed the SAME batches
    (same sampler seeds on every rank).  After every step model.E / model.Rel are the complete, current tables on every rank --
    equal to what one GPU computes with kge_train_step_swap on that batch (up to fp32 summation order of the loss only: the rows
    themselves are computed by the same waves in the same order)."""

    def __init__(self, model, group=None):
        from . import _lib
        self._lib, self.model = _lib, model
        if model.optimizer is None:
            raise RuntimeError("compile the model first")
        if float(getattr(model, "dp", 0.0) or 0.0) != 0.0:
            raise NotImplementedError("row-sharded training is built for models without dropout")
        self.tables = t = RowShardedTables(model.E, model.Rel, group)
        self._install()
        (lo, hi), (rlo, rhi) = t.ent, t.rel
        de, dr = int(model.E.shape[1]), int(model.Rel.shape[1])
        self.slots = None
        if model.optimizer.kind == "adam":  # this rank's share of the moments: rows [lo, hi) / [rlo, rhi), addressed by global row id
            mk = lambda rows, w: torch.zeros((max(1, rows), w), dtype=torch.float32, device=model.device)  # noqa: E731
            self.slots = (mk(hi - lo, de), mk(hi - lo, de), mk(rhi - rlo, dr), mk(rhi - rlo, dr))
            self._slot_ptrs = [self.slots[0].data_ptr() - 4 * lo * de, self.slots[1].data_ptr() - 4 * lo * de,
                               self.slots[2].data_ptr() - 4 * rlo * dr, self.slots[3].data_ptr() - 4 * rlo * dr]
        else:
            self._slot_ptrs = [None] * 4
        self.win = _lib.RowWindow(lo, hi, rlo, rhi)
        self.loss = torch.zeros(1, dtype=torch.float32, device=model.device)
        self._ws = None
        self._own_state = None

    def _install(self):
        self.model.entity_embedding.weights = self.tables.E
        self.model.relational_embedding.weights = self.tables.Rel

    def step(self, plan):
        """One step on the batch `plan` was built from (a models.BatchPlan; consumed).  Returns the batch loss (0-d device tensor,
        summed over the ranks' partial losses)."""
        _lib, mo, t = self._lib, self.model, self.tables
        C, lib = _lib.C, _lib.lib
        md, ld = mo._model_desc(True), mo._loss_desc()
        need = _lib.check(lib.kge_train_step_workspace_bytes(C.byref(md), C.byref(ld), plan.m_pos, plan.m_neg))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(int(need), dtype=torch.uint8, device=mo.device)  # zero-filled once (kge_hip.h)
        o = mo.optimizer
        sd = getattr(mo, "_step_dev", None)
        if sd is None:  # (outside a TrainPipeline: a step state of this trainer's own)
            if self._own_state is None:
                self._own_state = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device=mo.device)
                od0 = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, int(o.iterations) + 1, None, None, None, None,
                                   self._own_state.data_ptr())
                _lib.check(lib.kge_opt_state_init(C.byref(od0), _lib.stream_ptr()))
            sd = self._own_state
        o.iterations += 1
        od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, max(1, o.iterations), *self._slot_ptrs, sd.data_ptr())
        _lib.check(lib.kge_train_step_swap_window(C.byref(md), C.byref(ld), C.byref(od), t.E.data_ptr(), t.Rel.data_ptr(), t.E_out.data_ptr(),
                                                  t.Rel_out.data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), plan.m_pos, plan.m_neg,
                                                  self.loss.data_ptr(), 0, self._ws.data_ptr(), self._ws.numel(), C.byref(self.win),
                                                  _lib.stream_ptr()))
        t.exchange(self.loss)
        self._install()
        return self.loss[0]


def make_rccl_comm(group=None):
    """An RCCL communicator of this process group's ranks, made through RCCL's own C API (torch.distributed does not hand its
    communicators out): rank 0 draws the ncclUniqueId, the group broadcasts its 128 bytes (any backend), every rank joins with
    ncclCommInitRank on its current device.  What the library's collectives (kge_allreduce_f32 / kge_allgather_rows /
    kge_allreduce_counts) and the native data-parallel pipeline (kge_pipeline_desc.comm) take.  Returns (comm as an int, destroy()).
    Raises when RCCL is not in the process or refuses (e.g. two ranks on one device)."""
    import ctypes as C
    rank, ws = world(group)
    path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln), "librccl.so.1")
    rccl, why = None, None
    try:
        rccl = C.CDLL(path)  # (the copy PyTorch-ROCm mapped, as the library's own run-time lookup does)
    except OSError as e:
        why = f"RCCL not loadable ({e})"
    if ws > 1:
        # Pre-flight over the EXISTING process group, before any rank enters ncclCommInitRank: that call is itself collective, so a rank
        # that raises inside it (RCCL refuses two ranks on one device) would leave its peers blocked there.  Every rank reports
        # (host, device identity, RCCL loaded); all ranks or none go on.
        import socket
        ident = None
        if torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            ident = str(getattr(pr, "uuid", None) or (getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None), torch.cuda.current_device()))
        seen = [None] * ws
        dist.all_gather_object(seen, (socket.gethostname(), ident, why), group=group)
        bad = [f"rank {r}: {w}" for r, (_h, _i, w) in enumerate(seen) if w]
        devs = [(h, i) for h, i, _w in seen]
        if len(set(devs)) < ws:
            bad.append(f"ranks share a device ({devs}): RCCL refuses duplicate devices in one communicator")
        if bad:
            raise RuntimeError("make_rccl_comm: " + "; ".join(bad))
    elif why:
        raise RuntimeError("make_rccl_comm: " + why)

    class UniqueId(C.Structure):  # ncclUniqueId: passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rc0 = rccl.ncclGetUniqueId(C.byref(uid)) if rank == 0 else 0
    if ws > 1:  # (rank 0 broadcasts also when it failed, so that every rank leaves together instead of waiting for the id)
        box = [(bytes(uid) if rc0 == 0 else None) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        C.memmove(C.byref(uid), box[0], 128)
    elif rc0 != 0:
        raise RuntimeError(f"ncclGetUniqueId failed ({rc0})")
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), ws, uid, rank)
    if rc != 0 or not comm.value:
        raise RuntimeError(f"ncclCommInitRank failed ({rc}) for rank {rank} of {ws}")
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    handle = int(comm.value)
    return handle, (lambda: rccl.ncclCommDestroy(C.c_void_p(handle)))


def sharded_eval(test, side_fn, n_entities, table_shard=None, set_table=None, group=None):
    """Filtered 1-vs-all ranks of `test` (list of triples) with the entity table row-sharded across ranks.
    table_shard/set_table: this rank's rows and a callback installing the gathered full table (skipped when None).
    side_fn(queries_subset) -> (tail_ranks, head_ranks) int64 tensors for the given subset (the GPU ranker).
    Returns (tail_ranks, head_ranks) for all of `test` on every rank."""
    rank, ws = world(group)
    if table_shard is not None and set_table is not None:
        set_table(allgather_rows(table_shard, n_entities, group))
    idx = split_queries(len(test), rank, ws)
    mine = [test[i] for i in idx.tolist()]
    if mine:
        t, h = side_fn(mine)
    else:
        t = h = torch.zeros(0, dtype=torch.int64)
    t = torch.as_tensor(t, dtype=torch.int64)
    h = torch.as_tensor(h, dtype=torch.int64)
    return gather_ranks(t, len(test), group), gather_ranks(h.to(t.device), len(test), group)


def referenced_rows(test):
    """Sorted distinct entity ids the queries of `test` (n,3) need as rows: the query entity and the target of either side."""
    import numpy as np
    t = np.asarray(test).reshape(-1, 3)
    return np.unique(np.concatenate([t[:, 0], t[:, 2]])).astype(np.int64)


def exchange_referenced_rows(shard, lo, ids, group=None):
    """(len(ids), d) rows of the global table for the sorted global ids `ids`, on every rank, with the table row-sharded
    (this rank holds rows [lo, lo + len(shard))): every rank fills the rows it owns, zeros elsewhere, one all-reduce(sum)
    (x + 0 is exact in floating point)."""
    ids_t = torch.as_tensor(ids, dtype=torch.int64, device=shard.device)
    ref = torch.zeros((ids_t.numel(), shard.shape[1]), dtype=shard.dtype, device=shard.device)
    own = (ids_t >= lo) & (ids_t < lo + shard.shape[0])
    if bool(own.any()):
        ref[own] = shard.index_select(0, ids_t[own] - lo)
    allreduce_sum_([ref], group)
    return ref


class ResidentShardPlan:
    """Everything of schedule (ii) that depends on the test set and on the shard boundaries only -- which rows the queries
    reference, which of them this rank owns, the queries re-addressed into the referenced rows -- prepared once (a model is
    validated on the same test set epoch after epoch) and kept on the device, so that an evaluation is: gather the owned rows,
    ONE all-reduce of the referenced rows, and per side one kge_rank_counts over the shard + one all-reduce of int32 counts.
    No host work and no host synchronisation per evaluation."""

    def __init__(self, test, shard_rows, lo, device, n_entities=None, group=None):
        """n_entities (rows of the whole table, sharded by shard_range): with it every rank knows which slice of the referenced
        rows every other rank owns (the ids are sorted and the shards contiguous), and the exchange is one all-gather of the owned
        rows padded to the largest slice instead of an all-reduce over rows that are mostly zeros (half the bytes on the wire)."""
        import numpy as np
        t = np.ascontiguousarray(np.asarray(test).reshape(-1, 3))
        ids = referenced_rows(t)
        self.gather_idx = None
        _, ws = world(group)
        if n_entities is not None and ws > 1:
            bounds = [shard_range(int(n_entities), r, ws) for r in range(ws)]
            cuts = [(int(np.searchsorted(ids, a)), int(np.searchsorted(ids, b))) for a, b in bounds]
            self.pad = max(1, max(b - a for a, b in cuts))
            idx = np.empty(ids.shape[0], dtype=np.int64)
            for r, (a, b) in enumerate(cuts):
                idx[a:b] = r * self.pad + np.arange(b - a)
            self.gather_idx = torch.as_tensor(idx, dtype=torch.int64, device=device)
        own = np.nonzero((ids >= lo) & (ids < lo + shard_rows))[0]
        self.n = int(t.shape[0])
        self.n_ref = int(ids.shape[0])
        self.lo = int(lo)
        self.own_pos = torch.as_tensor(own, dtype=torch.int64, device=device)             # positions in the referenced rows ...
        self.own_rows = torch.as_tensor(ids[own] - lo, dtype=torch.int64, device=device)   # ... of these rows of the shard
        local = t.copy()
        local[:, 0] = np.searchsorted(ids, t[:, 0])
        local[:, 2] = np.searchsorted(ids, t[:, 2])
        self.local = torch.as_tensor(local, dtype=torch.int32, device=device).contiguous()
        self.ref = None

    def referenced(self, shard, group=None):
        """(n_ref, d) referenced rows on every rank: owned rows filled in, zeros elsewhere, one all-reduce(sum) (x + 0 is exact)."""
        if self.gather_idx is not None:
            _, ws = world(group)
            send = torch.zeros((self.pad, shard.shape[1]), dtype=shard.dtype, device=shard.device)
            if self.own_rows.numel():
                send[:self.own_rows.numel()] = shard.index_select(0, self.own_rows)
            recv = torch.empty((self.pad * ws, shard.shape[1]), dtype=shard.dtype, device=shard.device)
            dist.all_gather_into_tensor(recv, send, group=group)
            self.ref = recv.index_select(0, self.gather_idx)
            return self.ref
        if self.ref is None or self.ref.shape[1] != shard.shape[1] or self.ref.dtype != shard.dtype:
            self.ref = torch.empty((self.n_ref, shard.shape[1]), dtype=shard.dtype, device=shard.device)
        self.ref.zero_()
        if self.own_pos.numel():
            self.ref.index_copy_(0, self.own_pos, shard.index_select(0, self.own_rows))
        allreduce_sum_([self.ref], group)
        return self.ref

    def run(self, count_fn, shard, group=None, both=False):
        """(tail_ranks, head_ranks) int64 on every rank; count_fn(ref_rows, test_local (device int32), side) as in resident_shard_eval.
        both=True: count_fn is called ONCE with side 2 and returns the 2 n counts of both sides, tail queries first (kge_rank_counts,
        side 2: one sweep of the shard for all queries) -- one launch and one all-reduce of the counts instead of two."""
        ref = self.referenced(shard, group)
        if both:
            c = count_fn(ref, self.local, 2).to(torch.int32)
            allreduce_sum_([c], group)
            r = 1 + c.to(torch.int64)
            return r[:self.n], r[self.n:]
        out = []
        for side in (0, 1):
            c = count_fn(ref, self.local, side).to(torch.int32)
            allreduce_sum_([c], group)
            out.append(1 + c.to(torch.int64))
        return out[0], out[1]


def resident_shard_eval(test, count_fn, shard, lo, group=None):
    """SURVEY 8(e) schedule (ii): filtered 1-vs-all ranks of `test` ((n,3) integer array) with the entity table row-sharded and
    never gathered.  count_fn(ref_rows, test_local, side) -> int32 (n,) tensor of #{kept c in this rank's shard:
    score(q, c) > score(q, target)} where ref_rows holds the referenced rows and test_local is `test` (an int32 tensor on the shard's
    device) with s / o replaced by their positions in ref_rows (the GPU implementation is EmbeddingModel.rank_counts_shard =
    kge_rank_counts with E = ref_rows, E_cand = shard, cand_offset = lo).  Returns (tail_ranks, head_ranks) int64 tensors on every
    rank.  One-shot form of ResidentShardPlan (prepare once, run per evaluation)."""
    return ResidentShardPlan(test, shard.shape[0], lo, shard.device).run(count_fn, shard, group)
