"""World-size-2 gloo tests (CPU) of the multi-GPU orchestration in kge_keras_amd.dist.  The compute that the GPU
kernels do in production is injected from the oracle here, so what is under test is the sharding / collective logic:
row-shard all-gather, query split + rank gather, data-parallel gradient all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _eval_worker(rank, world):
    from kge_keras_amd import dist as D
    rs = np.random.RandomState(0)
    N, R, d = 103, 5, 8  # N not divisible by world: last shard is short
    E = rs.uniform(-1, 1, (N, d))
    Rl = rs.uniform(-1, 1, (R, d))
    test = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(31)]
    filt = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(200)]
    filt = [t for t in filt if t not in set(test)]
    lo, hi = D.shard_range(N, rank, world)
    shard = torch.from_numpy(E[lo:hi].copy())
    box = {}

    def side_fn(queries):
        t, h = O.ranks_closed_form("DistMult", box["E"].numpy(), Rl, queries, filt)
        return torch.from_numpy(t), torch.from_numpy(h)

    t, h = D.sharded_eval(test, side_fn, N, shard, lambda full: box.__setitem__("E", full))
    assert torch.equal(box["E"], torch.from_numpy(E))  # all-gather rebuilt the table exactly
    return t.numpy(), h.numpy(), (lo, hi)


def test_sharded_eval_matches_unsharded():
    out = _spawn(_eval_worker)
    rs = np.random.RandomState(0)
    N, R, d = 103, 5, 8
    E = rs.uniform(-1, 1, (N, d))
    Rl = rs.uniform(-1, 1, (R, d))
    test = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(31)]
    filt = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(200)]
    filt = [t for t in filt if t not in set(test)]
    t, h = O.ranks_closed_form("DistMult", E, Rl, test, filt)
    for r in range(2):
        assert np.array_equal(out[r][0], t) and np.array_equal(out[r][1], h)
    assert out[0][2] == (0, 52) and out[1][2] == (52, 103)


def _resident_worker(rank, world):
    """Schedule (ii): shards stay resident; referenced rows exchanged; counts over own rows; all-reduce of int32 counts."""
    from kge_keras_amd import dist as D
    rs = np.random.RandomState(3)
    N, R, d = 103, 5, 8
    E = rs.uniform(-1, 1, (N, d))
    Rl = rs.uniform(-1, 1, (R, d))
    test = np.stack([rs.randint(0, N, 41), rs.randint(0, R, 41), rs.randint(0, N, 41)], 1)
    filt = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(300)]
    tset = set(map(tuple, test.tolist()))
    filt = [t for t in filt if t not in tset]
    fh, ft = O.build_filters(filt)
    lo, hi = D.shard_range(N, rank, world)
    shard = torch.from_numpy(E[lo:hi].copy())

    def count_fn(ref, local, side):  # what kge_rank_counts does on the GPU, restated with NumPy on this rank's rows
        if side == 2:  # both sides in one call: 2 n counts, tail queries first
            return torch.cat([count_fn(ref, local, 0), count_fn(ref, local, 1)])
        ref = ref.numpy()
        out = np.zeros(len(local), dtype=np.int32)
        for i, (s, p, o) in enumerate(local.tolist()):
            q = ref[s] * Rl[p] if side == 0 else Rl[p] * ref[o]
            tgt = float(q @ ref[o if side == 0 else s])
            sc = shard.numpy() @ q
            gs, gp, go = test[i].tolist()
            excl = ft.get((gs, gp), []) if side == 0 else fh.get((gp, go), [])
            keep = np.ones(hi - lo, dtype=bool)
            for x in [go if side == 0 else gs]:  # the target itself (the kernels compare bit-identical chains: never greater)
                if lo <= x < hi:
                    keep[x - lo] = False
            for x in excl:
                if lo <= x < hi:
                    keep[x - lo] = False
            out[i] = int(np.sum((sc > tgt) & keep))
        return torch.from_numpy(out)

    t, h = D.resident_shard_eval(test, count_fn, shard, lo)
    for n_all in (None, N):  # the prepared form, run twice on the same plan: rows exchanged by all-reduce / by all-gather
        plan = D.ResidentShardPlan(test, hi - lo, lo, "cpu", n_entities=n_all)
        assert (plan.gather_idx is not None) == (n_all is not None)
        for _ in range(2):
            t2, h2 = plan.run(count_fn, shard)
            assert torch.equal(t, t2) and torch.equal(h, h2)
            t3, h3 = plan.run(count_fn, shard, both=True)  # one count call and one all-reduce for both sides
            assert torch.equal(t, t3) and torch.equal(h, h3)
    return t.numpy(), h.numpy()


def test_resident_shard_eval_matches_unsharded():
    out = _spawn(_resident_worker)
    rs = np.random.RandomState(3)
    N, R, d = 103, 5, 8
    E = rs.uniform(-1, 1, (N, d))
    Rl = rs.uniform(-1, 1, (R, d))
    test = np.stack([rs.randint(0, N, 41), rs.randint(0, R, 41), rs.randint(0, N, 41)], 1)
    filt = [(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(300)]
    tset = set(map(tuple, test.tolist()))
    filt = [t for t in filt if t not in tset]
    t, h = O.ranks_closed_form("DistMult", E, Rl, [tuple(x) for x in test.tolist()], filt)
    for r in range(2):
        assert np.array_equal(out[r][0], t) and np.array_equal(out[r][1], h)


def _dp_worker(rank, world):
    from kge_keras_amd import dist as D
    rs = np.random.RandomState(1)
    N, R, d, m = 30, 4, 6, 64
    E = rs.uniform(-0.5, 0.5, (N, d))
    Rl = rs.uniform(-0.5, 0.5, (R, d))
    batches = [(np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1),
                np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)) for _ in range(2 * world)]
    opt = O.AdamTF()
    gE, gR, loss = torch.zeros(N, d, dtype=torch.float64), torch.zeros(R, d, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    for step in range(2):
        pos, neg = batches[step * world + rank]

        def grads_fn():
            L, _, _, a, b = O.dense_grads("DistMult", E, Rl, pos, neg, "pointwize_logistic", 1.0)
            gE.add_(torch.from_numpy(a))
            gR.add_(torch.from_numpy(b))
            loss[0] = L

        def apply_fn(scale):
            opt.begin_step()
            opt.apply("E", E, gE.numpy() * scale)
            opt.apply("R", Rl, gR.numpy() * scale)
            gE.zero_()
            gR.zero_()

        D.dp_train_step(grads_fn, apply_fn, [gE, gR], loss)
    return E, Rl, float(loss[0])


def test_data_parallel_step_equals_big_batch():
    out = _spawn(_dp_worker)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])  # replicas stay identical
    rs = np.random.RandomState(1)
    N, R, d, m = 30, 4, 6, 64
    E = rs.uniform(-0.5, 0.5, (N, d))
    Rl = rs.uniform(-0.5, 0.5, (R, d))
    batches = [(np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1),
                np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)) for _ in range(4)]
    opt = O.AdamTF()
    L = None
    for step in range(2):
        pos = np.concatenate([batches[2 * step][0], batches[2 * step + 1][0]])
        neg = np.concatenate([batches[2 * step][1], batches[2 * step + 1][1]])
        L = O.train_step("DistMult", E, Rl, opt, pos, neg, "pointwize_logistic", 1.0)
    np.testing.assert_allclose(out[0][0], E, atol=1e-12)
    np.testing.assert_allclose(out[0][1], Rl, atol=1e-12)
    assert out[0][2] == pytest.approx(L, rel=1e-12)


def _row_batches():
    rs = np.random.RandomState(7)
    N, R, d, m = 37, 5, 6, 96  # neither table divides by the world size: the last shards are short
    E = rs.uniform(-0.5, 0.5, (N, d))
    Rl = rs.uniform(-0.5, 0.5, (R, d))
    tri = lambda: np.stack([rs.randint(0, N - 3, m), rs.randint(0, R, m), rs.randint(0, N - 3, m)], 1)  # noqa: E731  (entities N-3.. untouched)
    return N, R, d, E, Rl, [(tri(), tri()) for _ in range(3)]


def _row_worker(rank, world):
    """Owner-computes steps (dist.RowShardedTables / row_sharded_step): EVERY rank holds the same batch; it updates the entity rows and
    relation rows of its window only (gradient rows + Adam with this rank's share of the moments, restated with the oracle), then
    the window rows are all-gathered and the partial losses summed."""
    from kge_keras_amd import dist as D
    N, R, d, E, Rl, batches = _row_batches()
    tables = D.RowShardedTables(torch.from_numpy(E.copy()), torch.from_numpy(Rl.copy()))
    assert tables.ent == D.shard_range(N, rank, world) and tables.rel == D.shard_range(R, rank, world)
    opt = O.AdamTF()  # slots hold this rank's window rows only
    loss = torch.zeros(1, dtype=torch.float64)
    losses = []
    for pos, neg in batches:
        def window_step(E_in, R_in, E_out, R_out, ent, rel):
            Ein, Rin = E_in.numpy(), R_in.numpy()
            _L, sp, sn, gE, gR = O.dense_grads("DistMult", Ein, Rin, pos, neg, "pointwize_logistic", 1.0)
            opt.begin_step()
            (lo, hi), (rlo, rhi) = ent, rel
            we, wr = Ein[lo:hi].copy(), Rin[rlo:rhi].copy()
            opt.apply("E", we, gE[lo:hi])
            opt.apply("R", wr, gR[rlo:rhi])
            E_out[lo:hi] = torch.from_numpy(we)
            R_out[rlo:rhi] = torch.from_numpy(wr)
            # this rank's partial loss: the terms of the batch rows whose relation lies in its window (utils.py:291-292, both means)
            tp = np.log(1e-6 + 1 + np.exp(-sp)) / len(sp)
            tn = np.log(1e-6 + 1 + np.exp(sn)) / len(sn)
            inp, inn = (pos[:, 1] >= rlo) & (pos[:, 1] < rhi), (neg[:, 1] >= rlo) & (neg[:, 1] < rhi)
            loss[0] = tp[inp].sum() + tn[inn].sum()
        D.row_sharded_step(tables, window_step, loss)
        losses.append(float(loss[0]))
    assert tables.exchanged_bytes() == 4 * (world - 1) * (tables.per_e * d + tables.per_r * d)
    return tables.E.numpy().copy(), tables.Rel.numpy().copy(), losses


def test_row_sharded_owner_computes_step_equals_one_process():
    """Three owner-computes steps on two ranks == three ordinary steps of one process on the same batches: tables identical on both
    ranks and equal to the unsharded oracle (the update of a row depends on its own gradient and moments only), losses equal."""
    out = _spawn(_row_worker)
    N, R, d, E, Rl, batches = _row_batches()
    opt = O.AdamTF()
    Ls = [O.train_step("DistMult", E, Rl, opt, pos, neg, "pointwize_logistic", 1.0) for pos, neg in batches]
    for r in range(2):
        np.testing.assert_allclose(out[r][0], E, atol=1e-15)
        np.testing.assert_allclose(out[r][1], Rl, atol=1e-15)
        np.testing.assert_allclose(out[r][2], Ls, rtol=1e-12)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_shard_helpers_single_process():
    from kge_keras_amd import dist as D
    assert D.shard_range(123182, 7, 8) == (107786, 123182) and D.shard_range(123182, 0, 8) == (0, 15398)
    assert D.shard_range(3, 2, 8) == (2, 3) and D.shard_range(3, 5, 8) == (3, 3)  # ragged / empty shards
    t = torch.arange(12.).reshape(6, 2)
    assert torch.equal(D.allgather_rows(t, 6), t)
    assert D.split_queries(7, 1, 3).tolist() == [1, 4]
    assert torch.equal(D.gather_ranks(torch.tensor([5, 6]), 2), torch.tensor([5, 6]))
