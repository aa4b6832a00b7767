"""Tables past 2^31 elements (what 288 GB of HBM are for): every kernel of the path indexes rows with 64-bit arithmetic.  One entity
table of 10.8 M x 200 floats (8.6 GB; with its twin, Adam moments and gradient table ~43 GB on the device); batches that touch rows on
both sides of the 2^31-element mark; the oracle runs on the touched rows only (ids remapped into a compact table)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)
N, R, D = 10_800_000, 7, 200


def _need_memory():
    free, _total = torch.cuda.mem_get_info()
    if free < 70 * 2**30:
        pytest.skip("needs ~70 GB of free device memory")


def _batch(rs, n, hot):
    """rows over `hot` entity ids (a few hundred ids spread over the table, most of them past the 2^31-element mark)"""
    return np.stack([hot[rs.randint(0, len(hot), n)], rs.randint(0, R, n), hot[rs.randint(0, len(hot), n)]], 1).astype(np.int32)


def _compact(mdl, hot):
    """(compact entity table (len(hot), D) float64, id map) of the rows `hot`"""
    idx = torch.from_numpy(hot.astype(np.int64)).cuda()
    return mdl.E.index_select(0, idx).cpu().numpy().astype(np.float64), {int(g): i for i, g in enumerate(hot)}


def _remap(a, m):
    out = a.copy()
    out[:, 0] = [m[int(x)] for x in a[:, 0]]
    out[:, 2] = [m[int(x)] for x in a[:, 2]]
    return out


@pytest.mark.parametrize("path", ["step", "planned"])
def test_train_step_on_a_table_past_2_31_elements(path):
    _need_memory()
    from kge_keras_amd import models as M
    assert N * D > 2**31
    rs = np.random.RandomState(5)
    mdl = M.DistMult(e_dim=D, r_dim=D, num_entities=N, num_relations=R, dp=0.0, seed=3, loss_function="pointwize_logistic")
    mdl.E.mul_(40.0)    # (Glorot limits of a 10 M row table are ~7e-4: scale the rows up so that gradients are not all rounding noise)
    mdl.Rel.mul_(3.0)
    mdl.compile("adam")
    hot = np.unique(np.concatenate([rs.randint(0, N, 150), rs.randint(N - 60_000, N, 150), [0, N - 1, 2**31 // D, 2**31 // D + 1]])).astype(np.int64)
    Ec, m = _compact(mdl, hot)
    Ro = mdl.Rel.cpu().numpy().astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _batch(rs, 64, hot), _batch(rs, 3000, hot)
        data = mdl.plan_batch([pos, neg]) if path == "planned" else [pos, neg]
        L = float(mdl.train_step(data)["loss"].item())
        Lo = O.train_step("DistMult", Ec, Ro, oo, _remap(pos, m), _remap(neg, m), "pointwize_logistic", 1.0, 1)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        got = mdl.E.index_select(0, torch.from_numpy(hot).cuda()).cpu().numpy()
        assert_tables(got, Ec)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    mdl._raise_on_flag()
    # rows nobody touched did not move (dense Adam: m = v = 0 there) -- spot check on both sides of the mark
    probe = torch.tensor([1, 12345, 2**31 // D + 7, N - 2], device="cuda")
    probe = probe[~torch.isin(probe, torch.from_numpy(hot).cuda())]
    before = M.DistMult(e_dim=D, r_dim=D, num_entities=N, num_relations=R, dp=0.0, seed=3).E.index_select(0, probe) * 40.0
    assert torch.equal(mdl.E.index_select(0, probe), before)
    # predict on rows at the end of the table
    q = _batch(rs, 500, hot)
    np.testing.assert_allclose(mdl.predict(q), O.call("DistMult", Ec, Ro, _remap(q, m)), atol=TOL)


def test_rank_counts_on_a_table_past_2_31_elements():
    """1-vs-all counts against a candidate range at the END of the big table (rows past the 2^31-element mark), both sides in one
    sweep, against the closed form on those rows."""
    _need_memory()
    from kge_keras_amd import models as M
    rs = np.random.RandomState(6)
    mdl = M.DistMult(e_dim=D, r_dim=D, num_entities=N, num_relations=R, dp=0.0, seed=4)
    mdl.E.mul_(1000.0)  # (entries ~ +-0.7: scores of order 10, neighbouring candidates ~1e-3 apart, fp32 noise ~1e-5)
    mdl.Rel.mul_(10.0)
    lo, rows = N - 5000, 5000
    assert lo * D > 2**31
    nq = 300
    test = np.stack([rs.randint(lo, N, nq), rs.randint(0, R, nq), rs.randint(lo, N, nq)], 1).astype(np.int32)
    cnt = mdl.rank_counts(test, 2, np.zeros(0, np.int32), np.zeros(0, np.int32), cand_offset=lo, cand_rows=rows).cpu().numpy()
    Ec = mdl.E[lo:].cpu().numpy().astype(np.float64)
    Ro = mdl.Rel.cpu().numpy().astype(np.float64)
    loc = test.copy()
    loc[:, 0] -= lo
    loc[:, 2] -= lo
    for side in (0, 1):
        qv = Ec[loc[:, 0]] * Ro[loc[:, 1]] if side == 0 else Ro[loc[:, 1]] * Ec[loc[:, 2]]
        sc = qv @ Ec.T
        tg = sc[np.arange(nq), loc[:, 2] if side == 0 else loc[:, 0]]
        ref = (sc > tg[:, None]).sum(1)
        gap = np.abs(sc - tg[:, None])
        gap[np.arange(nq), loc[:, 2] if side == 0 else loc[:, 0]] = np.inf
        clear = gap.min(1) > 1e-4
        got = cnt[side * nq:(side + 1) * nq]
        assert clear.mean() > 0.8 and np.array_equal(got[clear], ref[clear]) and np.all(np.abs(got - ref) <= 2)
    mdl._raise_on_flag()


@pytest.mark.parametrize("path", ["step", "planned", "pairwise"])
def test_list_offsets_of_a_table_of_many_tiles(path):
    """The scan of the per-entity list lengths of a 2.2 M entity table (2 149 tiles of 1024 keys in 3 groups: tile sums and group sums in
    front of the scan, pull.hpp) with hub entities -- lists of several chunks, i.e. all three scanned quantities non-trivial -- in
    different tiles and groups: two steps against the oracle on the touched rows.  `pairwise`: the generic-list variant (pull.hip,
    NK = N + R keys) through a pairwise loss."""
    from kge_keras_amd import models as M
    n_ent, d, n_rel = 2_200_000, 8, 5
    rs = np.random.RandomState(9)
    loss = "pairwize_hinge" if path == "pairwise" else "pointwize_logistic"
    mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=n_ent, num_relations=n_rel, dp=0.0, seed=2, loss_function=loss)
    mdl.E.mul_(300.0)
    mdl.compile("adam")
    hubs = np.array([5, 1024 * 700 + 3, 1024 * 1100 + 9, 1024 * 2100 + 1, n_ent - 1])  # first tile, groups 0 / 1 / 2, last row
    hot = np.unique(np.concatenate([rs.randint(0, n_ent, 400), hubs])).astype(np.int64)
    Ec, m = _compact(mdl, hot)
    Ro = mdl.Rel.cpu().numpy().astype(np.float64)
    oo = O.AdamTF()

    def batch(n):
        ent = lambda: np.where(rs.rand(n) < 0.5, hubs[rs.randint(0, len(hubs), n)], hot[rs.randint(0, len(hot), n)])  # noqa: E731
        return np.stack([ent(), rs.randint(0, n_rel, n), ent()], 1).astype(np.int32)
    for step in range(2):
        n_neg = 600 if path == "pairwise" else 4000
        pos, neg = batch(64 if path != "pairwise" else n_neg), batch(n_neg)
        data = mdl.plan_batch([pos, neg]) if path == "planned" else [pos, neg]
        L = float(mdl.train_step(data)["loss"].item())
        Lo = O.train_step("DistMult", Ec, Ro, oo, _remap(pos, m), _remap(neg, m), loss, 1.0, 1)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        assert_tables(mdl.E.index_select(0, torch.from_numpy(hot).cuda()).cpu().numpy(), Ec)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    mdl._raise_on_flag()
