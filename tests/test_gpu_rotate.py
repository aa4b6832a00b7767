"""RotatE exactly as the reference wrote it (KGEkeras/models.py:421-459): entity rows doubled, relation rows are phases, the rotated
subject minus the object summed WITH its signs -- against the oracle's restatement: scores, train steps (every loss, SGD / Adam,
dropout), filtered ranks, and the reference workflow through the harness."""
import random

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)
LOSSES = ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss", "pairwize_hinge", "pairwize_logistic", "pairwize_square_loss"]


def _mk(N, R, d, gamma=12, amp=0.5, seed=5, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    E0 = rs.uniform(-amp, amp, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-4.0, 4.0, (R, d)).astype(np.float32)  # phases beyond (-pi, pi]: the wrap of models.py:443 is exercised
    kw.setdefault("dp", 0.0)
    mdl = M.RotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, gamma=gamma, init_entities=E0, init_relations=R0, **kw)
    assert mdl.e_dim == 2 * d and mdl.r_dim == d and tuple(mdl.Rel.shape) == (R, d)
    return mdl, E0, R0, rs


def _triples(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


@pytest.mark.parametrize("d,n", [(16, 1), (100, 63), (15, 5000), (3, 77), (200, 300)])
@pytest.mark.parametrize("gamma", [12, 0])
def test_rotate_score_parity(d, n, gamma):
    mdl, E0, R0, rs = _mk(97, 11, d, gamma)
    t = _triples(rs, n, 97, 11)
    ref = O.call("RotatE", E0, R0, t, 2, gamma)
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-5)
    # func on already-gathered rows (models.py:160)
    s, p, o = E0[t[:, 0]], R0[t[:, 1]], E0[t[:, 2]]
    np.testing.assert_allclose(mdl.func(s, p, o).cpu().numpy(), ref, atol=TOL, rtol=1e-5)


def test_rotate_shape_errors():
    from kge_keras_amd import _lib
    import ctypes
    md = _lib.ModelDesc(_lib.MODEL_IDS["RotatE"], 10, 3, 8, 8, 1, 12.0, 0.0, 0, 0.0)  # r_dim must be e_dim / 2
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None) == _lib.KGE_ESHAPE


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("gamma", [12, 0])
def test_rotate_train_step_parity(loss, opt, gamma):
    N, R, d, m = 40, 6, 16, 300
    mdl, E0, R0, rs = _mk(N, R, d, gamma, loss_function=loss, margin=0.5)
    mdl.compile(opt)
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        sp, sn = O.call("RotatE", Eo, Ro, pos, 2, gamma), O.call("RotatE", Eo, Ro, neg, 2, gamma)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
        Lo = O.train_step("RotatE", Eo, Ro, oo, pos, neg, loss, 0.5, 1, 2, gamma)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0


def test_rotate_dropout_step_and_bad_ids():
    N, R, d, m = 30, 5, 12, 200
    mdl, E0, R0, rs = _mk(N, R, d, 12, dp=0.3, loss_function="pointwize_logistic", seed=8)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("RotatE", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1, 1, 2, 12, dropout=0.3, seed=mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    bad = torch.tensor([[0, 0, 1], [0, R, 1], [N, 0, 1]], dtype=torch.int32, device="cuda")
    sc = mdl.call(bad).cpu().numpy()
    assert np.isfinite(sc[0]) and np.isnan(sc[1]) and np.isnan(sc[2])
    with pytest.raises(IndexError):
        mdl._raise_on_flag()


@pytest.mark.parametrize("gamma", [12, 0])
def test_rotate_ranks_vs_oracle(gamma):
    """1-vs-all as a dot-product sweep (the score is linear in the candidate row; tails: every query ranks the candidates by their row
    sums).  Ranks identical to the oracle's wherever no candidate lies within 1e-4 of the target's score (the oracle's fp64 sum of
    (rotated s - o) terms and the kernels' fp32 <q, e> round differently, so knife-edge ties are left aside)."""
    from kge_keras_amd import models as M
    N, R, d = 300, 5, 16
    rs = np.random.RandomState(3)
    test = [tuple(int(v) for v in t) for t in _triples(rs, 60, N, R)]
    test = list(dict.fromkeys(test))
    filt = [tuple(int(v) for v in t) for t in _triples(rs, 400, N, R)]
    filt = [t for t in filt if t not in set(test)]
    fh, ft = O.build_filters(filt)
    ft, fh = {k: sorted(v) for k, v in ft.items()}, {k: sorted(v) for k, v in fh.items()}
    E0 = rs.uniform(-1, 1, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-4, 4, (R, d)).astype(np.float32)
    mdl = M.RotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, gamma=gamma, dp=0.0, init_entities=E0, init_relations=R0)
    tails, heads = mdl.rank(test, N, ft, fh)
    margins = []
    to, ho = O.ranks_closed_form("RotatE", E0, R0, test, filt, 2, gamma, margin_out=margins)
    m = np.asarray(margins).reshape(-1, 2)
    safe_t, safe_h = m[:, 0] > 1e-4, m[:, 1] > 1e-4
    assert np.array_equal(tails[safe_t], to[safe_t]) and np.array_equal(heads[safe_h], ho[safe_h])
    assert safe_t.mean() > 0.8 and safe_h.mean() > 0.8
    # every tail query ranks the candidates by the same key (the row sum): unfiltered tail ranks of one target agree across queries
    same_o = [i for i, t in enumerate(test) if t[2] == test[0][2]]
    t2, _ = mdl.rank([test[i] for i in same_o], N, {}, {})
    assert len(set(t2.tolist())) == 1


def test_rotate_through_the_harness(datasets):
    """build_model -> fit(DataGenerator) -> validate, as the reference's script does for any model name (tests.py:63-134)."""
    from kge_keras_amd.harness import DataGenerator, build_model
    from kge_keras_amd.utils import validate
    tr, te = ([tuple(int(v) for v in t) for t in datasets["nations_" + k].tolist()] for k in ("train", "test"))
    np.random.seed(0)
    random.seed(0)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=16, embedding_model="RotatE", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=1))
    assert not mdl.plan_supported()
    h = mdl.fit(DataGenerator(tr, ns=10, batch_size=64), epochs=5, verbose=0)
    assert len(h.history["loss"]) == 5 and h.history["loss"][-1] < h.history["loss"][0]
    tes = set(te)
    res = validate(mdl, te, 14, None, filtering_triples=[t for t in tr if t not in tes])
    assert 1.0 <= res["mr"] <= 14.0 and 0.0 < res["mrr"] <= 1.0
