"""HAKE exactly as the reference wrote it (KGEkeras/models.py:362-400): entity rows [phase | modulus], relation rows
[phase | modulus | bias], and a modulus term that is ONE tf.norm over every row handed to a call of the model -- so scores are compared
call by call: model(x) (all rows), model.predict(x) (batches of 32 rows), the two sides of a training batch, and the candidate lists
of `validate`.  Against the oracle's restatement (itself checked against autograd and a hand-computed value in tests/test_oracle.py)."""
import random

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, TOL_W_LONG, TOL_L_BIG, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)
LOSSES = ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss", "pairwize_hinge", "pairwize_logistic", "pairwize_square_loss"]


def _mk(N, R, h, gamma=12, seed=5, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    rng = (gamma + 2) / (2 * h) / 2
    E0 = np.concatenate([rs.uniform(-rng, rng, (N, h)), rs.uniform(-1.0, 1.0, (N, h))], 1).astype(np.float32)
    # bias column on both sides of 1 and of -|mod_p|: the clip (models.py:395) and the where (:396) are both exercised
    R0 = np.concatenate([rs.uniform(-rng, rng, (R, h)), rs.uniform(-1.0, 1.0, (R, h)), rs.uniform(-1.5, 1.5, (R, h))], 1).astype(np.float32)
    kw.setdefault("dp", 0.0)
    mdl = M.HAKE(e_dim=h, r_dim=h, num_entities=N, num_relations=R, gamma=gamma, init_entities=E0, init_relations=R0, **kw)
    assert mdl.e_dim == 2 * h and mdl.r_dim == 3 * h and tuple(mdl.Rel.shape) == (R, 3 * h)
    prm = O.hake_params(mdl.e_dim, gamma, 2, mdl.phase_weight, mdl.mod_weight)
    assert prm[2] == pytest.approx(mdl.embedding_range)
    return mdl, E0, R0, rs, prm


def _triples(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


@pytest.mark.parametrize("h,n", [(16, 1), (100, 63), (15, 5000), (3, 77), (200, 300), (8, 32), (8, 64), (8, 65)])
def test_hake_call_is_one_norm_and_predict_is_one_per_32_rows(h, n):
    mdl, E0, R0, rs, prm = _mk(97, 11, h)
    t = _triples(rs, n, 97, 11)
    whole = O.call("HAKE", E0, R0, t, prm, 12)
    np.testing.assert_allclose(mdl(t).cpu().numpy(), whole, atol=TOL, rtol=1e-5)
    np.testing.assert_allclose(mdl.predict(t), O.call("HAKE", E0, R0, t, prm, 12, chunk=32), atol=TOL, rtol=1e-5)
    np.testing.assert_allclose(mdl.predict(t, batch_size=7), O.call("HAKE", E0, R0, t, prm, 12, chunk=7), atol=TOL, rtol=1e-5)
    s, p, o = E0[t[:, 0]], R0[t[:, 1]], E0[t[:, 2]]
    np.testing.assert_allclose(mdl.func(s, p, o).cpu().numpy(), whole, atol=TOL, rtol=1e-5)
    if n > 32:  # the score of a row is NOT a function of the row alone: the reason this model has entry points of its own
        assert np.abs(mdl.predict(t) - whole).max() > 1e-3


def test_hake_weights_other_than_default_and_errors():
    from kge_keras_amd import _lib, models as M
    import ctypes
    mdl, E0, R0, rs, _ = _mk(40, 5, 12, gamma=6, phase_weight=1.25, mod_weight=0.3)
    prm = O.hake_params(24, 6, 2, 1.25, 0.3)
    t = _triples(rs, 100, 40, 5)
    np.testing.assert_allclose(mdl(t).cpu().numpy(), O.call("HAKE", E0, R0, t, prm, 6), atol=TOL, rtol=1e-5)
    md = _lib.ModelDesc(_lib.MODEL_IDS["HAKE"], 10, 3, 8, 8, 1, 12.0, 0.0, 0, 0.0)  # r_dim must be 3 * e_dim / 2
    hp = _lib.HakeParams(0.5, 1.0, 0.1)
    assert _lib.lib.kge_hake_score(ctypes.byref(md), ctypes.byref(hp), 1, 1, 1, 4, 0, None, 0, 1, 1, 1 << 20, 1, None) == _lib.KGE_ESHAPE
    md = mdl._model_desc(False)
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None) == _lib.KGE_EUNSUPPORTED  # the per-row entry point refuses it
    with pytest.raises(NotImplementedError):
        mdl.rank_counts(t, 0, None, None)
    with pytest.raises(KeyError):
        M.HAKE(num_entities=3, num_relations=2)
    bad = torch.tensor([[0, 0, 1], [0, 5, 1], [40, 0, 1]], dtype=torch.int32, device="cuda")
    out = mdl(bad).cpu().numpy()
    assert np.isnan(out[1]) and np.isnan(out[2]) and np.isfinite(out[0])
    with pytest.raises(IndexError):
        mdl._raise_on_flag()


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_hake_train_step_parity(loss, opt):
    N, R, h, m = 40, 6, 16, 300
    mdl, E0, R0, rs, prm = _mk(N, R, h, loss_function=loss, margin=0.5)
    mdl.compile(opt)
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        sp, sn = O.call("HAKE", Eo, Ro, pos, prm, 12), O.call("HAKE", Eo, Ro, neg, prm, 12)  # one norm per side (tests.py:77-78)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
        Lo = O.train_step("HAKE", Eo, Ro, oo, pos, neg, loss, 0.5, 1, prm, 12)
        assert L == pytest.approx(Lo, rel=TOL_L_BIG, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0


def test_hake_dropout_regulariser_and_unequal_sides():
    N, R, h = 30, 5, 12
    mdl, E0, R0, rs, prm = _mk(N, R, h, dp=0.3, loss_function="pointwize_logistic", regularization=1e-3, seed=8)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _triples(rs, 120, N, R), _triples(rs, 200, N, R)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("HAKE", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1, 1, prm, 12, regularization=1e-3, dropout=0.3, seed=mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    # val_loss: no dropout, no update
    before = mdl.E.clone()
    pos, neg = _triples(rs, 50, N, R), _triples(rs, 50, N, R)
    Lv = float(mdl.batch_loss([pos, neg]).item())
    Lr, *_ = O.dense_grads("HAKE", Eo, Ro, pos, neg, "pointwize_logistic", 1, 1, prm, 12, 1e-3, 0.0, 0)
    assert Lv == pytest.approx(float(Lr), rel=TOL_L, abs=TOL_LA)
    assert torch.equal(before, mdl.E) and float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0  # (forward only: nothing accumulated)


def test_hake_validate_is_the_references_candidate_lists_predicted_32_rows_at_a_time():
    """utils.py:199-264 on HAKE: per query the candidate list (target first, kept entities ascending), predicted in batches of 32 -- the
    modulus norm of each batch shifts the scores of its 32 rows together, so ranks differ from a per-row scorer's.  Ranks equal to the
    literal loop over the oracle's predict wherever no candidate lies within 1e-4 of the target."""
    from kge_keras_amd.utils import validate
    N, R, h = 150, 4, 8
    mdl, E0, R0, rs, prm = _mk(N, R, h, seed=3)
    test = list(dict.fromkeys(tuple(int(v) for v in t) for t in _triples(rs, 40, N, R)))
    filt = [tuple(int(v) for v in t) for t in _triples(rs, 300, N, R)]
    filt = [t for t in filt if t not in set(test)]
    fh, ft = O.build_filters(filt)
    gaps = {"tail": [], "head": []}
    ranks = {"tail": [], "head": []}
    for t in test:
        for side, f in (("tail", ft), ("head", fh)):
            sc = O.call("HAKE", E0, R0, O.candidate_rows(t, N, side, f), prm, 12, chunk=32)
            ranks[side].append(O.rank_literal(sc))
            gaps[side].append(np.abs(sc[1:] - sc[0]).min())
    ftl, fhl = {k: sorted(v) for k, v in ft.items()}, {k: sorted(v) for k, v in fh.items()}
    tails, heads = mdl.rank(test, N, ftl, fhl)
    tails2, heads2 = mdl.rank_triples(np.asarray(test), N, np.asarray(filt))
    assert np.array_equal(tails, tails2) and np.array_equal(heads, heads2)
    st, sh = np.asarray(gaps["tail"]) > 1e-4, np.asarray(gaps["head"]) > 1e-4
    assert st.mean() > 0.8 and sh.mean() > 0.8
    assert np.array_equal(tails[st], np.asarray(ranks["tail"])[st]) and np.array_equal(heads[sh], np.asarray(ranks["head"])[sh])
    # chunking of the test set does not change anything (lists are scored query by query)
    t3, h3 = mdl.rank(test, N, ftl, fhl, chunk=7)
    assert np.array_equal(t3, tails) and np.array_equal(h3, heads)
    res = validate(mdl, test, N, None, filtering_triples=filt)
    assert res["mr"] == pytest.approx((tails.mean() + heads.mean()) / 2)
    with pytest.raises(ValueError):  # list.remove(x): a filtered id that is the target
        mdl.rank([test[0]], N, {(test[0][0], test[0][1]): [test[0][2]]}, {})


def test_hake_through_the_harness(datasets):
    """build_model -> fit(DataGenerator) -> validate, as the reference's script does for any model name (tests.py:63-134)."""
    from kge_keras_amd.harness import DataGenerator, build_model
    from kge_keras_amd.utils import validate
    tr, te = ([tuple(int(v) for v in t) for t in datasets["nations_" + k].tolist()] for k in ("train", "test"))
    np.random.seed(0)
    random.seed(0)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=16, embedding_model="HAKE", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=1))
    assert not mdl.plan_supported() and tuple(mdl.E.shape) == (14, 32) and tuple(mdl.Rel.shape) == (55, 48)
    h = mdl.fit(DataGenerator(tr, ns=10, batch_size=64), epochs=5, verbose=0)
    assert len(h.history["loss"]) == 5 and h.history["loss"][-1] < h.history["loss"][0]
    tes = set(te)
    res = validate(mdl, te, 14, None, filtering_triples=[t for t in tr if t not in tes])
    assert 1.0 <= res["mr"] <= 14.0 and 0.0 < res["mrr"] <= 1.0


@pytest.mark.parametrize("loss,act", [("pointwize_logistic", "tanh"), ("pairwize_hinge", "sigmoid")])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("dp", [0.0, 0.25])
def test_literal_path_under_hake(loss, act, opt, dp):
    """models.py:152-158 is model agnostic: the literal transform (Dense on [entity row ; literal row], 2h wide like the entity rows) in
    front of HAKE.func (:383-400).  The library runs the model on the batch's transformed rows as staged tables, one call per side
    (kge_lit_hake_train_grads): the call-wide norm is the side's; two train steps -- scores, loss, both tables, the Dense -- against the
    oracle's composition, with and without dropout; then model(x) / predict on the transformed table."""
    N, R, h, L, m = 30, 4, 8, 5, 150
    from kge_keras_amd import models as M
    rs = np.random.RandomState(17)
    rng = (12 + 2) / (2 * h) / 2
    E0 = np.concatenate([rs.uniform(-rng, rng, (N, h)), rs.uniform(-1.0, 1.0, (N, h))], 1).astype(np.float32)
    R0 = np.concatenate([rs.uniform(-rng, rng, (R, h)), rs.uniform(-1.0, 1.0, (R, h)), rs.uniform(-1.5, 1.5, (R, h))], 1).astype(np.float32)
    lit = rs.rand(N, L).astype(np.float32)
    mdl = M.HAKE(e_dim=h, r_dim=h, num_entities=N, num_relations=R, gamma=12, init_entities=E0, init_relations=R0, literals=lit, literal_activation=act,
                 dp=dp, loss_function=loss, margin=0.5)
    prm = O.hake_params(mdl.e_dim, 12, 2, mdl.phase_weight, mdl.mod_weight)
    H = O.literal_init(2 * h, L, rs)
    H["b"] += rs.uniform(-0.1, 0.1, 2 * h).astype(np.float32)
    mdl.set_literal_dense(H["W"], H["b"])
    mdl.compile(opt)
    Eo, Ro, Lo_ = E0.astype(np.float64), R0.astype(np.float64), lit.astype(np.float64)
    Ho = {k: np.asarray(v, dtype=np.float64).copy() for k, v in H.items()}
    oo = O.make_optimizer(opt)
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m + 30, N, R)
        sc = torch.empty(2 * m + 30, dtype=torch.float32, device="cuda")
        Lg = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lr, sp, sn = O.literal_train_step("HAKE", Ho, act, Eo, Ro, Lo_, oo, pos, neg, loss, 0.5, 1, prm, 12, dropout=dp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert Lg == pytest.approx(Lr, rel=TOL_L_BIG, abs=TOL_LA), (step, Lg, Lr)
        assert_tables(mdl.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
        assert_tables(mdl.Rel.cpu().numpy(), Ro, atol=TOL_W_LONG)
        W, b = mdl.get_literal_dense()
        assert_tables(W, Ho["W"], atol=TOL_W_LONG)
        assert_tables(b, Ho["b"], atol=TOL_W_LONG)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0 and float(mdl._lit["gtheta"].abs().max()) == 0.0
    t = _triples(rs, 100, N, R)
    Et, _ = O.literal_transform(Ho, Eo, Lo_, np.arange(N), act)
    np.testing.assert_allclose(mdl(t).cpu().numpy(), O.call("HAKE", Et, Ro, t, prm, 12), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(mdl.predict(t), O.call("HAKE", Et, Ro, t, prm, 12, chunk=32), atol=TOL, rtol=1e-4)
    mdl._raise_on_flag()
