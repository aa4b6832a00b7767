"""GPU parity of the LiteralE-style input path (models.py:114-120,152-158 -- the configuration of the reference's own script:
DistMult + random literals + tanh + pairwize_hinge, tests.py:113-128)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, TOL_W_LONG, TOL_L_BIG, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)


def _mk(model, N, R, d, L, act, seed=0, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    lit = rs.rand(N, L).astype(np.float32)  # tests.py:113 np.random.rand(len(E), 5)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, init_entities=E0, init_relations=R0, literals=lit,
                            literal_activation=act, **kw)
    H = O.literal_init(de, L, rs)
    H["b"] += rs.uniform(-0.1, 0.1, de).astype(np.float32)
    mdl.set_literal_dense(H["W"], H["b"])
    return mdl, E0, R0, lit, H, rs


def _tri(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


def _f64(H):
    return {k: np.asarray(v, dtype=np.float64).copy() for k, v in H.items()}


@pytest.mark.parametrize("act", list(O.ACTIVATIONS))
def test_literal_predict_parity(act):
    mdl, E0, R0, lit, H, rs = _mk("DistMult", 40, 5, 32, 5, act, dp=0.0)
    t = _tri(rs, 300, 40, 5)
    Hd = _f64(H)
    s_, _ = O.literal_transform(Hd, E0.astype(np.float64), lit.astype(np.float64), t[:, 0], act)
    o_, _ = O.literal_transform(Hd, E0.astype(np.float64), lit.astype(np.float64), t[:, 2], act)
    ref = O.score_rows("DistMult", s_, R0[t[:, 1]].astype(np.float64), o_)
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-4)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=1, gamma=12)), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {}),
                                      ("HolE", {})])
@pytest.mark.parametrize("loss,act", [("pairwize_hinge", "tanh"), ("pointwize_logistic", "relu"), ("pointwize_hinge", "sigmoid"), ("pointwize_logistic", "softmax")])
@pytest.mark.parametrize("dp", [0.0, 0.25])
def test_literal_train_step_parity(model, kw, loss, act, dp):
    N, R, d, L, m = 30, 4, 16, 5, 200
    mdl, E0, R0, lit, H, rs = _mk(model, N, R, d, L, act, seed=2, dp=dp, loss_function=loss, margin=0.5, **kw)
    mdl.compile("adam")
    Eo, Ro, Ho, Lo = E0.astype(np.float64), R0.astype(np.float64), _f64(H), lit.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L_ = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lr, sp, sn = O.literal_train_step(model, Ho, act, Eo, Ro, Lo, oo, pos, neg, loss, 0.5, 1, kw.get("norm", 1), kw.get("gamma", 12),
                                          dropout=dp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert L_ == pytest.approx(Lr, rel=TOL_L, abs=TOL_LA), (step, L_, Lr)
        np.testing.assert_allclose(mdl.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
        np.testing.assert_allclose(mdl.Rel.cpu().numpy(), Ro, atol=TOL_W_LONG)
        W, b = mdl.get_literal_dense()
        np.testing.assert_allclose(W, Ho["W"], atol=TOL_W_LONG)
        np.testing.assert_allclose(b, Ho["b"], atol=TOL_W_LONG)


@pytest.mark.parametrize("model,kw", [("pRotatE", {}), ("pRotatE", dict(gamma=0)), ("TransE", dict(norm=3.5, gamma=12)), ("TransE", dict(norm=1.5, gamma=0)),
                                      ("RotatE", {}), ("RotatE", dict(gamma=0))])
@pytest.mark.parametrize("dp", [0.0, 0.25])
def test_literal_under_protate_and_general_p_norm(model, kw, dp):
    """models.py:152-158 is model agnostic: the literal transform in front of pRotatE's phase distance (:479-489), of TransE with any
    positive real tf.norm order (:188) and of RotatE (:437-459: entity rows -- and the Dense -- 2 e_dim wide, relation rows e_dim phases)
    -- two steps (scores, loss, both tables, the Dense) against the oracle, with and without dropout; predict on the transformed table;
    RotatE also the filtered 1-vs-all ranks."""
    N, R, d, L, m = 30, 4, 16, 5, 200
    mdl, E0, R0, lit, H, rs = _mk(model, N, R, d, L, "tanh", seed=4, dp=dp, loss_function="pointwize_logistic", margin=0.5, **kw)
    mdl.compile("adam")
    norm = mdl.modulus if model == "pRotatE" else kw.get("norm", 1)
    gamma = kw.get("gamma", 12)
    Eo, Ro, Ho, Lo = E0.astype(np.float64), R0.astype(np.float64), _f64(H), lit.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L_ = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lr, sp, sn = O.literal_train_step(model, Ho, "tanh", Eo, Ro, Lo, oo, pos, neg, "pointwize_logistic", 0.5, 1, norm, gamma, dropout=dp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert L_ == pytest.approx(Lr, rel=TOL_L, abs=TOL_LA), (step, L_, Lr)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        W, b = mdl.get_literal_dense()
        assert_tables(W, Ho["W"])
        assert_tables(b, Ho["b"])
    t = _tri(rs, 100, N, R)
    s_, _ = O.literal_transform(Ho, Eo, Lo, t[:, 0], "tanh")
    o_, _ = O.literal_transform(Ho, Eo, Lo, t[:, 2], "tanh")
    np.testing.assert_allclose(mdl.predict(t), O.score_rows(model, s_, Ro[t[:, 1]], o_, norm, gamma), atol=TOL, rtol=1e-4)
    if model == "RotatE":
        Et, _ = O.literal_transform(Ho, Eo, Lo, np.arange(N), "tanh")
        test = [tuple(int(v) for v in x) for x in t[:30]]
        filt = [f for f in (tuple(int(v) for v in x) for x in _tri(rs, 100, N, R)) if f not in set(test)]
        tr, hr = O.ranks_closed_form("RotatE", Et, Ro, test, filt, norm, gamma)
        fh, ft = O.build_filters(filt)
        gt, gh = mdl.rank(test, N, ft, fh)
        assert np.mean(gt == tr) > 0.95 and np.mean(gh == hr) > 0.95 and np.max(np.abs(gt - tr)) <= 1 and np.max(np.abs(gh - hr)) <= 1
    mdl._raise_on_flag()


def test_literal_weight_gradient_split_contraction():
    """Batches large enough for the Dense layer's weight gradient to be computed as slabs of the contraction over the stacked
    rows (strided-batched GEMM + ordered sum, plus a tail GEMM for the rows that do not fill a slab): 2n = 8310 rows = 4 slabs of 2077 + a tail of 2."""
    N, R, d, L = 50, 4, 16, 5
    mdl, E0, R0, lit, H, rs = _mk("DistMult", N, R, d, L, "tanh", seed=3, dp=0.0, loss_function="pointwize_logistic", margin=0.5)
    mdl.compile("adam")
    Eo, Ro, Ho, Lo = E0.astype(np.float64), R0.astype(np.float64), _f64(H), lit.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _tri(rs, 2100, N, R), _tri(rs, 2055, N, R)
        L_ = float(mdl.train_step([pos, neg])["loss"].item())
        Lr, _, _ = O.literal_train_step("DistMult", Ho, "tanh", Eo, Ro, Lo, oo, pos, neg, "pointwize_logistic", 0.5, 1)
        assert L_ == pytest.approx(Lr, rel=TOL_L, abs=TOL_LA), step
        W, b = mdl.get_literal_dense()
        assert_tables(W, Ho["W"])
        assert_tables(b, Ho["b"])
        assert_tables(mdl.E.cpu().numpy(), Eo)


def test_reference_script_configuration(datasets):
    """tests.py:96-134 on nations: DistMult, dim from hp, random literals (len(E), 5), tanh, pairwize_hinge, margin 1, Adam;
    fit runs, the loss falls, validate() ranks on the literal-transformed table."""
    import random
    from kge_keras_amd.harness import DataGenerator, build_model
    from kge_keras_amd.utils import validate
    tr = [tuple(int(v) for v in t) for t in datasets["nations_train"]]
    te = [tuple(int(v) for v in t) for t in datasets["nations_test"]]
    np.random.seed(0)
    random.seed(0)
    literals = np.random.rand(14, 5)
    mdl = build_model({'num_entities': 14, 'num_relations': 55, 'dim': 32, 'embedding_model': 'DistMult', 'literals': literals,
                       'literal_activation': 'tanh', 'loss_function': 'pairwize_hinge', 'margin': 1, 'seed': 1})
    h = mdl.fit(DataGenerator(tr, batch_size=32), epochs=6, verbose=0)
    assert h.history["loss"][-1] < h.history["loss"][0]
    tes = set(te)
    met = validate(mdl, te[:60], 14, None, filtering_triples=[t for t in tr if t not in tes])
    assert len(met) == 15 and 1.0 <= met["mr"] <= 14


def test_reference_training_script_flow(datasets):
    """harness.main == tests.py:96-134 (id maps, random literals behind a tanh Dense, DistMult, pairwize_hinge with the
    harness's all-pairs broadcast, Adam, fit with a validation split) on the nations triples: it runs and learns."""
    from kge_keras_amd.harness import main

    def strs(a):
        return [(f"e{s}", f"r{p}", f"e{o}") for s, p, o in a.tolist()]
    np.random.seed(1)
    import random
    random.seed(2)
    model, h = main(strs(datasets["nations_train"]), strs(datasets["nations_valid"]), strs(datasets["nations_test"]),
                    dim=16, bs=64, epochs=4, verbose=0)
    assert model.num_entities == 14 and len(h.history["loss"]) == 4 and len(h.history["val_loss"]) == 4
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert np.all(np.isfinite(h.history["val_loss"]))


def test_literal_under_hole_predict_and_rank():
    """models.py:152-158 is model agnostic: HolE on literal-transformed rows -- predict() and the filtered 1-vs-all ranks on the
    transformed table equal the oracle's."""
    N, R, d, L = 40, 5, 24, 5
    mdl, E0, R0, lit, H, rs = _mk("HolE", N, R, d, L, "tanh", seed=4, dp=0.0)
    t = _tri(rs, 200, N, R)
    Hd = _f64(H)
    ids = np.arange(N)
    Et, _ = O.literal_transform(Hd, E0.astype(np.float64), lit.astype(np.float64), ids, "tanh")
    ref = O.score_rows("HolE", Et[t[:, 0]], R0[t[:, 1]].astype(np.float64), Et[t[:, 2]])
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-4)
    test = [tuple(int(v) for v in x) for x in t[:30]]
    filt = [tuple(int(v) for v in x) for x in _tri(rs, 100, N, R)]
    filt = [f for f in filt if f not in set(test)]
    tr, hr = O.ranks_closed_form("HolE", Et, R0.astype(np.float64), test, filt)
    fh, ft = O.build_filters(filt)
    gt, gh = mdl.rank(test, N, ft, fh)
    assert np.mean(gt == tr) > 0.95 and np.mean(gh == hr) > 0.95 and np.max(np.abs(gt - tr)) <= 1 and np.max(np.abs(gh - hr)) <= 1
