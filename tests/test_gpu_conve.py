"""GPU parity of ConvE (models.py:239-275) against the oracle's restatement (conv + batch norm + dense; parity unpinned at
the TensorFlow boundary like the other arithmetic)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)


def _mk(N, R, d, seed=0, **kw):
    from kge_keras_amd.models import ConvE
    rs = np.random.RandomState(seed)
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    mdl = ConvE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, init_entities=E0, init_relations=R0, **kw)
    P = O.conve_init(d, rs, filters=kw.get("conv_filters", 8), kh=kw.get("conv_size_w", 3), kw=kw.get("conv_size_h", 3))
    # non-trivial values everywhere so that every term of the formulas is exercised
    for k in ("bc", "b1", "bd", "b2", "mm1", "mm2"):
        P[k] += rs.uniform(-0.2, 0.2, P[k].shape).astype(np.float32)
    for k in ("g1", "g2", "mv1", "mv2"):
        P[k] *= rs.uniform(0.7, 1.3, P[k].shape).astype(np.float32)
    mdl.set_head_params(P)
    return mdl, E0, R0, P, rs


def _tri(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


def _f64(P):
    return {k: np.asarray(v, dtype=np.float64).copy() for k, v in P.items()}


@pytest.mark.parametrize("d", [12, 32, 200])
def test_conve_predict_parity(d):
    mdl, E0, R0, P, rs = _mk(40, 5, d)
    assert (mdl.w, mdl.h) == O.conve_shape(d)[:2]
    t = _tri(rs, 300, 40, 5)
    got = mdl.predict(t)
    Pd = _f64(P)
    _, ref = O.conve_forward(Pd, E0[t[:, 0]].astype(np.float64), R0[t[:, 1]].astype(np.float64), E0[t[:, 2]].astype(np.float64), False)
    np.testing.assert_allclose(got, ref, atol=TOL, rtol=1e-4)
    x = mdl.hidden(t).cpu().numpy()
    xr, _ = O.conve_forward(Pd, E0[t[:, 0]].astype(np.float64), R0[t[:, 1]].astype(np.float64), E0[t[:, 2]].astype(np.float64), False)
    np.testing.assert_allclose(x, xr, atol=TOL, rtol=1e-4)


@pytest.mark.parametrize("loss", ["pointwize_logistic", "pairwize_hinge"])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("dp,hdp", [(0.0, 0.0), (0.2, 0.3)])
def test_conve_train_step_parity(loss, opt, dp, hdp):
    N, R, d, m = 30, 4, 12, 200
    mdl, E0, R0, P, rs = _mk(N, R, d, seed=3, dp=dp, hidden_dp=hdp, loss_function=loss, margin=0.5, seed_=0)
    mdl._seed = 7
    mdl.compile(opt)
    Eo, Ro, Po = E0.astype(np.float64), R0.astype(np.float64), _f64(P)
    oo = O.make_optimizer(opt)
    for step in range(2):
        pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lo, sp, sn = O.conve_train_step(Po, Eo, Ro, oo, pos, neg, loss, 0.5, 1, dp, hdp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        got = mdl.get_head_params()
        for k in ("K", "g1", "b1", "Wd", "g2", "b2", "mm1", "mv1", "mm2", "mv2"):
            np.testing.assert_allclose(got[k], Po[k], atol=2e-4, rtol=1e-3, err_msg=k)
        # bc and bd feed straight into a batch norm: their true gradient is zero (TensorFlow would apply rounding noise);
        # this implementation leaves them untouched
        np.testing.assert_array_equal(got["bc"], P["bc"])
        np.testing.assert_array_equal(got["bd"], P["bd"])
        Po["bc"][...] = P["bc"]
        Po["bd"][...] = P["bd"]
    assert float(mdl._gtheta.abs().max()) == 0.0 and float(mdl._gE.abs().max()) == 0.0


def test_conve_rank_parity():
    from kge_keras_amd import utils as U
    N, R, d = 60, 3, 12
    mdl, E0, R0, P, rs = _mk(N, R, d, seed=5)
    test = list({(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(40)})
    tes = set(test)
    filt = [t for t in {(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(600)} if t not in tes]
    tails, heads = mdl.rank_triples(np.asarray(test, dtype=np.int32), N, np.asarray(filt, dtype=np.int32))
    Pd = _f64(P)
    fh, ft = O.build_filters(filt)
    E64, R64 = E0.astype(np.float64), R0.astype(np.float64)
    allc = np.arange(N)
    for i, (s, p, o) in enumerate(test):
        _, sc_t = O.conve_forward(Pd, E64[[s] * N], R64[[p] * N], E64[allc], False)
        _, sc_h = O.conve_forward(Pd, E64[allc], R64[[p] * N], E64[[o] * N], False)
        for sc, tgt, fl, got in ((sc_t, o, ft.get((s, p), ()), tails[i]), (sc_h, s, fh.get((p, o), ()), heads[i])):
            keep = np.ones(N, bool)
            keep[tgt] = False
            keep[list(fl)] = False
            ref = 1 + int(np.sum(sc[keep] > sc[tgt]))
            gap = np.min(np.abs(sc[keep] - sc[tgt]))
            assert got == ref or gap < 1e-5, (i, got, ref, gap)
    met = U.validate(mdl, test, N, None, filtering_triples=filt)
    assert len(met) == 15 and 1.0 <= met["mr"] <= N


def test_conve_fit_runs(datasets):
    import random
    from kge_keras_amd.harness import DataGenerator, build_model
    np.random.seed(0)
    random.seed(0)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="ConvE", loss_function="pointwize_logistic",
                           margin=1, seed=2))
    assert mdl.dp == 0.2 and mdl.hidden_dp == 0.2  # reference defaults (models.py:22,241)
    h = mdl.fit(DataGenerator(datasets["nations_train"], ns=10, batch_size=32), epochs=5, verbose=0)
    assert h.history["loss"][-1] < h.history["loss"][0]


# models.py:241-245: conv_filters / conv_size_w / conv_size_h other than the defaults (the general kernels of csrc/conve.hip)
SHAPES = [dict(conv_filters=4, conv_size_w=2, conv_size_h=3), dict(conv_filters=6, conv_size_w=1, conv_size_h=1),
          dict(conv_filters=3, conv_size_w=3, conv_size_h=2), dict(conv_filters=16, conv_size_w=4, conv_size_h=2),
          dict(conv_filters=1, conv_size_w=2, conv_size_h=2)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("d", [12, 40])
def test_conve_general_shape_predict_parity(shape, d):
    mdl, E0, R0, P, rs = _mk(40, 5, d, seed=2, **shape)
    assert P["K"].shape == (shape["conv_size_w"], shape["conv_size_h"], shape["conv_filters"])
    t = _tri(rs, 300, 40, 5)
    Pd = _f64(P)
    xr, ref = O.conve_forward(Pd, E0[t[:, 0]].astype(np.float64), R0[t[:, 1]].astype(np.float64), E0[t[:, 2]].astype(np.float64), False)
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(mdl.hidden(t).cpu().numpy(), xr, atol=TOL, rtol=1e-4)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dp,hdp", [(0.0, 0.0), (0.2, 0.3)])
def test_conve_general_shape_train_step_parity(shape, dp, hdp):
    N, R, d, m = 30, 4, 12, 200
    mdl, E0, R0, P, rs = _mk(N, R, d, seed=3, dp=dp, hidden_dp=hdp, loss_function="pointwize_logistic", margin=0.5, **shape)
    mdl._seed = 7
    mdl.compile("adam")
    Eo, Ro, Po = E0.astype(np.float64), R0.astype(np.float64), _f64(P)
    oo = O.make_optimizer("adam")
    for step in range(2):
        pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lo, sp, sn = O.conve_train_step(Po, Eo, Ro, oo, pos, neg, "pointwize_logistic", 0.5, 1, dp, hdp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        got = mdl.get_head_params()
        for k in ("K", "g1", "b1", "Wd", "g2", "b2", "mm1", "mv1", "mm2", "mv2"):
            np.testing.assert_allclose(got[k], Po[k], atol=2e-4, rtol=1e-3, err_msg=k)
        Po["bc"][...] = P["bc"]
        Po["bd"][...] = P["bd"]
    mdl._raise_on_flag()


def test_conve_kernel_larger_than_image_is_a_shape_error():
    from kge_keras_amd.models import ConvE
    with pytest.raises(ValueError):  # 2w x h = 6 x 4 image, 7 rows of kernel: the reference's Conv2D raises at build time
        ConvE(e_dim=12, r_dim=12, num_entities=5, num_relations=2, conv_size_w=7)


@pytest.mark.parametrize("loss,act", [("pointwize_logistic", "tanh"), ("pairwize_hinge", "relu")])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("dp,hdp", [(0.0, 0.0), (0.2, 0.3)])
def test_literal_path_under_conve(loss, act, opt, dp, hdp):
    """models.py:152-158 is model agnostic: the literal transform (Dense on [entity row ; literal row], the same layer for s and o) in
    front of ConvE.func (:264-275).  The library runs the head on the batch's transformed rows as staged tables (kge_lit_conve_train_grads);
    two train steps -- scores, loss, both tables, the head's parameters and moving statistics, the Dense -- against the oracle's
    composition of literal_transform and conve_forward / conve_backward, with and without the two dropouts; then predict and the
    filtered ranks on the transformed table."""
    N, R, d, L, m = 30, 4, 12, 5, 200
    from kge_keras_amd.models import ConvE
    rs = np.random.RandomState(11)
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    lit = rs.rand(N, L).astype(np.float32)
    mdl = ConvE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, init_entities=E0, init_relations=R0, literals=lit, literal_activation=act, dp=dp,
                hidden_dp=hdp, loss_function=loss, margin=0.5)
    P = O.conve_init(d, rs)
    for k in ("bc", "b1", "bd", "b2", "mm1", "mm2"):
        P[k] += rs.uniform(-0.2, 0.2, P[k].shape).astype(np.float32)
    for k in ("g1", "g2", "mv1", "mv2"):
        P[k] *= rs.uniform(0.7, 1.3, P[k].shape).astype(np.float32)
    mdl.set_head_params(P)
    H = O.literal_init(d, L, rs)
    H["b"] += rs.uniform(-0.1, 0.1, d).astype(np.float32)
    mdl.set_literal_dense(H["W"], H["b"])
    mdl._seed = 7
    mdl.compile(opt)
    Eo, Ro, Po, Ho, Lo_ = E0.astype(np.float64), R0.astype(np.float64), _f64(P), _f64(H), lit.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(2):
        pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        Lg = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lr, sp, sn = O.literal_conve_train_step(Po, Ho, act, Eo, Ro, Lo_, oo, pos, neg, loss, 0.5, 1, dp, hdp, seed=mdl._seed)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-4)
        assert Lg == pytest.approx(Lr, rel=5e-6, abs=TOL_LA), (step, Lg, Lr)
        assert_tables(mdl.E.cpu().numpy(), Eo, atol=5e-6)   # (TOL_W_LONG's class: a Dense layer and a batch norm between the rows and the loss)
        assert_tables(mdl.Rel.cpu().numpy(), Ro, atol=5e-6)
        W, b = mdl.get_literal_dense()
        assert_tables(W, Ho["W"], atol=5e-6)
        assert_tables(b, Ho["b"], atol=5e-6)
        got = mdl.get_head_params()
        for k in ("K", "g1", "b1", "Wd", "g2", "b2", "mm1", "mv1", "mm2", "mv2"):
            np.testing.assert_allclose(got[k], Po[k], atol=2e-4, rtol=1e-3, err_msg=k)
        Po["bc"][...] = P["bc"]  # (their true gradient is zero: see test_conve_train_step_parity)
        Po["bd"][...] = P["bd"]
    assert float(mdl._gtheta.abs().max()) == 0.0 and float(mdl._gE.abs().max()) == 0.0 and float(mdl._lit["gtheta"].abs().max()) == 0.0
    # inference on the transformed table
    t = _tri(rs, 100, N, R)
    Et, _ = O.literal_transform(Ho, Eo, Lo_, np.arange(N), act)
    _, ref = O.conve_forward(Po, Et[t[:, 0]], Ro[t[:, 1]], Et[t[:, 2]], False)
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-4)
    test = list({(int(rs.randint(N)), int(rs.randint(R)), int(rs.randint(N))) for _ in range(20)})
    tails, heads = mdl.rank_triples(np.asarray(test, dtype=np.int32), N, np.zeros((0, 3), dtype=np.int32))
    allc = np.arange(N)
    for i, (s, p, o) in enumerate(test):
        _, sc_t = O.conve_forward(Po, Et[[s] * N], Ro[[p] * N], Et[allc], False)
        _, sc_h = O.conve_forward(Po, Et[allc], Ro[[p] * N], Et[[o] * N], False)
        for scv, tgt, got_r in ((sc_t, o, tails[i]), (sc_h, s, heads[i])):
            keep = np.ones(N, bool)
            keep[tgt] = False
            refr = 1 + int(np.sum(scv[keep] > scv[tgt]))
            gap = np.min(np.abs(scv[keep] - scv[tgt]))
            assert got_r == refr or gap < 1e-5, (i, got_r, refr, gap)
    mdl._raise_on_flag()
