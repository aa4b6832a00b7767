"""GPU parity tests (run with -m gpu on the MI355X): every call goes through the C ABI of libkge_hip.so and is
compared with the oracle on identical inputs.  Tolerance: 1e-4 absolute on fp32 scores / losses / weights
(BASELINE.json north_star); index work is bit-exact and is covered by the CPU suite."""
import random

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, TOL_W_LONG, TOL_L_BIG, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)


@pytest.fixture(params=["pull", "push"], autouse=True)
def grad_mode(request):
    """Every training test runs on both gradient paths: per-destination gather-reduce (default) and float atomics."""
    from kge_keras_amd import _lib
    _lib.check(_lib.lib.kge_set_grad_mode(0 if request.param == "pull" else 1))
    yield request.param
    _lib.lib.kge_set_grad_mode(0)


def _mk(model, N, R, d, seed=0, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    cls = getattr(M, model)
    mdl = cls(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, **kw)
    return mdl, E0, R0, rs


def _triples(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


CASES = [("DistMult", {}), ("TransE", dict(norm=1, gamma=12)), ("TransE", dict(norm=2, gamma=12)),
         ("TransE", dict(norm=2, gamma=0)), ("ComplEx", {}), ("HolE", {}), ("HolE", dict(spectral=False))]


@pytest.mark.parametrize("model,kw", CASES)
@pytest.mark.parametrize("d", [32, 200, 30, 3])
@pytest.mark.parametrize("n", [1, 63, 5000])
def test_score_parity(model, kw, d, n):
    mdl, E0, R0, rs = _mk(model, 97, 11, d, **kw)
    t = _triples(rs, n, 97, 11)
    got = mdl.predict(t)
    ref = O.call(model, E0, R0, t, kw.get("norm", 1), kw.get("gamma", 12))
    assert got.shape == (n,) and got.dtype == np.float32
    np.testing.assert_allclose(got, ref, atol=TOL, rtol=1e-5)
    # func() on pre-gathered rows goes through the same kernel
    if n == 63:
        de = E0.shape[1]
        f = mdl.func(E0[t[:, 0]], R0[t[:, 1]], E0[t[:, 2]]).cpu().numpy()
        np.testing.assert_allclose(f, ref, atol=TOL, rtol=1e-5)
        assert de == mdl.e_dim


def test_score_kats():
    from kge_keras_amd.models import DistMult, TransE, ComplEx
    E = np.array([[1., 2.], [5., 6.], [0., 5.], [1., 1.]], dtype=np.float32)
    Rl = np.array([[3., 4.], [1., 1.]], dtype=np.float32)
    dm = DistMult(e_dim=2, r_dim=2, num_entities=4, num_relations=2, dp=0, init_entities=E, init_relations=Rl)
    assert dm.predict([[0, 0, 1]])[0] == 63.0  # models.py:172
    te = TransE(e_dim=2, r_dim=2, num_entities=4, num_relations=2, dp=0, init_entities=E, init_relations=Rl)
    assert te.predict([[0, 1, 2]])[0] == 8.0  # 12 - (|2| + |-2|)
    cx = ComplEx(e_dim=1, r_dim=1, num_entities=4, num_relations=2, dp=0, init_entities=E, init_relations=Rl)
    assert cx.predict([[0, 0, 1]])[0] == 35.0  # Re((1+2i)(3+4i)conj(5+6i))


def test_score_fb15k237_real_batches(datasets):
    """Config 2/3 of BASELINE.json on the real valid+test triples that exist (38 001 rows), d=200."""
    from kge_keras_amd.models import DistMult, TransE
    N, R = (int(x) for x in datasets["fb15k237_dims"])
    t = np.concatenate([datasets["fb15k237_valid"], datasets["fb15k237_test"]])
    rs = np.random.RandomState(1234)
    E0, R0 = O.glorot_uniform(N, 200, rs), O.glorot_uniform(R, 200, rs)
    for cls, name, kw in ((DistMult, "DistMult", {}), (TransE, "TransE", dict(norm=2, gamma=12))):
        mdl = cls(e_dim=200, r_dim=200, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0, **kw)
        got = mdl.predict(t)
        ref = O.call(name, E0, R0, t, kw.get("norm", 1), kw.get("gamma", 12))
        np.testing.assert_allclose(got, ref, atol=TOL, rtol=1e-5)


def test_out_of_range_ids():
    mdl, E0, R0, rs = _mk("DistMult", 10, 3, 8)
    with pytest.raises(IndexError):  # host arrays are checked before upload
        mdl.predict([[0, 0, 10]])
    bad = torch.tensor([[0, 0, 1], [0, 3, 1], [-1, 0, 1]], dtype=torch.int32, device="cuda")
    sc = mdl.call(bad).cpu().numpy()
    assert np.isfinite(sc[0]) and np.isnan(sc[1]) and np.isnan(sc[2])
    with pytest.raises(IndexError):  # device arrays are checked by the kernel through the error flag
        mdl._raise_on_flag()
    assert mdl.predict(np.zeros((0, 3), dtype=np.int64)).shape == (0,)  # empty input


LOSSES = ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss", "pairwize_hinge", "pairwize_logistic",
          "pairwize_square_loss"]


@pytest.mark.parametrize("model,kw", CASES)
@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_train_step_parity(model, kw, loss, opt):
    N, R, d, m = 40, 6, 32, 300
    mdl, E0, R0, rs = _mk(model, N, R, d, loss_function=loss, margin=0.5, **kw)
    mdl.compile(opt)
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        sp = O.call(model, Eo, Ro, pos, kw.get("norm", 1), kw.get("gamma", 12))
        sn = O.call(model, Eo, Ro, neg, kw.get("norm", 1), kw.get("gamma", 12))
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 0.5, 1, kw.get("norm", 1), kw.get("gamma", 12))
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0  # accumulators re-zeroed


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=1, gamma=0.9)), ("ComplEx", {}), ("HolE", {})])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_cross_entropy_train_step(model, kw, opt):
    """pointwize_cross_entropy (utils.py:297-298) = Keras BCE on the raw scores: 1/score gradients, so the parity check runs
    where the loss is well conditioned (scores mostly inside (0.05, 0.95)) plus rows saturated on both sides, whose
    gradient is exactly zero."""
    N, R, d, m = 40, 6, 8, 300
    rs = np.random.RandomState(11)
    de, dr = O.table_dims(model, d, d)
    if model == "TransE":
        E0, R0 = rs.uniform(-0.03, 0.03, (N, de)), rs.uniform(0.0, 0.06, (R, dr))
    else:
        E0, R0 = rs.uniform(0.35, 0.6, (N, de)), rs.uniform(0.35, 0.6, (R, dr))
    E0[:3] *= 4.0   # a few rows that push scores above 1 / below 0 (TransE)
    E0, R0 = E0.astype(np.float32), R0.astype(np.float32)
    from kge_keras_amd import models as M
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                            loss_function="pointwize_cross_entropy", **kw)
    mdl.compile(opt)
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_cross_entropy", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)


def test_train_gradients_exact_semantics():
    """SGD with lr=1 turns the weight delta into the summed sparse gradient: duplicates must accumulate."""
    from kge_keras_amd.models import DistMult, _Optimizer
    N, R, d = 5, 2, 8
    rs = np.random.RandomState(3)
    E0 = rs.uniform(-1, 1, (N, d)).astype(np.float32)
    R0 = rs.uniform(-1, 1, (R, d)).astype(np.float32)
    mdl = DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0,
                   loss_function="pointwize_square_loss")
    mdl.compile(_Optimizer("sgd", lr=1.0))
    pos = np.array([[0, 0, 1]] * 7 + [[1, 1, 1]])  # same triple seven times + a self-loop (s == o)
    neg = np.array([[2, 0, 3]] * 4)
    mdl.train_step([pos, neg])
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    L, _, _, gE, gR = O.dense_grads("DistMult", Eo, Ro, pos, neg, "pointwize_square_loss", 1.0)
    np.testing.assert_allclose(E0 - mdl.E.cpu().numpy(), gE, atol=2e-5)
    np.testing.assert_allclose(R0 - mdl.Rel.cpu().numpy(), gR, atol=2e-5)
    assert np.all(gE[4] == 0) and np.array_equal(mdl.E.cpu().numpy()[4], E0[4])  # untouched row: bit-identical


def test_l3_regulariser():
    mdl, E0, R0, rs = _mk("DistMult", 30, 4, 16, regularization=0.01)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for _ in range(2):
        pos, neg = _triples(rs, 100, 30, 4), _triples(rs, 100, 30, 4)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, "pointwize_hinge", 1.0, regularization=0.01)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        assert_tables(mdl.E.cpu().numpy(), Eo)


def test_nations_harness_fit_matches_oracle(datasets):
    """Config 1 of BASELINE.json: nations, DistMult d=32, 10 negatives, B=32, the harness's pairwise hinge with the
    (m,m) broadcast AND pointwise hinge, Adam, dp=0 -- batches from the bit-exact generator, 12 steps."""
    from kge_keras_amd.harness import DataGenerator, build_model
    tr = datasets["nations_train"]
    N, R = 14, 55
    for loss in ("pairwize_hinge", "pointwize_hinge"):
        rs = np.random.RandomState(1234)
        E0, R0 = O.glorot_uniform(N, 32, rs), O.glorot_uniform(R, 32, rs)
        mdl = build_model(dict(num_entities=N, num_relations=R, dim=32, embedding_model="DistMult", loss_function=loss,
                               margin=1, dp=0.0, init_entities=E0, init_relations=R0))
        np.random.seed(0)
        random.seed(0)
        gen = DataGenerator(tr, ns=10, batch_size=32)
        Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.AdamTF()
        np.random.seed(0)
        random.seed(0)
        idx = O.epoch_permutation(len(tr))
        trl = [tuple(int(v) for v in t) for t in tr]
        for b in range(12):
            (pos, neg), _ = gen[b]
            np_state, py_state = np.random.get_state(), random.getstate()
            # the oracle consumes the same global streams: rewind to where the generator started this batch
            L = float(mdl.train_step([pos, neg])["loss"].item())
            Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, loss, 1.0, 1)
            assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (loss, b)
            np.random.set_state(np_state)
            random.setstate(py_state)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        del idx, trl


def test_fit_runs_and_learns(datasets):
    from kge_keras_amd.harness import DataGenerator, build_model
    np.random.seed(1)
    random.seed(1)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult",
                           loss_function="pointwize_logistic", margin=1, dp=0.0, seed=3))
    gen = DataGenerator(datasets["nations_train"], ns=10, batch_size=32)
    h = mdl.fit(gen, epochs=4, verbose=0, validation_data=DataGenerator(datasets["nations_valid"], ns=10, batch_size=32))
    assert len(h.history["loss"]) == 4 and h.history["loss"][-1] < h.history["loss"][0]
    assert len(h.history["val_loss"]) == 4


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=2, gamma=12))])
def test_full_size_fb15k237_step(model, kw):
    """BASELINE configs 2 and 3 at full size: N=14541, R=237, d=200, B=2048, ns=100 -> 409 600 scored triples."""
    from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
    from kge_keras_amd import models as M
    N, R, d, B, ns = 14541, 237, 200, 2048, 100
    rs = np.random.RandomState(1234)
    E0, R0 = O.glorot_uniform(N, d, rs), O.glorot_uniform(R, d, rs)
    if model == "TransE":  # keep scores in the active hinge region
        E0, R0 = E0 * 20, R0 * 2
    batch = _triples(rs, B, N, R)
    pos, neg = build_batch(NumpyStream(0), PythonStream(0), batch, ns, N)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0,
                            loss_function="pointwize_logistic", **kw)
    mdl.compile("adam")
    L = float(mdl.train_step([pos, neg])["loss"].item())
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    Lo = O.train_step(model, Eo, Ro, O.AdamTF(), pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
    assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
    assert_tables(mdl.E.cpu().numpy(), Eo)
    assert_tables(mdl.Rel.cpu().numpy(), Ro)
    # size-independent property: the update is a function of the multiset of rows, not of their order
    mdl2 = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0,
                             loss_function="pointwize_logistic", **kw)
    mdl2.compile("adam")
    perm = np.random.RandomState(5).permutation(len(pos))
    mdl2.train_step([pos[perm], neg[perm[::-1]]])
    np.testing.assert_allclose(mdl2.E.cpu().numpy(), mdl.E.cpu().numpy(), atol=2e-5)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=1, gamma=12)), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {})])
@pytest.mark.parametrize("loss", ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss"])
def test_duplicate_row_folding(model, kw, loss):
    """kge_set_dedup_mode: the step on the distinct rows weighted by multiplicity == the step on every row == the oracle.
    Batches like oversample_data's (few distinct positives, many repeats), a triple present in BOTH arrays, scores_out
    expanded back to the input rows, and ids outside the tables (error flag + NaN score, the row takes no part)."""
    from kge_keras_amd import _lib
    N, R, d = 50, 7, 32
    res = {}
    for mode in (2, 0):
        _lib.check(_lib.lib.kge_set_dedup_mode(mode))
        try:
            mdl, E0, R0, rs = _mk(model, N, R, d, loss_function=loss, margin=0.5, **kw)
            mdl.compile("adam")
            Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
            oo = O.AdamTF()
            out = []
            for step in range(3):
                base = _triples(rs, 12, N, R)
                pos = base[rs.randint(0, 12, 700)]
                negs = _triples(rs, 400, N, R)
                neg = np.concatenate([negs[rs.randint(0, 400, 500)], base[:3], base[:3]])  # positives reappear as negatives
                sc = torch.empty(len(pos) + len(neg), dtype=torch.float32, device="cuda")
                L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
                sp = O.call(model, Eo, Ro, pos, kw.get("norm", 1), kw.get("gamma", 12))
                sn = O.call(model, Eo, Ro, neg, kw.get("norm", 1), kw.get("gamma", 12))
                np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
                Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 0.5, 1, kw.get("norm", 1), kw.get("gamma", 12))
                assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (mode, step)
                assert_tables(mdl.E.cpu().numpy(), Eo)
                assert_tables(mdl.Rel.cpu().numpy(), Ro)
                out.append(L)
            res[mode] = (out, mdl.E.cpu().numpy())
            # ids outside the tables
            bad = torch.tensor([[0, 0, 1], [0, R, 1], [-1, 0, 1], [0, 0, 1]], dtype=torch.int32, device="cuda")
            sc = torch.empty(8, dtype=torch.float32, device="cuda")
            mdl.train_step([bad, bad], scores_out=sc)
            v = sc.cpu().numpy()
            assert np.isfinite(v[[0, 3, 4, 7]]).all() and np.isnan(v[[1, 2, 5, 6]]).all() and v[0] == v[3]
            with pytest.raises(IndexError):
                mdl._raise_on_flag()
        finally:
            _lib.lib.kge_set_dedup_mode(1)
    np.testing.assert_allclose(res[2][0], res[0][0], rtol=1e-5)
    np.testing.assert_allclose(res[2][1], res[0][1], atol=2e-5)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {})])
@pytest.mark.parametrize("dp", [0.0, 0.3])
def test_batch_plan_step_equals_plain_step(model, kw, dp, grad_mode):
    """kge_batch_plan_build on another stream + kge_train_step_planned == kge_train_step on the same batch == the oracle;
    a plan is reusable as a buffer; models / losses without a plan say so."""
    from kge_keras_amd import _lib, models as M
    if grad_mode == "push":
        mdl, *_ = _mk(model, 20, 3, 8, **kw)
        mdl.compile("adam")
        assert not mdl.plan_supported()
        with pytest.raises(NotImplementedError):
            mdl.plan_batch([np.zeros((4, 3), dtype=np.int64)] * 2)
        return
    N, R, d = 60, 5, 32
    for fold_mode in (2, 0):
        _lib.check(_lib.lib.kge_set_dedup_mode(fold_mode))
        try:
            rs = np.random.RandomState(4)
            de, dr = O.table_dims(model, d, d)
            E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
            R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
            a, b = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, init_entities=E0, init_relations=R0,
                                      loss_function="pointwize_logistic", **kw) for _ in range(2))
            a.compile("adam")
            b.compile("adam")
            assert a.plan_supported()
            Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
            oo = O.AdamTF()
            side = torch.cuda.Stream()
            plan = None
            for step in range(3):
                base = _triples(rs, 30, N, R)
                pos, neg = base[rs.randint(0, 30, 500)], _triples(rs, 400, N, R)[rs.randint(0, 400, 600)]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    plan = a.plan_batch([pos, neg], out=plan)
                torch.cuda.current_stream().wait_stream(side)
                La = float(a.train_step(plan)["loss"].item())
                Lb = float(b.train_step([pos, neg])["loss"].item())
                Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12), 0.0, dp, b._seed)
                assert La == pytest.approx(Lb, rel=1e-5, abs=1e-6) and La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (fold_mode, step)
                np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=2e-5)
                assert_tables(a.E.cpu().numpy(), Eo)
                assert_tables(a.Rel.cpu().numpy(), Ro)
            # the gradient half alone (data-parallel split) + ids outside the tables raise through the plan too
            g = a.train_grads(a.plan_batch([pos, neg]))
            assert np.isfinite(float(g["loss"].item())) and float(a._gE.abs().max()) > 0
            a.apply_grads()
            bad = torch.tensor([[0, 0, 1], [0, R, 1]], dtype=torch.int32, device="cuda")
            a.train_step(a.plan_batch([bad, bad]))
            with pytest.raises(IndexError):
                a._raise_on_flag()
        finally:
            _lib.lib.kge_set_dedup_mode(1)
    h = M.HolE(e_dim=8, r_dim=8, num_entities=N, num_relations=R, dp=0.0, spectral=False)  # direct-correlation kernels: no plan
    h.compile("adam")
    p2 = M.DistMult(e_dim=8, r_dim=8, num_entities=N, num_relations=R, dp=0.0, loss_function="pairwize_hinge")
    p2.compile("adam")
    assert not h.plan_supported() and not p2.plan_supported()


@pytest.mark.parametrize("model,kw", CASES)
@pytest.mark.parametrize("loss", ["pointwize_logistic", "pairwize_hinge"])
def test_dropout_training_parity(model, kw, loss, grad_mode):
    """Dropout(dp) on the three gathered rows (models.py:122,144-148): masks are regenerated by the oracle from the
    same (seed, row, role, element) hash, so the whole step is checked, forward and backward."""
    if loss.startswith("pairwize") and grad_mode == "push":
        pytest.skip("pairwise + dropout is built on the pull path only")
    N, R, d, m = 40, 6, 32, 300
    from kge_keras_amd import models as M
    rs = np.random.RandomState(11)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.3, init_entities=E0, init_relations=R0,
                            loss_function=loss, margin=0.5, seed=5, **kw)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 0.5, 1, kw.get("norm", 1), kw.get("gamma", 12),
                          dropout=0.3, seed=mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        np.testing.assert_allclose(mdl.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
        np.testing.assert_allclose(mdl.Rel.cpu().numpy(), Ro, atol=TOL_W_LONG)
    # predict is dropout-free (keras Dropout is the identity outside fit)
    t = _triples(rs, 50, N, R)
    np.testing.assert_allclose(mdl.predict(t), O.call(model, Eo, Ro, t, kw.get("norm", 1), kw.get("gamma", 12)), atol=2 * TOL)


@pytest.mark.parametrize("d", [3, 30, 200, 256, 260])
@pytest.mark.parametrize("dp", [0.0, 0.3])
def test_hole_direct_kernels_over_row_widths(d, dp):
    """The O(d^2) HolE kernels (csrc/hole.hip: what dropout steps run on): rows up to 256 floats take two problems per wave, eight
    outputs per lane (ccorr8) -- widths that are not multiples of four or eight, the boundary itself -- wider rows one problem per wave
    in passes of 256 outputs; destination lists of odd and even length, a row nobody touches."""
    from kge_keras_amd import models as M
    N, R, m = 23, 4, 77
    rs = np.random.RandomState(d)
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    mdl = M.HolE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, init_entities=E0, init_relations=R0, loss_function="pointwize_logistic",
                 seed=3, spectral=False)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        pos, neg = _triples(rs, m, N - 1, R), _triples(rs, m + 4, N - 1, R)  # (entity N-1 stays untouched)
        sc = torch.empty(2 * m + 4, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lo = O.train_step("HolE", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1, 1, 1, 12, dropout=dp, seed=mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        np.testing.assert_allclose(mdl.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
        np.testing.assert_allclose(mdl.Rel.cpu().numpy(), Ro, atol=TOL_W_LONG)
    t = _triples(rs, 9, N, R)
    np.testing.assert_allclose(mdl.predict(t), O.call("HolE", Eo, Ro, t, 1, 12), atol=2 * TOL)


def _pair_sort(mode):
    """kge_set_pair_sort_mode: "0" all-pairs kernels, "1" sorted scores / four sums, None: by pair count (the default)."""
    from kge_keras_amd import _lib
    _lib.check(_lib.lib.kge_set_pair_sort_mode(-1 if mode is None else int(mode)))


@pytest.fixture(autouse=True)
def _pair_sort_default():
    yield
    _pair_sort(None)


@pytest.mark.parametrize("m_t,m_f", [(1, 7), (300, 300), (5000, 4097), (1, 3000), (2048, 20480)])
@pytest.mark.parametrize("margin", [1.0, 0.25])
def test_pairwise_hinge_from_sorted_scores_equals_all_pairs(m_t, m_f, margin, monkeypatch):
    """csrc/pairsort.hip: the all-pairs mean of pairwize_hinge (utils.py:300-305 with the harness's negative_samples = 1) from the two
    sorted score vectors -- counts and a running sum instead of m_f * m_t pairs -- against the all-pairs kernels and the oracle.
    Scores on a quarter grid, so that many pairs sit exactly on the hinge's corner (relu'(0) = 0: strict comparisons on both sides),
    plus a NaN on either side (rows with bad ids: no part in the loss, zero gradient)."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(m_t + m_f)
    t = (np.round(rs.uniform(-3, 3, m_t) * 4) / 4).astype(np.float32)
    f = (np.round(rs.uniform(-3, 3, m_f) * 4) / 4).astype(np.float32)
    out = {}
    for mode in ("0", "1"):
        _pair_sort(mode)
        L, gt, gf = M.loss_value("pairwize_hinge", t, f, margin, 1, return_grads=True)
        out[mode] = (float(L.item()), gt.cpu().numpy(), gf.cpu().numpy())
    Lo, gto, gfo = O.loss_value_and_grads("pairwize_hinge", t.astype(np.float64), f.astype(np.float64), margin, 1)
    for mode in ("0", "1"):
        L, gt, gf = out[mode]
        assert L == pytest.approx(float(Lo), rel=1e-5, abs=1e-6), mode
        np.testing.assert_allclose(gt, gto, rtol=1e-5, atol=1e-9, err_msg=mode)
        np.testing.assert_allclose(gf, gfo, rtol=1e-5, atol=1e-9, err_msg=mode)
    if m_t > 1 and m_f > 1:  # NaN rows
        t2, f2 = t.copy(), f.copy()
        t2[0] = np.nan
        f2[-1] = np.nan
        _pair_sort("1")
        L, gt, gf = M.loss_value("pairwize_hinge", t2, f2, margin, 1, return_grads=True)
        gt, gf = gt.cpu().numpy(), gf.cpu().numpy()
        keep_t, keep_f = t2[1:].astype(np.float64), f2[:-1].astype(np.float64)
        Lk, gtk, gfk = O.loss_value_and_grads("pairwize_hinge", keep_t, keep_f, margin, 1)
        scale = (len(keep_t) * len(keep_f)) / (m_t * m_f)  # the mean still divides by all m_f * m_t entries
        assert gt[0] == 0.0 and gf[-1] == 0.0
        assert float(L.item()) == pytest.approx(float(Lk) * scale, rel=1e-5, abs=1e-6)
        np.testing.assert_allclose(gt[1:], gtk * scale, rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(gf[:-1], gfk * scale, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("m_t,m_f", [(1, 7), (300, 300), (5000, 4097), (1, 3000), (2048, 20480)])
def test_pairwise_square_from_sums_equals_all_pairs(m_t, m_f, monkeypatch):
    """csrc/pairsort.hip: -(f_j - t_i)^2 over all pairs (utils.py:314-319 with negative_samples = 1) separates into four sums; against
    the all-pairs kernels and the oracle."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(m_t * 7 + m_f)
    t, f = rs.uniform(-3, 3, m_t).astype(np.float32), rs.uniform(-2, 4, m_f).astype(np.float32)
    Lo, gto, gfo = O.loss_value_and_grads("pairwize_square_loss", t.astype(np.float64), f.astype(np.float64), 1.0, 1)
    for mode in ("0", "1"):
        _pair_sort(mode)
        L, gt, gf = M.loss_value("pairwize_square_loss", t, f, 1.0, 1, return_grads=True)
        assert float(L.item()) == pytest.approx(float(Lo), rel=2e-5), mode
        np.testing.assert_allclose(gt.cpu().numpy(), gto, rtol=2e-4, atol=1e-7, err_msg=mode)
        np.testing.assert_allclose(gf.cpu().numpy(), gfo, rtol=2e-4, atol=1e-7, err_msg=mode)
    if m_t > 1 and m_f > 1:
        # a NaN score (a row with ids outside the tables) on the `true` side: -(f - t)^2 has no corner to hide it behind -- in the reference
        # the mean is NaN and so is every d/df_j; d/dt_i of the other rows stays finite.  Both forms must show the same pattern.
        t2 = t.copy()
        t2[0] = np.nan
        for mode in ("0", "1"):
            _pair_sort(mode)
            L, gt, gf = M.loss_value("pairwize_square_loss", t2, f, 1.0, 1, return_grads=True)
            gt, gf = gt.cpu().numpy(), gf.cpu().numpy()
            assert np.isnan(float(L.item())) and np.isnan(gt[0]) and np.all(np.isnan(gf)), mode
            np.testing.assert_allclose(gt[1:], gto[1:], rtol=2e-4, atol=1e-7, err_msg=mode)


def test_dropout_default_rate_trains(datasets):
    """The reference's default dp=0.2 (models.py:22) through fit()."""
    from kge_keras_amd.harness import DataGenerator, build_model
    np.random.seed(2)
    random.seed(2)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult",
                           loss_function="pointwize_logistic", margin=1, seed=4))
    assert mdl.dp == 0.2
    h = mdl.fit(DataGenerator(datasets["nations_train"], ns=10, batch_size=32), epochs=3, verbose=0)
    assert h.history["loss"][-1] < h.history["loss"][0]


def test_changing_batch_size_and_loss_on_one_model():
    """The workspace counters sit at a fixed header and are left clean by every step: batch sizes and the gradient-path
    variant (pointwise -> relation-grouped forward, pairwise -> generic lists) may change between steps of one model."""
    N, R, d = 60, 9, 32
    mdl, E0, R0, rs = _mk("DistMult", N, R, d, loss_function="pointwize_hinge", margin=0.5)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for m, loss in ((300, "pointwize_hinge"), (64, "pairwize_hinge"), (1000, "pointwize_logistic"), (7, "pointwize_hinge"),
                    (513, "pairwize_logistic"), (300, "pointwize_square_loss")):
        mdl.loss_function = loss
        pos, neg = _triples(rs, m, N, R), _triples(rs, m + 3, N, R)  # len(pos) != len(neg) is legal for the reference's losses
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, loss, 0.5, 1)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (m, loss)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)


@pytest.mark.parametrize("name", LOSSES + ["pointwize_cross_entropy"])
def test_loss_functions_callable(name):
    """loss_function_lookup(name)(true, false, margin, negative_samples) as the reference exposes it (utils.py:288-330)."""
    from kge_keras_amd.utils import loss_function_lookup
    from kge_keras_amd.models import loss_value
    rs = np.random.RandomState(2)
    t, f = rs.uniform(-2, 2, 700), rs.uniform(-2, 2, 300)
    L = float(loss_function_lookup(name)(t, f, 0.7, 1).item())
    Lo, gto, gfo = O.loss_value_and_grads(name, t, f, 0.7, 1)
    assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
    _, gt, gf = loss_value(name, t, f, 0.7, 1, return_grads=True)
    np.testing.assert_allclose(gt.cpu().numpy(), gto, atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(gf.cpu().numpy(), gfo, atol=1e-6, rtol=1e-4)
    if name.startswith("pairwize"):
        with pytest.raises(ValueError):  # (100,3) does not broadcast against (700,)
            loss_function_lookup(name)(t, f, 0.7, 3)
        # reduce_mean=False: the raw pair tensor, for the three legal broadcasts (all pairs, one positive, column aligned)
        for tt, ff, ns in ((t[:50], f[:40], 1), (t[:1], f, 5), (t[:60], f, 60)):
            P = loss_function_lookup(name)(tt, ff, 0.7, ns, reduce_mean=False).cpu().numpy()
            Po = O.loss_pairs(name, tt, ff, 0.7, ns)
            assert P.shape == Po.shape
            np.testing.assert_allclose(P, Po, atol=1e-5, rtol=1e-5)
        # column-aligned broadcast, reduced: negative_samples == len(true)
        L = float(loss_function_lookup(name)(t[:60], f, 0.7, 60).item())
        Lo, gto, gfo = O.loss_value_and_grads(name, t[:60], f, 0.7, 60)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        _, gt, gf = loss_value(name, t[:60], f, 0.7, 60, return_grads=True)
        np.testing.assert_allclose(gt.cpu().numpy(), gto, atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(gf.cpu().numpy(), gfo, atol=1e-6, rtol=1e-4)
    else:  # the pointwize losses ignore reduce_mean (utils.py:288-298)
        assert float(loss_function_lookup(name)(t, f, 0.7, 1, reduce_mean=False).item()) == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
    with pytest.raises(KeyError):
        loss_function_lookup("nope")


@pytest.mark.parametrize("model", ["DistMult", "HolE"])
def test_pairwise_column_aligned_train_step(model):
    """negative_samples == len(true): false.reshape(-1, ns) broadcasts column-wise against true (utils.py:301,308,315)."""
    N, R, d, mt, rows = 30, 4, 16, 24, 5
    for loss in ("pairwize_hinge", "pairwize_logistic", "pairwize_square_loss"):
        mdl, E0, R0, rs = _mk(model, N, R, d, loss_function=loss, margin=0.5)
        mdl.loss_negative_samples = mt
        mdl.compile("adam")
        Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
        oo = O.AdamTF()
        for _ in range(2):
            pos, neg = _triples(rs, mt, N, R), _triples(rs, mt * rows, N, R)
            L = float(mdl.train_step([pos, neg])["loss"].item())
            Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 0.5, mt)
            assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), loss
            assert_tables(mdl.E.cpu().numpy(), Eo)
            assert_tables(mdl.Rel.cpu().numpy(), Ro)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {})])
@pytest.mark.parametrize("dp", [0.0, 0.2])
def test_heavy_tailed_batch(model, kw, dp, grad_mode):
    """Real graphs are heavy tailed.  A batch where one relation carries 80 % of the rows (> 4096 of them: the forward
    switches to 128-row units and the relation's partial gradient rows are summed by a whole workgroup) and entity ids are
    Zipf distributed (hub rows split into chunks; the entity histogram aggregates in LDS), with and without dropout."""
    N, R, d, m = 300, 3, 32, 6000
    rs = np.random.RandomState(21)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    from kge_keras_amd import models as M
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, init_entities=E0, init_relations=R0,
                            loss_function="pointwize_logistic", seed=2, **kw)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    pe = 1.0 / np.arange(1, N + 1) ** 1.0
    pe /= pe.sum()

    def draw(n):
        return np.stack([rs.choice(N, n, p=pe), rs.choice(R, n, p=[0.8, 0.15, 0.05]), rs.choice(N, n, p=pe)], 1)
    for step in range(2):
        pos, neg = draw(m), draw(m)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12), 0.0, dp, mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    if grad_mode == "pull":  # the same through a plan built ahead
        pos, neg = draw(m), draw(m)
        L = float(mdl.train_step(mdl.plan_batch([pos, neg]))["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12), 0.0, dp, mdl._seed)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)


@pytest.mark.parametrize("loss", ["pairwize_hinge", "pairwize_logistic", "pairwize_square_loss"])
def test_pairwise_allpairs_at_script_scale(loss):
    """The all-pairs broadcast at sizes where the sweep over the other score vector is cut into column chunks (the reference's
    own script trains with pairwize_hinge on 20 480-row arrays, tests.py:119-134): loss value / gradients of the callable and
    one Adam step, against the oracle."""
    from kge_keras_amd.utils import loss_function_lookup
    from kge_keras_amd.models import loss_value
    rs = np.random.RandomState(5)
    t, f = rs.uniform(-2, 2, 5000), rs.uniform(-2, 2, 3100)
    L = float(loss_function_lookup(loss)(t, f, 0.7, 1).item())
    Lo, gto, gfo = O.loss_value_and_grads(loss, t, f, 0.7, 1)
    assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
    _, gt, gf = loss_value(loss, t, f, 0.7, 1, return_grads=True)
    np.testing.assert_allclose(gt.cpu().numpy(), gto, atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(gf.cpu().numpy(), gfo, atol=1e-7, rtol=1e-4)
    N, R, d, m = 60, 5, 16, 2600
    mdl, E0, R0, rs = _mk("DistMult", N, R, d, loss_function=loss, margin=0.5)
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    pos, neg = _triples(rs, m, N, R), _triples(rs, m + 300, N, R)
    Ls = float(mdl.train_step([pos, neg])["loss"].item())
    Lso = O.train_step("DistMult", Eo, Ro, O.AdamTF(), pos, neg, loss, 0.5, 1)
    assert Ls == pytest.approx(Lso, rel=TOL_L, abs=TOL_LA)
    assert_tables(mdl.E.cpu().numpy(), Eo)
    assert_tables(mdl.Rel.cpu().numpy(), Ro)


def test_wn18rr_complex400_train_step(datasets):
    """BASELINE config 4 shape in training: WN18RR (N = 40 943, R = 11), ComplEx d=400 (rows of 800 floats -> the generic-list
    pull variant with two float4 chunks per lane), real training triples, one Adam step against the oracle."""
    from kge_keras_amd.models import ComplEx
    from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
    N, R, d = 40943, 11, 400
    rs = np.random.RandomState(7)
    E0, R0 = O.glorot_uniform(N, 2 * d, rs) * 10, O.glorot_uniform(R, 2 * d, rs) * 3
    batch = datasets["wn18rr_train"][:256]
    pos, neg = build_batch(NumpyStream(1), PythonStream(1), batch, 20, N)
    mdl = ComplEx(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0, loss_function="pointwize_logistic")
    mdl.compile("adam")
    L = float(mdl.train_step([pos, neg])["loss"].item())
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    Lo = O.train_step("ComplEx", Eo, Ro, O.AdamTF(), pos, neg, "pointwize_logistic", 1.0)
    assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
    assert_tables(mdl.E.cpu().numpy(), Eo)
    assert_tables(mdl.Rel.cpu().numpy(), Ro)


def test_edge_cases_single_row_and_skewed_batches():
    """m = 1, one relation / one entity receiving every contribution (hub lists split into many chunks), odd widths."""
    for model, d in (("DistMult", 32), ("TransE", 30), ("ComplEx", 6)):
        kw = dict(norm=1, gamma=12) if model == "TransE" else {}
        N, R = 9, 3
        mdl, E0, R0, rs = _mk(model, N, R, d, loss_function="pointwize_logistic", **kw)
        mdl.compile("sgd")
        Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.SGDTF()
        for pos, neg in ((np.array([[0, 0, 1]]), np.array([[2, 1, 3]])),
                         (np.tile(np.array([[4, 2, 4]]), (1500, 1)), np.tile(np.array([[4, 2, 5]]), (700, 1)))):
            L = float(mdl.train_step([pos, neg])["loss"].item())
            Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
            assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (model, len(pos))
            assert_tables(mdl.E.cpu().numpy(), Eo)
            assert_tables(mdl.Rel.cpu().numpy(), Ro)


@pytest.mark.parametrize("d", [31, 200, 6, 2])
@pytest.mark.parametrize("loss", ["pointwize_hinge", "pairwize_logistic"])
def test_hole_spectral_step(d, loss):
    """HolE without dropout trains in the frequency domain (csrc/hole_spec.hip): even and odd widths (the Hermitian weights of
    the half spectrum differ), against the fp64 oracle of models.py:231-237, and against the direct-correlation kernels."""
    N, R, m = 50, 5, 400
    mdl, E0, R0, rs = _mk("HolE", N, R, d, loss_function=loss, margin=0.5)
    ref, _, _, _ = _mk("HolE", N, R, d, loss_function=loss, margin=0.5, spectral=False)
    assert mdl._spectral_on() and not ref._spectral_on()
    mdl.compile("adam"); ref.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer("adam")
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        Lr = float(ref.train_step([pos, neg])["loss"].item())
        sp, sn = O.call("HolE", Eo, Ro, pos, 1, 12), O.call("HolE", Eo, Ro, neg, 1, 12)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
        Lo = O.train_step("HolE", Eo, Ro, oo, pos, neg, loss, 0.5, 1, 1, 12)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA) and Lr == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        np.testing.assert_allclose(mdl.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
        np.testing.assert_allclose(mdl.Rel.cpu().numpy(), Ro, atol=TOL_W_LONG)
        np.testing.assert_allclose(ref.E.cpu().numpy(), Eo, atol=TOL_W_LONG)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0
    inner = mdl._spec["inner"]
    assert float(inner._gE.abs().max()) == 0.0 and float(inner._gR.abs().max()) == 0.0  # spectral accumulators re-zeroed


def test_hole_spectral_planned_step_and_halves():
    """A BatchPlan (ids-only half of the step) drives the frequency-domain HolE step; train_grads + apply_grads equals train_step."""
    N, R, d, m = 300, 7, 64, 6000
    a, E0, R0, rs = _mk("HolE", N, R, d)
    b, _, _, _ = _mk("HolE", N, R, d)
    c, _, _, _ = _mk("HolE", N, R, d)
    for x in (a, b, c):
        x.compile("adam")
    if not a.plan_supported():
        pytest.skip("batch plans exist on the pull gradient path only")
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        La = float(a.train_step([pos, neg])["loss"].item())
        Lb = float(b.train_step(b.plan_batch([pos, neg]))["loss"].item())
        c.train_grads([pos, neg]); c.apply_grads()
        assert Lb == pytest.approx(La, abs=1e-5, rel=1e-5) and float(c._loss[0].item()) == pytest.approx(La, abs=1e-5, rel=1e-5)
        np.testing.assert_allclose(b.E.cpu().numpy(), a.E.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(c.E.cpu().numpy(), a.E.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(b.Rel.cpu().numpy(), a.Rel.cpu().numpy(), atol=2e-6)
    hd, _, _, _ = _mk("HolE", N, R, d)
    hd.dp = 0.3
    assert not hd._spectral_on() and not hd.plan_supported()


# ---------------------------------------------------------------------------------------------- pRotatE (models.py:461-489)
def _mk_protate(N, R, d, gamma=12, amp=0.5, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(5)
    E0 = rs.uniform(-amp, amp, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-amp, amp, (R, 2 * d)).astype(np.float32)
    mdl = M.pRotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, gamma=gamma, init_entities=E0, init_relations=R0, **kw)
    assert mdl.e_dim == 2 * d and mdl.modulus == pytest.approx(O.protate_modulus(2 * d, gamma, 2, kw.get("modulus", 0.5)))
    return mdl, E0, R0, rs


@pytest.mark.parametrize("d,n", [(16, 1), (100, 63), (15, 5000), (3, 77)])
@pytest.mark.parametrize("gamma,amp", [(12, 0.5), (0, 4.0)])
def test_protate_score_parity(d, n, gamma, amp):
    mdl, E0, R0, rs = _mk_protate(97, 11, d, gamma, amp)
    t = _triples(rs, n, 97, 11)
    ref = O.call("pRotatE", E0, R0, t, mdl.modulus, gamma)
    np.testing.assert_allclose(mdl.predict(t), ref, atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("gamma", [12, 0])
def test_protate_train_step_parity(loss, opt, gamma):
    N, R, d, m = 40, 6, 16, 300
    mdl, E0, R0, rs = _mk_protate(N, R, d, gamma, 2.0, loss_function=loss, margin=0.5, modulus=3.0)
    mdl.compile(opt)
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        sc = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        L = float(mdl.train_step([pos, neg], scores_out=sc)["loss"].item())
        sp, sn = O.call("pRotatE", Eo, Ro, pos, mdl.modulus, gamma), O.call("pRotatE", Eo, Ro, neg, mdl.modulus, gamma)
        np.testing.assert_allclose(sc.cpu().numpy(), np.concatenate([sp, sn]), atol=TOL, rtol=1e-5)
        Lo = O.train_step("pRotatE", Eo, Ro, oo, pos, neg, loss, 0.5, 1, mdl.modulus, gamma)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
    assert float(mdl._gE.abs().max()) == 0.0 and float(mdl._gR.abs().max()) == 0.0


def test_protate_large_planned_and_dropout():
    """Rows of 400 floats (d=200), a batch large enough for the relation-grouped path with duplicate folding and a BatchPlan; then a
    dropout step against the oracle's masks."""
    N, R, d, m = 500, 9, 200, 8000
    a, E0, R0, rs = _mk_protate(N, R, d, 12, 1.0)
    b, _, _, _ = _mk_protate(N, R, d, 12, 1.0)
    a.compile("adam"); b.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer("adam")
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        pos[: m // 2] = pos[m // 2: 2 * (m // 2)]  # duplicates
        La = float(a.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("pRotatE", Eo, Ro, oo, pos, neg, "pointwize_hinge", 1, 1, a.modulus, 12)
        assert La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        assert_tables(a.E.cpu().numpy(), Eo)
        assert_tables(a.Rel.cpu().numpy(), Ro)
        if b.plan_supported():
            Lb = float(b.train_step(b.plan_batch([pos, neg]))["loss"].item())
            assert Lb == pytest.approx(La, abs=1e-5, rel=1e-5)
            np.testing.assert_allclose(b.E.cpu().numpy(), a.E.cpu().numpy(), atol=2e-6)


def test_hole_spectral_large_table_row_blocks():
    """Tables of >= 4096 rows carry the gradients back in 16 row blocks (strided-batched GEMM + remainder): same step as the direct kernels."""
    N, R, d, m = 5003, 7, 32, 3000
    a, E0, R0, rs = _mk("HolE", N, R, d)
    b, _, _, _ = _mk("HolE", N, R, d, spectral=False)
    a.compile("adam"); b.compile("adam")
    for step in range(2):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        La = float(a.train_step([pos, neg])["loss"].item())
        Lb = float(b.train_step([pos, neg])["loss"].item())
        assert La == pytest.approx(Lb, abs=1e-5, rel=1e-5)
        np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=1e-5)
        np.testing.assert_allclose(a.Rel.cpu().numpy(), b.Rel.cpu().numpy(), atol=1e-5)


def test_hole_spectral_with_l3_regulariser():
    """The L3 regulariser (models.py:10-11,78-92) acts on the REAL entity table in the optimizer sweep while the data term comes back
    from the frequency domain: loss value and tables against the oracle."""
    N, R, d, m = 60, 5, 24, 500
    mdl, E0, R0, rs = _mk("HolE", N, R, d, loss_function="pointwize_logistic", regularization=0.01)
    assert mdl._spectral_on()
    mdl.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer("adam")
    for step in range(3):
        pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
        L = float(mdl.train_step([pos, neg])["loss"].item())
        Lo = O.train_step("HolE", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1, 1, 1, 12, regularization=0.01)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (step, L, Lo)
        assert_tables(mdl.E.cpu().numpy(), Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)


@pytest.mark.parametrize("norm,gamma", [(3, 12), (1.5, 12), (4, 0)])
def test_transe_general_p_norm(norm, gamma):
    """tf.norm(x, ord=p) for any positive real p (models.py:188 `ord=self.norm`): scores, train steps on every loss family (push path),
    filtered ranks."""
    N, R, d, m = 60, 6, 24, 300
    for loss, opt in (("pointwize_hinge", "adam"), ("pointwize_logistic", "sgd"), ("pairwize_hinge", "adam")):
        mdl, E0, R0, rs = _mk("TransE", N, R, d, norm=norm, gamma=gamma, loss_function=loss, margin=0.5)
        t = _triples(rs, 500, N, R)
        np.testing.assert_allclose(mdl.predict(t), O.call("TransE", E0, R0, t, norm, gamma), atol=TOL, rtol=1e-5)
        assert not mdl.plan_supported()
        mdl.compile(opt)
        Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
        oo = O.make_optimizer(opt)
        for step in range(2):
            pos, neg = _triples(rs, m, N, R), _triples(rs, m, N, R)
            L = float(mdl.train_step([pos, neg])["loss"].item())
            Lo = O.train_step("TransE", Eo, Ro, oo, pos, neg, loss, 0.5, 1, norm, gamma)
            assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), (loss, step)
            assert_tables(mdl.E.cpu().numpy(), Eo)
            assert_tables(mdl.Rel.cpu().numpy(), Ro)
    test = list(dict.fromkeys(tuple(int(v) for v in x) for x in _triples(rs, 40, N, R)))
    filt = [tuple(int(v) for v in x) for x in _triples(rs, 200, N, R)]
    filt = [x for x in filt if x not in set(test)]
    fh, ft = O.build_filters(filt)
    tails, heads = mdl.rank(test, N, {k: sorted(v) for k, v in ft.items()}, {k: sorted(v) for k, v in fh.items()})
    margins = []
    to, ho = O.ranks_closed_form("TransE", mdl.E.cpu().numpy(), mdl.Rel.cpu().numpy(), test, filt, norm, gamma, margin_out=margins)
    mm = np.asarray(margins).reshape(-1, 2)
    st, sh = mm[:, 0] > 1e-4, mm[:, 1] > 1e-4
    assert np.array_equal(tails[st], to[st]) and np.array_equal(heads[sh], ho[sh]) and st.mean() > 0.8
    from kge_keras_amd import models as M
    with pytest.raises(NotImplementedError):
        M.TransE(e_dim=8, r_dim=8, num_entities=5, num_relations=2, norm=np.inf).predict(np.array([[0, 0, 1]]))


@pytest.mark.parametrize("m_t,m_f", [(1024, 1024), (2048, 3000), (5000, 4097), (2048, 20480), (20480, 20480)])
def test_pairwise_logistic_one_pass_equals_all_pairs(m_t, m_f):
    """csrc/pairlog.hip: the all-pairs mean of pairwize_logistic (utils.py:307-312 with the harness's negative_samples = 1) in one tiled
    pass -- e^f and e^-t once per score, every pair evaluated once into its row sum AND its column sum -- against the all-pairs
    kernels (one sweep per side) and the oracle; ragged sizes (padded workgroup ranges), scores beyond +-40 (the exact per-pair form
    inside the same kernel), a NaN score (the mean is NaN in the reference too) and run-to-run bit-identity (no atomics)."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(m_t * 3 + m_f)
    t, f = rs.uniform(-6, 9, m_t).astype(np.float32), rs.uniform(-8, 6, m_f).astype(np.float32)
    Lo, gto, gfo = O.loss_value_and_grads("pairwize_logistic", t.astype(np.float64), f.astype(np.float64), 1.0, 1)
    out = {}
    for mode in ("0", "1"):
        _pair_sort(mode)
        L, gt, gf = M.loss_value("pairwize_logistic", t, f, 1.0, 1, return_grads=True)
        out[mode] = (float(L.item()), gt.cpu().numpy(), gf.cpu().numpy())
        assert out[mode][0] == pytest.approx(float(Lo), rel=2e-6), mode
        np.testing.assert_allclose(out[mode][1], gto, rtol=2e-5, atol=1e-12, err_msg=mode)
        np.testing.assert_allclose(out[mode][2], gfo, rtol=2e-5, atol=1e-12, err_msg=mode)
    _pair_sort("1")
    L2, gt2, gf2 = M.loss_value("pairwize_logistic", t, f, 1.0, 1, return_grads=True)
    assert float(L2.item()) == out["1"][0] and np.array_equal(gt2.cpu().numpy(), out["1"][1]) and np.array_equal(gf2.cpu().numpy(), out["1"][2])
    if m_t == 5000:
        tb, fb = t.copy(), f.copy()
        tb[7], fb[11] = -45.0, 42.0   # beyond +-40: the per-pair form e^(f - t) (up to e^87, still a float32) inside the same kernel
        Lb, gtb, gfb = O.loss_value_and_grads("pairwize_logistic", tb.astype(np.float64), fb.astype(np.float64), 1.0, 1)
        L, gt, gf = M.loss_value("pairwize_logistic", tb, fb, 1.0, 1, return_grads=True)
        assert float(L.item()) == pytest.approx(float(Lb), rel=2e-6)
        np.testing.assert_allclose(gt.cpu().numpy(), gtb, rtol=2e-5, atol=1e-12)
        np.testing.assert_allclose(gf.cpu().numpy(), gfb, rtol=2e-5, atol=1e-12)
        tn = t.copy()
        tn[3] = np.nan
        L, gt, gf = M.loss_value("pairwize_logistic", tn, f, 1.0, 1, return_grads=True)
        assert np.isnan(float(L.item())) and np.all(np.isnan(gf.cpu().numpy())) and np.isnan(gt.cpu().numpy()[3])
