"""GPU tests of the model-surface holes closed in round 4 (all through the C ABI, checked against the oracle):
  * EmbeddingModel.call(x, training=True) / func(s, p, o, training=True): the Dropout layer on the three gathered rows is active
    (models.py:122,134-162) -- kge_score_training with the training kernels' counter-based masks, which the oracle regenerates;
  * batch_loss / evaluate / fit(validation_data=...): forward-only (kge_loss_only: scores + loss, no gradient half), for every model class
    of the north star including ConvE (models.py:239-275 in inference mode; tests.py:130-134 always passes validation_data);
  * ConvE.func on already-gathered rows (models.py:264-275)."""
import numpy as np
import pytest
import torch

import oracle as O
from tols import TOL, TOL_L, TOL_LA

pytestmark = pytest.mark.gpu

CASES = [("DistMult", {}, 200), ("DistMult", {}, 30), ("TransE", dict(norm=1, gamma=12), 40), ("TransE", dict(norm=2, gamma=12), 200),
         ("ComplEx", {}, 100), ("ComplEx", {}, 3), ("RotatE", {}, 24), ("pRotatE", {}, 24), ("HolE", {}, 32)]


def _mk(model, N, R, d, dp, seed=0, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, init_entities=E0, init_relations=R0, **kw)
    return mdl, E0, R0, rs


def _oracle_scores(model, mdl, E0, R0, t, kw, seed=None, rate=0.0):
    s, p, o = E0[t[:, 0]].astype(np.float64), R0[t[:, 1]].astype(np.float64), E0[t[:, 2]].astype(np.float64)
    if seed is not None:
        ms = [O.dropout_multipliers(seed, len(t), r, (R0 if r == 1 else E0).shape[1], rate, 0) for r in range(3)]
        s, p, o = s * ms[0], p * ms[1], o * ms[2]
    norm = mdl.modulus if model == "pRotatE" else kw.get("norm", 1)
    return O.score_rows(model, s, p, o, norm, kw.get("gamma", 12))


@pytest.mark.parametrize("model,kw,d", CASES)
def test_call_and_func_with_training_true_apply_dropout(model, kw, d):
    N, R, n, rate = 60, 5, 257, 0.3
    mdl, E0, R0, rs = _mk(model, N, R, d, rate, seed=3, **kw)
    t = np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1).astype(np.int32)
    plain = mdl(t).cpu().numpy()                      # training=False: no dropout
    np.testing.assert_allclose(plain, _oracle_scores(model, mdl, E0, R0, t, kw), atol=TOL, rtol=1e-5)
    seed0 = mdl._seed
    a = mdl(t, training=True).cpu().numpy()           # masks of step seed0 + 1
    assert mdl._seed == seed0 + 1
    np.testing.assert_allclose(a, _oracle_scores(model, mdl, E0, R0, t, kw, seed0 + 1, rate), atol=TOL, rtol=1e-5)
    b = mdl(t, training=True).cpu().numpy()           # a fresh set of masks every call, like Keras
    np.testing.assert_allclose(b, _oracle_scores(model, mdl, E0, R0, t, kw, seed0 + 2, rate), atol=TOL, rtol=1e-5)
    assert np.abs(a - plain).max() > 1e-3 and np.abs(a - b).max() > 1e-3
    f = mdl.func(E0[t[:, 0]], R0[t[:, 1]], E0[t[:, 2]], training=True).cpu().numpy()
    np.testing.assert_allclose(f, _oracle_scores(model, mdl, E0, R0, t, kw, seed0 + 3, rate), atol=TOL, rtol=1e-5)
    # rate 0: training=True changes nothing
    m0, _E, _R, _ = _mk(model, N, R, d, 0.0, seed=3, **kw)
    np.testing.assert_array_equal(m0(t, training=True).cpu().numpy(), m0(t).cpu().numpy())
    mdl._raise_on_flag()


LOSSES = ["pointwize_hinge", "pointwize_logistic", "pairwize_hinge", "pairwize_logistic", "pointwize_square_loss", "pairwize_square_loss"]


@pytest.mark.parametrize("loss", LOSSES)
@pytest.mark.parametrize("model,kw,d", [("DistMult", {}, 200), ("TransE", dict(norm=2, gamma=12), 40), ("ComplEx", {}, 16), ("HolE", {}, 32), ("RotatE", {}, 24)])
def test_batch_loss_is_forward_only_and_matches_oracle(model, kw, d, loss):
    """batch_loss: the oracle's loss on inference-mode scores (+ the l3 term); the model's tables, gradient accumulators and optimizer
    state are untouched -- no gradient half runs (the accumulators are never written, an uncompiled model stays uncompiled)."""
    N, R, m = 50, 4, 300
    mdl, E0, R0, rs = _mk(model, N, R, d, 0.2, seed=5, loss_function=loss, margin=0.7, regularization=0.01, **kw)
    pos = np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)
    neg = np.stack([rs.randint(0, N, 2 * m), rs.randint(0, R, 2 * m), rs.randint(0, N, 2 * m)], 1)
    if loss.startswith("pairwize") and loss != "pairwize_hinge":
        neg = neg[:m]
    sp = O.call(model, E0.astype(np.float64), R0.astype(np.float64), pos, kw.get("norm", 1), kw.get("gamma", 12))
    sn = O.call(model, E0.astype(np.float64), R0.astype(np.float64), neg, kw.get("norm", 1), kw.get("gamma", 12))
    Lo = O.loss_value(loss, sp, sn, 0.7, 1) + 0.01 * np.sum(np.abs(E0.astype(np.float64)) ** 3)
    L = float(mdl.batch_loss([pos, neg]).item())
    assert L == pytest.approx(Lo, rel=5 * TOL_L, abs=TOL_LA)
    assert mdl.optimizer is None and mdl._gE is None
    assert torch.equal(mdl.E.cpu(), torch.from_numpy(E0)) and torch.equal(mdl.Rel.cpu(), torch.from_numpy(R0))
    mdl._raise_on_flag()


def test_evaluate_sums_on_the_device_and_equals_mean_batch_loss(datasets):
    from kge_keras_amd.harness import DataGenerator, build_model
    np.random.seed(3)
    import random
    random.seed(3)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic", margin=1, seed=1,
                           regularization=0.001))
    gen = DataGenerator(datasets["nations_valid"], ns=10, batch_size=32)
    batches = [gen[i][0] for i in range(len(gen))]

    class Fixed:
        def __len__(self):
            return len(batches)

        def __getitem__(self, i):
            return batches[i], None
    ev = mdl.evaluate(Fixed())
    ref = np.mean([float(mdl.batch_loss(b).item()) for b in batches])
    assert ev == pytest.approx(ref, rel=1e-6)
    Eo, Ro = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    refo = np.mean([O.loss_value("pointwize_logistic", O.call("DistMult", Eo, Ro, b[0]), O.call("DistMult", Eo, Ro, b[1]), 1, 1) for b in batches])
    assert ev == pytest.approx(refo + 0.001 * np.sum(np.abs(Eo) ** 3), rel=5 * TOL_L)


def test_conve_fit_with_validation_data_batch_loss_and_func(datasets):
    """The reference script's own flow for ConvE (tests.py:121-134): fit(generator, validation_data=generator) -- val_loss per epoch from the
    forward-only loss in inference mode -- plus ConvE.func on gathered rows == call on ids, and batch_loss == the oracle's loss on the
    oracle's inference-mode scores."""
    from kge_keras_amd import models as M
    from kge_keras_amd.harness import DataGenerator
    import random
    N, R, d = 14, 55, 32
    mdl = M.ConvE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.2, hidden_dp=0.2, loss_function="pointwize_hinge", margin=1, seed=4)
    mdl.compile("adam")
    np.random.seed(1)
    random.seed(1)
    h = mdl.fit(DataGenerator(datasets["nations_train"], ns=4, batch_size=64), epochs=2, verbose=0,
                validation_data=DataGenerator(datasets["nations_valid"], ns=4, batch_size=64))
    assert len(h.history["val_loss"]) == 2 and all(np.isfinite(h.history["val_loss"])) and all(np.isfinite(h.history["loss"]))
    rs = np.random.RandomState(0)
    pos = np.stack([rs.randint(0, N, 90), rs.randint(0, R, 90), rs.randint(0, N, 90)], 1).astype(np.int32)
    neg = np.stack([rs.randint(0, N, 90), rs.randint(0, R, 90), rs.randint(0, N, 90)], 1).astype(np.int32)
    E, Rl = mdl.E.cpu().numpy(), mdl.Rel.cpu().numpy()
    P = {k: np.asarray(v, dtype=np.float64) for k, v in mdl.get_head_params().items()}
    ref = {}
    for name, t in (("pos", pos), ("neg", neg)):
        _x, ref[name] = O.conve_forward(P, E[t[:, 0]].astype(np.float64), Rl[t[:, 1]].astype(np.float64), E[t[:, 2]].astype(np.float64), False)
    np.testing.assert_allclose(mdl.predict(pos), ref["pos"], atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(mdl.func(E[pos[:, 0]], Rl[pos[:, 1]], E[pos[:, 2]]).cpu().numpy(), mdl.predict(pos), atol=1e-6)
    L = float(mdl.batch_loss([pos, neg]).item())
    assert L == pytest.approx(O.loss_value("pointwize_hinge", ref["pos"], ref["neg"], 1, 1), rel=1e-4, abs=1e-6)
    mdl._raise_on_flag()
