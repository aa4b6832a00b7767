"""Checkpoint round trips (save_weights / load_weights / get_weights / set_weights) and the evaluation-only loss: a resumed model
must predict and train exactly like the one that was saved -- tables, every head / literal tensor, batch-norm statistics, all
optimizer slots, the step count."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA  # noqa: E402  (tests/tols.py: what each constant bounds)


def _tri(rs, n, N, R):
    return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)


def _make(kind, seed=11):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    N, R, d = 40, 5, 16
    if kind == "DistMult":
        return M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=seed, loss_function="pointwize_logistic"), N, R, rs
    if kind == "ConvE":
        return M.ConvE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, hidden_dp=0.0, seed=seed, loss_function="pointwize_logistic"), N, R, rs
    if kind == "HAKE":
        return M.HAKE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=seed, loss_function="pointwize_logistic"), N, R, rs
    if kind == "RotatE":
        return M.RotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=seed, loss_function="pointwize_logistic"), N, R, rs
    lit = np.random.RandomState(3).rand(N, 5).astype(np.float32)
    return M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=seed, literals=lit, literal_activation="tanh",
                      loss_function="pairwize_hinge"), N, R, rs


@pytest.mark.parametrize("kind", ["DistMult", "ConvE", "literal", "RotatE", "HAKE"])
def test_save_load_round_trip(kind, tmp_path):
    a, N, R, rs = _make(kind)
    a.compile("adam")
    batches = [(_tri(rs, 200, N, R), _tri(rs, 200, N, R)) for _ in range(4)]
    for pos, neg in batches[:2]:
        a.train_step([pos, neg])
    path = str(tmp_path / "ck.pt")
    a.save_weights(path)
    b, _, _, _ = _make(kind, seed=99)  # different initialisation everywhere
    b.compile("adam")
    b.load_weights(path)
    q = _tri(rs, 300, N, R)
    assert np.array_equal(a.predict(q), b.predict(q))
    assert b.optimizer.iterations == 2
    for pos, neg in batches[2:]:
        la, lb = float(a.train_step([pos, neg])["loss"].item()), float(b.train_step([pos, neg])["loss"].item())
        assert la == pytest.approx(lb, rel=1e-6, abs=1e-7)
    np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(a.Rel.cpu().numpy(), b.Rel.cpu().numpy(), atol=1e-6)
    for x, y in zip(a.get_weights(), b.get_weights()):
        np.testing.assert_allclose(x, y, atol=1e-6)
    # get_weights / set_weights are symmetric
    c, _, _, _ = _make(kind, seed=5)
    c.set_weights(a.get_weights())
    assert np.array_equal(a.predict(q), c.predict(q))
    with pytest.raises(ValueError):
        c.set_weights(a.get_weights()[:1])


def test_load_rejects_mismatches(tmp_path):
    from kge_keras_amd import models as M
    a, N, R, rs = _make("DistMult")
    a.compile("adam")
    a.train_step([_tri(rs, 50, N, R), _tri(rs, 50, N, R)])
    path = str(tmp_path / "ck.pt")
    a.save_weights(path)
    other = M.TransE(e_dim=16, r_dim=16, num_entities=N, num_relations=R, dp=0.0)
    with pytest.raises(ValueError, match="holds a DistMult"):
        other.load_weights(path)
    small = M.DistMult(e_dim=8, r_dim=8, num_entities=N, num_relations=R, dp=0.0)
    with pytest.raises(ValueError, match="shape"):
        small.load_weights(path)
    sgd, _, _, _ = _make("DistMult")
    sgd.compile("sgd")
    with pytest.raises(ValueError, match="optimizer"):
        sgd.load_weights(path)
    # optimizer state in the file, model not compiled: the tables load, and that is said out loud
    fresh, _, _, _ = _make("DistMult", seed=8)
    with pytest.warns(UserWarning, match="not compiled"):
        fresh.load_weights(path)
    assert torch.equal(fresh.E, a.E) and fresh.optimizer is None


def test_format_1_checkpoints_still_load(tmp_path):
    """The first checkpoint layout of this package ({E, Rel, model, opt}: no `format`, `extra`, `seed` keys) is read as format 1."""
    a, N, R, rs = _make("DistMult")
    a.compile("adam")
    a.train_step([_tri(rs, 50, N, R), _tri(rs, 50, N, R)])
    path = str(tmp_path / "old.pt")
    torch.save({"E": a.E.cpu(), "Rel": a.Rel.cpu(), "model": "DistMult",
                "opt": dict(kind="adam", iterations=a.optimizer.iterations, slots=[s.cpu() for s in a.optimizer.slots])}, path)
    b, _, _, _ = _make("DistMult", seed=6)
    b.compile("adam")
    b.load_weights(path)
    assert torch.equal(a.E, b.E) and torch.equal(a.Rel, b.Rel) and b.optimizer.iterations == a.optimizer.iterations
    assert all(torch.equal(x, y) for x, y in zip(a.optimizer.slots, b.optimizer.slots))
    batch = [_tri(rs, 50, N, R), _tri(rs, 50, N, R)]
    a.train_step(batch)
    b.train_step(batch)
    assert torch.equal(a.E, b.E)
    lit, _, _, _ = _make("literal")
    with pytest.raises(ValueError, match="format-1"):
        lit.load_weights(path)
    torch.save({"format": 3, "E": a.E.cpu(), "Rel": a.Rel.cpu(), "model": "DistMult"}, path)
    with pytest.raises(ValueError, match="newer"):
        b.load_weights(path)


def test_val_loss_goes_through_the_literal_transform():
    """batch_loss (what fit() reports as val_loss) of a model with literals == the loss of the literal forward without dropout
    (models.py:152-158), not the loss of the bare tables; and it changes nothing."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(2)
    N, R, d, L, m = 30, 4, 16, 5, 200
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    lit = rs.rand(N, L).astype(np.float32)
    mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, init_entities=E0, init_relations=R0, literals=lit,
                     literal_activation="tanh", dp=0.3, loss_function="pairwize_hinge", margin=0.5)
    H = O.literal_init(d, L, rs)
    H["b"] += rs.uniform(-0.1, 0.1, d).astype(np.float32)
    mdl.set_literal_dense(H["W"], H["b"])
    mdl.compile("adam")
    pos, neg = _tri(rs, m, N, R), _tri(rs, m, N, R)
    got = float(mdl.batch_loss([pos, neg]).item())
    Hd = {k: np.asarray(v, dtype=np.float64).copy() for k, v in H.items()}
    ref = O.literal_train_step("DistMult", Hd, "tanh", E0.astype(np.float64), R0.astype(np.float64), lit.astype(np.float64), O.AdamTF(), pos, neg,
                               "pairwize_hinge", 0.5, 1)[0]
    assert got == pytest.approx(ref, rel=TOL_L, abs=TOL_LA)
    bare = O.dense_grads("DistMult", E0.astype(np.float64), R0.astype(np.float64), pos, neg, "pairwize_hinge", 0.5)[0]
    assert abs(bare - ref) > 1e-3  # the two models really differ
    assert np.array_equal(mdl.E.cpu().numpy(), E0) and float(mdl._gall.abs().max()) == 0.0 and float(mdl._lit["gtheta"].abs().max()) == 0.0
    # with the L3 regulariser the value carries its term too
    mdl.regularization = 0.01
    got2 = float(mdl.batch_loss([pos, neg]).item())
    assert got2 == pytest.approx(ref + 0.01 * np.sum(np.abs(E0.astype(np.float64)) ** 3), abs=TOL, rel=1e-4)
