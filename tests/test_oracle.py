"""Oracle vs the reference-generated golden vectors, known-answer tests and torch-CPU autograd (CPU only)."""
import hashlib
import random

import numpy as np
import pytest
import torch

import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ sampler restatement vs reference vectors
def test_generate_negative_golden(golden):
    g = golden("generate_negative")
    for c in g["cases"]:
        np.random.seed(c["seed"])
        neg = O.generate_negative(np.asarray(c["kg"]), c["N"], c["ns"])
        assert list(neg.shape) == c["shape"] and str(neg.dtype) == c["dtype"]
        assert neg.reshape(-1, 3).tolist() == c["neg"]
        assert np.random.randint(0, 2**31 - 1, size=2).tolist() == c["after"]
    for c in g["big"]:
        rs = np.random.RandomState(c["kg_seed"])
        kg = np.stack([rs.randint(0, c["N"], c["B"]), rs.randint(0, 237, c["B"]), rs.randint(0, c["N"], c["B"])], axis=1)
        np.random.seed(c["seed"])
        neg = O.generate_negative(kg, c["N"], c["ns"]).reshape(-1, 3)
        assert sha(neg.astype(np.int64)) == c["sha256_int64"]
        assert neg[:8].tolist() == c["head"] and neg[-8:].tolist() == c["tail"]


def test_oversample_golden(golden):
    g = golden("oversample")
    for c in g["cases"]:
        random.seed(c["seed"])
        neg = np.asarray(c["neg"]).reshape(-1, 3, 1)
        pos_os, neg_os = O.oversample_data([np.asarray(c["kg"]), neg])
        assert pos_os.tolist() == c["pos_os"] and neg_os.tolist() == c["neg_os"]
        assert [random.getrandbits(32) for _ in range(2)] == c["after"]


def test_shuffle_golden(golden):
    for c in golden("shuffle")["cases"]:
        np.random.seed(c["seed"])
        idx = O.epoch_permutation(c["n"])
        assert sha(idx.astype(np.int64)) == c["sha256_int64"]
        if c["perm"] is not None:
            assert idx.tolist() == c["perm"]


def test_batch_stream_golden(golden, datasets):
    """tests.py:43-61 order: epoch shuffle (numpy) -> generate_negative (numpy) -> oversample (python random)."""
    g = golden("batch_stream")
    tr = [tuple(int(x) for x in t) for t in datasets["nations_train"]]
    np.random.seed(g["np_seed"])
    random.seed(g["py_seed"])
    idx = O.epoch_permutation(len(tr))
    assert len(tr) // g["B"] == g["steps_per_epoch"]
    for b, ref in enumerate(g["batches"]):
        assert idx[b * g["B"]:(b + 1) * g["B"]].tolist() == ref["index"]
        pos_os, neg_os = O.harness_batch(tr, idx, b, g["B"], g["ns"], g["num_e"])
        assert pos_os.tolist() == ref["pos"] and neg_os.tolist() == ref["neg"]


# ------------------------------------------------------------------ evaluator restatement vs reference vectors
class _DM:
    def __init__(self, N, R, d, seed, quant):
        rs = np.random.RandomState(seed)
        self.E = rs.uniform(-1, 1, (N, d))
        self.Rel = rs.uniform(-1, 1, (R, d))
        if quant:
            self.E = np.round(self.E * quant) / quant
            self.Rel = np.round(self.Rel * quant) / quant

    def predict(self, t):
        return O.call("DistMult", self.E, self.Rel, t)


def _ids(a):
    return [tuple(int(x) for x in t) for t in a]


def test_validate_golden_literal_and_closed_form(golden, datasets):
    for c in golden("validate")["cases"]:
        m = c["model"]
        fm = _DM(m["N"], m["R"], m["d"], m["seed"], m["quant"])
        test = [tuple(t) for t in c["test"]]
        ds = c["dataset"]
        tes = set(_ids(datasets[ds + "_test"]))
        filt = [t for t in _ids(datasets[ds + "_train"]) + _ids(datasets[ds + "_valid"]) if t not in tes]
        assert sha(np.asarray(filt, dtype=np.int64)) == c["filt_sha"]
        t2, h2 = O.ranks_closed_form("DistMult", fm.E, fm.Rel, test, filt)
        met2 = O.metrics_from_ranks(t2, h2)
        for k, v in c["metrics"].items():
            assert met2[k] == pytest.approx(v, rel=0, abs=1e-12), k
        if ds != "nations":
            continue  # the literal O(N)-list restatement is for small N only
        met, tails, heads = O.validate(fm.predict, test, m["N"], filt)
        for k, v in c["metrics"].items():
            assert met[k] == pytest.approx(v, rel=0, abs=1e-12), k
        assert t2.tolist() == tails.tolist() and h2.tolist() == heads.tolist()
        ft = O.build_filters(filt)
        for i, rows in enumerate(c["tails"]):
            assert O.candidate_rows(test[i], m["N"], "tail", ft[1]).tolist() == rows
        for i, rows in enumerate(c["heads"]):
            assert O.candidate_rows(test[i], m["N"], "head", ft[0]).tolist() == rows


def test_validate_error_behaviour(golden):
    errs = golden("misc")["misc"]["errors"]
    assert errs["validate_filter_contains_test"] == "ValueError"
    fm = _DM(14, 55, 4, 5, None)
    with pytest.raises(ValueError):
        O.validate(fm.predict, [(0, 1, 2)], 14, [(0, 1, 2)])
    with pytest.raises(KeyError):
        O.loss_value_and_grads("nope", np.zeros(2), np.zeros(2))


# ------------------------------------------------------------------ known-answer tests from the source formulas
def test_kats():
    s, p, o = (np.array([[1., 2.]]), np.array([[3., 4.]]), np.array([[5., 6.]]))
    assert O.func_distmult(s, p, o)[0] == 63.0  # models.py:172
    assert O.func_transe(np.array([[1., 2.]]), np.array([[1., 1.]]), np.array([[0., 5.]]), 1, 12)[0] == 8.0  # |2|+|-2| -> 12-4
    assert O.func_transe(np.array([[3., 0.]]), np.array([[0., 4.]]), np.array([[0., 0.]]), 2, 12)[0] == 7.0
    assert O.func_transe(np.array([[3., 0.]]), np.array([[0., 4.]]), np.array([[0., 0.]]), 2, 0)[0] == 5.0  # sign flip
    # ComplEx width 2: s=1+2i p=3+4i o=5+6i : Re(s*p*conj(o)) = 35
    assert O.func_complex(np.array([[1., 2.]]), np.array([[3., 4.]]), np.array([[5., 6.]]))[0] == 35.0
    z = complex(1, 2) * complex(3, 4) * complex(5, -6)
    assert z.real == 35.0
    # ComplEx with zero imaginary halves == DistMult
    rs = np.random.RandomState(0)
    a, b, c = (np.concatenate([rs.randn(5, 8), np.zeros((5, 8))], 1) for _ in range(3))
    np.testing.assert_allclose(O.func_complex(a, b, c), O.func_distmult(a[:, :8], b[:, :8], c[:, :8]))
    # HolE: direct definition
    a, b, c = rs.randn(3, 6), rs.randn(3, 6), rs.randn(3, 6)
    x = np.array([[sum(a[n, i] * c[n, (i + k) % 6] for i in range(6)) for k in range(6)] for n in range(3)])
    np.testing.assert_allclose(O.func_hole(a, b, c), np.sum(b * x, 1), atol=1e-12)


def test_loss_kats():
    t = np.array([0.5, 2.0])
    f = np.array([-2.0, 0.0])
    assert O.loss_value("pointwize_hinge", t, f, 1) == pytest.approx((0.5 + 0) / 2 + (0 + 1) / 2)
    assert O.loss_value("pointwize_logistic", t, f) == pytest.approx(
        np.mean(np.log(1e-6 + 1 + np.exp(-t))) + np.mean(np.log(1e-6 + 1 + np.exp(f))))
    # pairwise with negative_samples=1 is the (m, m) broadcast (tests.py:81)
    L = O.loss_value("pairwize_hinge", t, f, 1, 1)
    full = np.maximum(1 + f[:, None] - t[None, :], 0)
    assert full.shape == (2, 2) and L == pytest.approx(full.mean())
    # aligned form when ns == len(true)
    L2 = O.loss_value("pairwize_hinge", t, np.array([0., 1., 2., 3.]), 1, 2)
    assert L2 == pytest.approx(np.maximum(1 + np.array([[0., 1.], [2., 3.]]) - t, 0).mean())


@pytest.mark.parametrize("name", O.LOSSES)
def test_loss_grads_vs_torch(name):
    rs = np.random.RandomState(1)
    t = rs.uniform(0.05, 0.95, 6)
    f = rs.uniform(0.05, 0.95, 6)
    ns = 1
    L, gt, gf = O.loss_value_and_grads(name, t, f, 0.7, ns)
    tt = torch.tensor(t, requires_grad=True)
    ff = torch.tensor(f, requires_grad=True)
    eps = 1e-6
    if name == "pointwize_hinge":
        l = torch.relu(0.7 - tt).mean() + torch.relu(0.7 + ff).mean()
    elif name == "pointwize_logistic":
        l = torch.log(eps + 1 + torch.exp(-tt)).mean() + torch.log(eps + 1 + torch.exp(ff)).mean()
    elif name == "pointwize_square_loss":
        l = ((0.7 - tt) ** 2).mean() + ((0.7 + ff) ** 2).mean()
    elif name == "pointwize_cross_entropy":
        l = -(torch.log(tt.clamp(1e-7, 1 - 1e-7) + 1e-7)).mean() - (torch.log(1 - ff.clamp(1e-7, 1 - 1e-7) + 1e-7)).mean()
    elif name == "pairwize_hinge":
        l = torch.relu(0.7 + ff.reshape(-1, ns) - tt).mean()
    elif name == "pairwize_logistic":
        l = torch.log(eps + 1 + torch.exp(ff.reshape(-1, ns) - tt)).mean()
    else:
        l = (-(ff.reshape(-1, ns) - tt) ** 2).mean()
    l.backward()
    assert L == pytest.approx(l.item(), rel=1e-12)
    np.testing.assert_allclose(gt, tt.grad.numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(gf, ff.grad.numpy(), rtol=1e-10, atol=1e-14)


def _torch_score(model, s, p, o, norm, gamma):
    if model == "DistMult":
        return (s * p * o).sum(-1)
    if model == "TransE":
        n = torch.linalg.vector_norm(s + p - o, ord=norm, dim=1)
        return gamma - n if gamma > 0 else n
    if model == "ComplEx":
        h = s.shape[1] // 2
        sr, si, pr, pi, orr, oi = s[:, :h], s[:, h:], p[:, :h], p[:, h:], o[:, :h], o[:, h:]
        return (sr * pr * orr + pr * si * oi + pi * sr * oi - pi * si * orr).sum(-1)
    if model == "HolE":
        x = torch.fft.ifft(torch.conj(torch.fft.fft(s)) * torch.fft.fft(o)).real
        return (p * x).sum(-1)
    if model == "pRotatE":  # models.py:479-489 op for op; `norm` carries the modulus product of :476
        ph = lambda x: torch.atan2(torch.sin(x), torch.cos(x))  # noqa: E731
        r = torch.abs(torch.sin((ph(s) + ph(p) - ph(o)) / 2)).sum(1) * norm
        return gamma - r if gamma > 0 else r
    if model == "RotatE":  # models.py:439-459 op for op
        h = s.shape[1] // 2
        re_s, im_s, re_o, im_o = s[:, :h], s[:, h:], o[:, :h], o[:, h:]
        phase = torch.atan2(torch.sin(p), torch.cos(p))
        re_r, im_r = torch.cos(phase), torch.sin(phase)
        re = re_s * re_r - im_s * im_r - re_o
        im = re_s * im_r + im_s * re_r - im_o
        sc = torch.cat([re, im], 1).sum(1)
        return gamma - sc if gamma > 0 else sc


@pytest.mark.parametrize("model,norm,gamma", [("DistMult", 1, 12), ("TransE", 1, 12), ("TransE", 2, 12), ("TransE", 2, 0), ("TransE", 3, 12), ("TransE", 1.5, 0),
                                              ("ComplEx", 1, 12), ("HolE", 1, 12), ("pRotatE", 0.35, 12), ("pRotatE", 0.35, 0),
                                              ("RotatE", 2, 12), ("RotatE", 2, 0)])
@pytest.mark.parametrize("loss", ["pointwize_hinge", "pointwize_logistic", "pairwize_hinge"])
def test_dense_grads_and_adam_vs_torch(model, norm, gamma, loss):
    rs = np.random.RandomState(3)
    N, R, d, m = 11, 4, 6, 40
    de, dr = O.table_dims(model, d, d)
    amp = 4.0 if model in ("pRotatE", "RotatE") else 0.6  # phases beyond (-pi, pi]: the wrap of models.py:481-483 / :443 is exercised
    E = rs.uniform(-amp, amp, (N, de))
    Rel = rs.uniform(-amp, amp, (R, dr))
    pos = np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)
    neg = np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)
    L, sp, sn, gE, gR = O.dense_grads(model, E, Rel, pos, neg, loss, 1.0, 1, norm, gamma, regularization=0.01)
    Et = torch.tensor(E, requires_grad=True)
    Rt = torch.tensor(Rel, requires_grad=True)
    tp = _torch_score(model, Et[pos[:, 0]], Rt[pos[:, 1]], Et[pos[:, 2]], norm, gamma)
    tn = _torch_score(model, Et[neg[:, 0]], Rt[neg[:, 1]], Et[neg[:, 2]], norm, gamma)
    if loss == "pointwize_hinge":
        l = torch.relu(1 - tp).mean() + torch.relu(1 + tn).mean()
    elif loss == "pointwize_logistic":
        l = torch.log(1e-6 + 1 + torch.exp(-tp)).mean() + torch.log(1e-6 + 1 + torch.exp(tn)).mean()
    else:
        l = torch.relu(1 + tn.reshape(-1, 1) - tp).mean()
    l = l + 0.01 * (Et.abs() ** 3).sum()
    l.backward()
    np.testing.assert_allclose(sp, tp.detach().numpy(), atol=1e-12)
    assert L == pytest.approx(l.item(), rel=1e-11)
    np.testing.assert_allclose(gE, Et.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(gR, Rt.grad.numpy(), atol=1e-12)


def test_adam_tf_formula():
    # two steps by hand.  [TF] the hyper-parameters are float32 tensors (OptimizerV2._get_hyper(name, var_dtype)): 0.9, 0.999, 1e-3 and
    # 1e-7 enter the formula rounded to float32
    lr, b1, b2, eps = (float(np.float32(x)) for x in (1e-3, 0.9, 0.999, 1e-7))
    w = np.array([[1.0, -2.0]])
    g1 = np.array([[0.5, 0.0]])
    opt = O.AdamTF()
    opt.begin_step()
    opt.apply("w", w, g1)
    m = (1 - b1) * g1
    v = (1 - b2) * g1 ** 2
    lr1 = lr * np.sqrt(1 - b2) / (1 - b1)
    exp = np.array([[1.0, -2.0]]) - lr1 * m / (np.sqrt(v) + eps)
    np.testing.assert_allclose(w, exp, rtol=1e-14)
    # the closed form of step 1 (m / sqrt(v) = sqrt(1 - b2) / (1 - b1) ... ) with the decimal constants, to the precision float32 constants allow
    np.testing.assert_allclose(w, np.array([[1.0, -2.0]]) - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g1) / (np.sqrt(0.001 * g1 ** 2) + 1e-7), rtol=1e-7)
    # untouched element still decays its moments on later steps (dense semantics)
    opt.begin_step()
    opt.apply("w", w, np.zeros_like(w))
    m2 = b1 * m
    v2 = b2 * v
    lr2 = lr * np.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    np.testing.assert_allclose(w, exp - lr2 * m2 / (np.sqrt(v2) + eps), rtol=1e-14)
    assert opt.lr_t(2) == pytest.approx(1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2), rel=1e-5)  # (float32 beta_2: 6.6e-6 off the decimal form at t = 2)


def test_glorot_limits():
    rs = np.random.RandomState(0)
    E = O.glorot_uniform(14541, 200, rs)
    assert abs(np.abs(E).max() - np.sqrt(6 / (14541 + 200))) < 1e-4 and E.dtype == np.float32


def test_dropout_multipliers_statistics_and_determinism():
    m = O.dropout_multipliers(7, 400, 1, 200, 0.2)
    assert m.shape == (400, 200)
    assert set(np.unique(m).tolist()) == {0.0, float(np.float32(1) / (np.float32(1) - np.float32(0.2)))}
    assert abs((m > 0).mean() - 0.8) < 0.01  # keep probability
    assert abs(m.mean() - 1.0) < 0.02        # inverted dropout is unbiased
    assert np.array_equal(m, O.dropout_multipliers(7, 400, 1, 200, 0.2))
    assert not np.array_equal(m, O.dropout_multipliers(8, 400, 1, 200, 0.2))
    assert not np.array_equal(m, O.dropout_multipliers(7, 400, 2, 200, 0.2))
    # row_offset addresses the same batch rows
    assert np.array_equal(m[100:], O.dropout_multipliers(7, 300, 1, 200, 0.2, row_offset=100))
    assert (O.dropout_multipliers(1, 10, 0, 31, 0.0) == 1.0).all()


@pytest.mark.parametrize("model,norm,gamma", [("DistMult", 1, 12), ("TransE", 1, 12), ("TransE", 2, 12), ("TransE", 2, 0), ("ComplEx", 1, 12)])
@pytest.mark.parametrize("loss", ["pointwize_hinge", "pointwize_logistic"])
@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_c_port_matches_numpy_oracle(model, norm, gamma, loss, opt):
    """oracle/kge_oracle_c.c (the cpu_baseline port) against the NumPy restatement, three steps."""
    from oracle import c_port
    rs = np.random.RandomState(4)
    N, R, d, m = 50, 6, 16, 500
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    ct = c_port.CTrainer(model, E0, R0, opt, loss, 0.7, norm, gamma)
    Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.make_optimizer(opt)
    for _ in range(3):
        pos = np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)
        neg = np.stack([rs.randint(0, N, m), rs.randint(0, R, m), rs.randint(0, N, m)], 1)
        Lc = ct.step(pos, neg)
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 0.7, 1, norm, gamma)
        assert Lc == pytest.approx(Lo, abs=1e-4, rel=1e-5)
        np.testing.assert_allclose(ct.E, Eo, atol=1e-4)
        np.testing.assert_allclose(ct.Rel, Ro, atol=1e-4)
    assert c_port.threads() >= 1


def test_protate_known_answers():
    """pRotatE (models.py:461-489): doubled widths, modulus = modulus * (gamma + epsilon) / e_dim / 2 on the doubled e_dim,
    score = gamma - modulus * sum |sin((s + p - o) / 2)|, phases wrapped first (no effect under |sin|)."""
    assert O.table_dims("pRotatE", 5, 7) == (10, 14)
    assert O.protate_modulus(8, gamma=12, epsilon=2, modulus=0.5) == pytest.approx(0.5 * 14 / 8 / 2)
    s = np.array([[np.pi, 0.0]]); p = np.array([[0.0, np.pi / 3]]); o = np.array([[0.0, 0.0]])
    assert O.func_protate(s, p, o, 2.0, 12)[0] == pytest.approx(12 - 2.0 * (1.0 + 0.5))
    assert O.func_protate(s, p, o, 2.0, 0)[0] == pytest.approx(2.0 * 1.5)
    assert O.func_protate(s + 2 * np.pi, p - 4 * np.pi, o + 6 * np.pi, 2.0, 12)[0] == pytest.approx(12 - 3.0)  # wrap invariance


def test_hake_restatement_vs_torch_autograd_and_known_answer():
    """HAKE exactly as written (models.py:362-400): phase part an L1 norm per row, modulus part ONE tf.norm over the whole tensor handed
    to the call -- scores and the gradients of sum(g * score) against torch autograd (clamp / where / abs branches included), a
    hand-computed value, and the batch dependence itself: a row scored alone != the same row scored in a batch."""
    torch = pytest.importorskip("torch")
    rs = np.random.RandomState(0)
    m, h = 17, 6
    s, o = rs.uniform(-1, 1, (m, 2 * h)), rs.uniform(-1, 1, (m, 2 * h))
    p = rs.uniform(-1.5, 1.5, (m, 3 * h))
    p[:, 2 * h:] *= 2  # biases beyond 1 and below -|mod_p|
    prm = O.hake_params(2 * h, 12, 2, 0.5, 1.0)
    g = rs.uniform(-1, 1, m)
    sc = O.score_rows("HAKE", s, p, o, prm, 12)
    ds, dp, do = O.score_row_grads("HAKE", s, p, o, g, prm, 12)
    S, P, Ob = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (s, p, o)]
    pw, mw, rng = prm
    scale = rng / np.pi
    phs, ms = S[:, :h] / scale, S[:, h:]
    pho, mo = Ob[:, :h] / scale, Ob[:, h:]
    php, mp, bp = P[:, :h] / scale, P[:, h:2 * h], P[:, 2 * h:]
    b = torch.clamp(bp, max=1.0)
    b = torch.where(b < -mp.abs(), -mp.abs(), b)
    out = 12 - (pw * torch.sin((phs + php - pho) / 2).abs().sum(-1) + mw * torch.linalg.vector_norm(ms * (mp + b) - mo.abs() * (1 - b)))
    assert np.allclose(out.detach().numpy(), sc, atol=1e-12)
    (out * torch.tensor(g)).sum().backward()
    for a, t in ((ds, S.grad), (dp, P.grad), (do, Ob.grad)):
        assert np.allclose(a, t.numpy(), atol=1e-10)
    # known answer (h = 1, one row): phases 0 -> p_score 0; M = 2*(3 + 0.5) - |-4|*(1 - 0.5) = 5 -> score = 12 - 5 = 7
    one = O.func_hake(np.array([[0.0, 2.0]]), np.array([[0.0, 3.0, 0.5]]), np.array([[0.0, -4.0]]), (0.5, 1.0, 1.0), 12)
    assert one == pytest.approx([7.0])
    # batch dependence: the modulus norm is taken over all rows of the call
    assert not np.allclose(O.score_rows("HAKE", s[:1], p[:1], o[:1], prm, 12), sc[:1])
    assert np.allclose(O.call("HAKE", s, p, np.stack([np.arange(m)] * 3, 1) * 0 + np.array([[0, 0, 0]]), prm, 12)[:1], O.score_rows("HAKE", s[:1].repeat(m, 0), p[:1].repeat(m, 0), s[:1].repeat(m, 0), prm, 12)[:1])
