"""GPU corner tests of the Keras-semantics Adam the training kernels apply (tests.py:86 `optimizer='adam'` [TF-knowledge: lr 1e-3,
beta 0.9 / 0.999, epsilon 1e-7 OUTSIDE the bias correction, lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), dense moment decay and dense
update of every row; SURVEY 8(a) O1]) against oracle.AdamTF on every step form of the library: kge_train_step (plan built on the
spot), kge_train_step_planned and kge_train_step_swap (lr_t read from the device-resident step state).

What each test pins, at tests/tols.py's TOL_W = 1e-6 on the tables:
  * gradients of 1e-7 .. 1e-5 per element -- the range where epsilon = 1e-7 shapes the update (m / (sqrt(v) + eps) is neither
    sign(g) nor g / eps there): an epsilon inside the root or inside the bias correction moves those elements by 1e-5 .. 5e-4;
  * 1, 10 and 1000 updates -- lr_t's bias correction as the device step state advances it (1000: lr_t within 26 % of its limit);
  * one row touched once among 409 600 batch rows, next to entities no row touches at all (Adam moves those too once they have moments).
The oracle is the checker only."""
import numpy as np
import pytest
import torch

import oracle as O
from tols import TOL_W, TOL_L, TOL_LA, assert_tables

pytestmark = pytest.mark.gpu

FORMS = ["step", "planned", "swap"]


def _state(mdl):
    from kge_keras_amd import _lib
    sd = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device=mdl.device)
    o = mdl.optimizer
    od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, int(o.iterations) + 1, None, None, None, None, sd.data_ptr())
    _lib.check(_lib.lib.kge_opt_state_init(_lib.C.byref(od), _lib.stream_ptr()))
    mdl._step_dev = sd
    return sd


def _model(form, model, N, R, d, E0, R0, loss, m_pos, m_neg, **kw):
    from kge_keras_amd import models as M
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                            loss_function=loss, margin=1.0, **kw)
    mdl.compile("adam")
    mdl._ensure_train_state(m_pos, m_neg)
    sd = None
    if form != "step":
        sd = _state(mdl)  # device-resident step state: lr_t of every update is read from it
    if form == "swap":
        mdl._step_twins = (torch.empty_like(mdl.E), torch.empty_like(mdl.Rel))
    return mdl, sd


def _do_step(form, mdl, pos, neg):
    if form == "step":
        return float(mdl.train_step([pos, neg])["loss"].item())
    return float(mdl.train_step(mdl.plan_batch([pos, neg]))["loss"].item())


@pytest.mark.parametrize("form", FORMS)
def test_adam_gradients_between_1e_7_and_1e_5_pin_epsilon(form):
    """DistMult + pointwize_logistic (every row active, dL/dscore ~ 0.5 / m) on tables of U(0.01, 0.1): row products of 1e-4 .. 1e-2
    times 1e-3 = gradient elements of 1e-7 .. 1e-5.  Three steps against the oracle at 1e-6 -- and the same inputs through two WRONG
    Adam forms must miss the GPU tables by far more than that, i.e. the tolerance really tells them apart."""
    N, R, d, m = 400, 5, 32, 500
    rs = np.random.RandomState(11)
    E0 = rs.uniform(0.01, 0.1, (N, d)).astype(np.float32)
    R0 = rs.uniform(0.01, 0.1, (R, d)).astype(np.float32)
    mdl, _sd = _model(form, "DistMult", N, R, d, E0, R0, "pointwize_logistic", m, m)
    Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.make_optimizer("adam")
    tri = lambda n: np.stack([rs.permutation(N)[:n], rs.randint(0, R, n), rs.permutation(N)[:n]], 1).astype(np.int32)  # every entity at most twice per side
    for step in range(3):
        pos, neg = tri(m // 2), tri(m // 2)
        if step == 0:
            _L, _sp, _sn, gE, _gR = O.dense_grads("DistMult", Eo, Ro, pos, neg, "pointwize_logistic", 1.0)
            g = np.abs(gE[gE != 0])
            assert np.percentile(g, 5) > 5e-8 and np.percentile(g, 95) < 5e-5, (g.min(), g.max())  # the range this test is about
            lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
            # wrong form 1: epsilon inside the bias correction (m_hat / (sqrt(v_hat) + eps)); wrong form 2: epsilon under the root
            E_w1 = Eo - lr * gE / (np.abs(gE) + eps)
            E_w2 = Eo - lr * np.sqrt(1 - b2) / (1 - b1) * ((1 - b1) * gE) / np.sqrt((1 - b2) * gE * gE + eps)
        L = _do_step(form, mdl, pos, neg)
        Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        Eg = mdl.E.cpu().numpy()
        assert_tables(Eg, Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        if step == 0:
            assert np.max(np.abs(Eg - E_w1)) > 1e-4 and np.max(np.abs(Eg - E_w2)) > 1e-5
    mdl._raise_on_flag()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("updates", [1, 10, 1000])
def test_adam_bias_correction_after_1_10_1000_updates(form, updates):
    """lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) as the step count advances (host count for kge_train_step, the device step state for
    the planned forms): tables after `updates` steps against the oracle.  A smooth loss (pointwize_logistic), so that no row sits on a
    hinge where fp32 and fp64 could switch it differently; the same four batches in rotation."""
    N, R, d, m = 60, 4, 16, 96
    rs = np.random.RandomState(5)
    E0 = rs.uniform(-0.3, 0.3, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.3, 0.3, (R, d)).astype(np.float32)
    mdl, sd = _model(form, "DistMult", N, R, d, E0, R0, "pointwize_logistic", m, m)
    Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.make_optimizer("adam")
    tri = lambda n: np.stack([rs.randint(0, N - 6, n), rs.randint(0, R, n), rs.randint(0, N - 6, n)], 1).astype(np.int32)  # entities N-6.. never touched
    batches = [(tri(m), tri(m)) for _ in range(4)]
    plans = None
    if form != "step":
        plans = [mdl.plan_batch(list(b)) for b in batches]
    for t in range(updates):
        pos, neg = batches[t % 4]
        if form == "step":
            mdl.train_step([pos, neg])
        else:  # (a plan is consumed by its step: rebuilt into the same buffer)
            mdl.train_step(mdl.plan_batch([pos, neg], out=plans[t % 4]))
        O.train_step("DistMult", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0)
    # (1000 updates: every step's fp32 rounding of a few 1e-9 accumulates like a random walk -- 3e-6 leaves room for it and is still
    #  two orders below what a wrong bias correction does over that many steps)
    tol = TOL_W if updates <= 10 else 3e-6
    np.testing.assert_allclose(mdl.E.cpu().numpy(), Eo, atol=tol)
    np.testing.assert_allclose(mdl.Rel.cpu().numpy(), Ro, atol=tol)
    if sd is not None:
        assert int(sd[0].item()) == updates
        lr_next = float(sd.view(torch.float32)[2].item())
        assert lr_next == pytest.approx(oo.lr_t(updates + 1), rel=1e-6)
    # untouched entities moved only by the optimizer's own arithmetic on zero gradients: not at all (m = v = 0 -> update 0)
    np.testing.assert_array_equal(mdl.E.cpu().numpy()[N - 6:], E0[N - 6:])
    mdl._raise_on_flag()


@pytest.mark.parametrize("form", FORMS)
def test_adam_row_touched_once_among_409600(form):
    """The BASELINE batch size (204 800 + 204 800 rows) over 2 000 entities, with ONE entity that a single negative row touches as its
    object and nothing else does, and a band of entities nothing touches.  With the mean over 204 800 rows that row's gradient elements are
    ~1e-8 .. 6e-7 -- epsilon's linear regime, update ~ lr_t (1 - b1) g / eps = 316 g -- so a dropped or doubled contribution on that
    low-count row moves it by up to 1e-4 against the oracle; step 2 does not touch it again -- Keras' dense Adam still moves it (moments
    decay, update non-zero), a sparse ('lazy') Adam would leave it where it was."""
    N, R, d, m = 2000, 7, 24, 204800
    rs = np.random.RandomState(9)
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    lone, band = N - 1, slice(N - 50, N - 1)
    mdl, _sd = _model(form, "DistMult", N, R, d, E0, R0, "pointwize_logistic", m, m)
    Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.make_optimizer("adam")
    tri = lambda n: np.stack([rs.randint(0, N - 50, n), rs.randint(0, R, n), rs.randint(0, N - 50, n)], 1).astype(np.int32)
    for step in range(2):
        pos, neg = tri(m), tri(m)
        if step == 0:
            neg[123456, 2] = lone
        L = _do_step(form, mdl, pos, neg)
        Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0)
        assert L == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        Eg = mdl.E.cpu().numpy()
        assert_tables(Eg, Eo)
        assert_tables(mdl.Rel.cpu().numpy(), Ro)
        moved = np.abs(Eg[lone].astype(np.float64) - E0[lone])
        if step == 0:
            assert moved.max() > 2e-5 and moved.max() < 1e-3         # it moved, by ~316 g
            first = Eg[lone].copy()
        else:
            assert np.abs(Eg[lone] - first).max() > 2e-5             # dense Adam: the row keeps moving on its first moment
        np.testing.assert_array_equal(Eg[band], E0[band])            # rows with zero moments stay put exactly
    mdl._raise_on_flag()
