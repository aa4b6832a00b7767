"""GPU tests of kge_train_step_swap: the planned step between twin tables (self-contained entity half with the optimizer applied by
the wave that summed a row's gradient, relation half beside it) against kge_train_step_planned and the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)

CASES = [("DistMult", {}, 200), ("DistMult", {}, 24), ("TransE", dict(norm=1, gamma=12), 40), ("TransE", dict(norm=2, gamma=12), 200),
         ("ComplEx", {}, 100), ("ComplEx", {}, 8), ("DistMult", {}, 256), ("ComplEx", {}, 200), ("pRotatE", {}, 100), ("pRotatE", {}, 24)]


def _state(mdl):
    from kge_keras_amd import _lib
    sd = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device=mdl.device)
    o = mdl.optimizer
    od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, int(o.iterations) + 1, None, None, None, None, sd.data_ptr())
    _lib.check(_lib.lib.kge_opt_state_init(_lib.C.byref(od), _lib.stream_ptr()))
    mdl._step_dev = sd
    return sd


@pytest.mark.parametrize("opt", ["adam", "sgd"])
@pytest.mark.parametrize("model,kw,d", CASES)
def test_swap_step_equals_planned_step_and_oracle(model, kw, d, opt):
    """Three steps on batches with a hub entity (a list of more than 128 entries: chunk waves + ticket), entities that no row touches
    (Adam still moves them: dense update, Keras semantics) and a dominant relation (many partial rows).  Twin step == three-launch
    planned step to fp32 rounding, both == oracle within 1e-4; the input tables of a twin step are left as they were; the step state
    (count, loss sum) advances the same way."""
    from kge_keras_amd import models as M
    N, R = 300, 6
    rs = np.random.RandomState(3)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    a, b = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                              loss_function="pointwize_hinge", margin=2.0, **kw) for _ in range(2))
    for m in (a, b):
        m.compile(opt)
        m._ensure_train_state(64, 1500)
    sa, sb = _state(a), _state(b)
    a._step_twins = (torch.empty_like(a.E), torch.empty_like(a.Rel))
    from kge_keras_amd import _lib
    md, ld, od = a._model_desc(True), a._loss_desc(), a._opt_desc()
    assert _lib.lib.kge_train_step_swap_supported(_lib.C.byref(md), _lib.C.byref(ld), _lib.C.byref(od)) == 1
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    tot = 0.0
    for step in range(3):
        n_pos, n_neg = 64, 1500
        ent = lambda n: np.where(rs.rand(n) < 0.3, 7, rs.randint(0, N - 40, n))  # hub: entity 7; entities N-40.. never touched
        rel = lambda n: np.where(rs.rand(n) < 0.6, 2, rs.randint(0, R, n))
        pos = np.stack([ent(n_pos), rel(n_pos), ent(n_pos)], 1).astype(np.int32)
        neg = np.stack([ent(n_neg), rel(n_neg), ent(n_neg)], 1).astype(np.int32)
        La = float(a.train_step(a.plan_batch([pos, neg]))["loss"].item())
        Lb = float(b.train_step(b.plan_batch([pos, neg]))["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_hinge", 2.0, 1, a.modulus if model == "pRotatE" else kw.get("norm", 1), kw.get("gamma", 12))
        tot += La
        assert La == pytest.approx(Lb, rel=1e-6, abs=1e-7) and La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(a.Rel.cpu().numpy(), b.Rel.cpu().numpy(), atol=2e-6)
        assert_tables(a.E.cpu().numpy(), Eo)
        assert_tables(a.Rel.cpu().numpy(), Ro)
    assert int(sa[0].item()) == int(sb[0].item()) == 3
    fa, fb = sa.view(torch.float32).cpu().numpy(), sb.view(torch.float32).cpu().numpy()
    assert fa[2] == fb[2]                                  # learning rate of the next update
    la, lb = float(sa.view(torch.float64)[2].item()), float(sb.view(torch.float64)[2].item())  # loss sum (a double)
    assert la == pytest.approx(tot, rel=1e-7) and lb == pytest.approx(tot, rel=1e-5)
    a._raise_on_flag()
    # the plans' counters are left zero: a fourth build with "counters known to be zero" (parts | 8) still trains right
    pos = np.stack([rs.randint(0, N, 32), rs.randint(0, R, 32), rs.randint(0, N, 32)], 1).astype(np.int32)
    neg = np.stack([rs.randint(0, N, 1500), rs.randint(0, R, 1500), rs.randint(0, N, 1500)], 1).astype(np.int32)
    pa = a.plan_batch([pos, neg])
    a.train_step(pa)
    a.train_step(a.plan_batch([pos, neg], out=pa, parts=7 | 8))
    for _ in range(2):
        b.train_step(b.plan_batch([pos, neg]))
    np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=4e-6)


def test_swap_leaves_input_tables_and_rejects_aliases():
    from kge_keras_amd import _lib, models as M
    C = _lib.C
    N, R, d = 50, 3, 16
    rs = np.random.RandomState(1)
    m = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, loss_function="pointwize_logistic")
    m.compile("adam")
    pos = np.stack([rs.randint(0, N, 8), rs.randint(0, R, 8), rs.randint(0, N, 8)], 1).astype(np.int32)
    neg = np.stack([rs.randint(0, N, 80), rs.randint(0, R, 80), rs.randint(0, N, 80)], 1).astype(np.int32)
    m._ensure_train_state(8, 80)
    _state(m)
    E_before, R_before = m.E.clone(), m.Rel.clone()
    Et, Rt = torch.zeros_like(m.E), torch.zeros_like(m.Rel)
    plan = m.plan_batch([pos, neg])
    md, ld, od = m._model_desc(True), m._loss_desc(), m._opt_desc()
    args = lambda eo, ro: (C.byref(md), C.byref(ld), C.byref(od), m.E.data_ptr(), m.Rel.data_ptr(), eo, ro, plan.buf.data_ptr(), plan.buf.numel(), 8, 80,
                           m._loss.data_ptr(), 0, m._ws.data_ptr(), m._ws.numel(), 3, _lib.stream_ptr())
    assert _lib.lib.kge_train_step_swap(*args(m.E.data_ptr(), Rt.data_ptr())) == -1      # aliased tables: invalid argument
    _lib.check(_lib.lib.kge_train_step_swap(*args(Et.data_ptr(), Rt.data_ptr())))
    torch.cuda.synchronize()
    assert torch.equal(m.E, E_before) and torch.equal(m.Rel, R_before)
    assert not torch.equal(Et, E_before) and float((Et - E_before).abs().max()) < 0.01
    # without the device step state or with an L3 term the twin step is not offered; with dropout it is
    od2 = _lib.OptDesc(od.kind, od.lr, od.beta1, od.beta2, od.eps, 1, od.mE, od.vE, od.mRel, od.vRel, None)
    assert _lib.lib.kge_train_step_swap_supported(C.byref(md), C.byref(ld), C.byref(od2)) == 0
    ld2 = m._loss_desc()
    ld2.regularization = 0.01
    assert _lib.lib.kge_train_step_swap_supported(C.byref(md), C.byref(ld2), C.byref(od)) == 0
    md2 = m._model_desc(True)
    md2.dropout = 0.2
    assert _lib.lib.kge_train_step_swap_supported(C.byref(md2), C.byref(ld), C.byref(od)) == 1


@pytest.mark.parametrize("order", [(1, 2), (2, 1)])
def test_swap_halves_equal_the_whole_step(order):
    """parts = 1 (entity half) and parts = 2 (relation half) of a step, issued one after the other in either order, leave what
    parts = 3 (both in one launch + the closing launch) leaves -- tables, moments, loss, step state -- to fp32 rounding (the same
    per-wave functions compiled into different kernels: contraction differs by an ulp here and there)."""
    from kge_keras_amd import _lib, models as M
    C = _lib.C
    N, R, d = 120, 4, 48
    rs = np.random.RandomState(5)
    E0 = rs.uniform(-0.5, 0.5, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, 2 * d)).astype(np.float32)
    batches = [(np.stack([rs.randint(0, N, 16), rs.randint(0, R, 16), rs.randint(0, N, 16)], 1).astype(np.int32),
                np.stack([rs.randint(0, N, 900), rs.randint(0, R, 900), rs.randint(0, N, 900)], 1).astype(np.int32)) for _ in range(3)]
    out = []
    for parts_seq in ((3,), order):
        m = M.ComplEx(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, loss_function="pointwize_logistic")
        m.compile("adam")
        m._ensure_train_state(16, 900)
        sd = _state(m)
        cur, nxt = (m.E, m.Rel), (torch.empty_like(m.E), torch.empty_like(m.Rel))
        losses = []
        for k, (pos, neg) in enumerate(batches):
            plan = m.plan_batch([pos, neg])
            md, ld = m._model_desc(True), m._loss_desc()
            m.optimizer.iterations = k + 1
            od = m._opt_desc()
            for parts in parts_seq:
                _lib.check(_lib.lib.kge_train_step_swap(C.byref(md), C.byref(ld), C.byref(od), cur[0].data_ptr(), cur[1].data_ptr(), nxt[0].data_ptr(),
                                                        nxt[1].data_ptr(), plan.buf.data_ptr(), plan.buf.numel(), 16, 900, m._loss.data_ptr(), 0,
                                                        m._ws.data_ptr(), m._ws.numel(), parts, _lib.stream_ptr()))
            losses.append(float(m._loss[0].item()))
            cur, nxt = nxt, cur
        out.append((cur[0].clone(), cur[1].clone(), [x.clone() for x in m.optimizer.slots], losses, sd.clone()))
    a, b = out
    diffs = [float((x - y).abs().max()) for x, y in zip([a[0], a[1]] + a[2], [b[0], b[1]] + b[2])]
    assert max(diffs) < 2e-7, diffs
    assert a[3] == pytest.approx(b[3], rel=1e-6) and int(a[4][0].item()) == int(b[4][0].item()) == 3
    assert a[4].view(torch.float32)[2].item() == b[4].view(torch.float32)[2].item()  # lr_t of the next update


@pytest.mark.parametrize("model,kw,d", [("DistMult", {}, 200), ("TransE", dict(norm=2, gamma=12), 64), ("ComplEx", {}, 40)])
def test_gradient_halves_equal_the_whole(model, kw, d):
    """kge_train_grads_planned_parts: entity half (parts 1) + relation half (parts 2), in either order, fill the gradient tables and
    the loss word like parts 3 and like the oracle's gradients; a hub entity and a dominant relation included."""
    from kge_keras_amd import models as M
    if os.environ.get("KGE_GRAD_SELF") == "0":
        pytest.skip("KGE_GRAD_SELF=0 keeps the gradient half in one piece")
    N, R = 200, 5
    rs = np.random.RandomState(11)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    ent = lambda n: np.where(rs.rand(n) < 0.3, 3, rs.randint(0, N - 20, n))
    rel = lambda n: np.where(rs.rand(n) < 0.6, 1, rs.randint(0, R, n))
    pos = np.stack([ent(48), rel(48), ent(48)], 1).astype(np.int32)
    neg = np.stack([ent(1200), rel(1200), ent(1200)], 1).astype(np.int32)
    out = []
    for seq in ((3,), (1, 2), (2, 1)):
        m = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                              loss_function="pointwize_logistic", **kw)
        m.compile("sgd")
        assert m.grad_halves_supported()
        plan = m.plan_batch([pos, neg])
        for parts in seq:
            m.train_grads(plan, parts=parts)
        out.append((m._gE.clone(), m._gR.clone(), float(m._loss[0].item())))
        m._raise_on_flag()
    Lo, _sp, _sn, gEo, gRo = O.dense_grads(model, E0.astype(np.float64), R0.astype(np.float64), pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
    for gE, gR, L in out:
        np.testing.assert_allclose(gE.cpu().numpy(), gEo, atol=1e-5)
        np.testing.assert_allclose(gR.cpu().numpy(), gRo, atol=1e-5)
        assert L == pytest.approx(Lo, rel=1e-5)
    for gE, gR, L in out[1:]:
        assert float((gE - out[0][0]).abs().max()) < 1e-7 and float((gR - out[0][1]).abs().max()) < 1e-7 and L == pytest.approx(out[0][2], rel=1e-6)
    assert M.DistMult(e_dim=16, r_dim=16, num_entities=N, num_relations=R, dp=0.3, loss_function="pointwize_logistic").grad_halves_supported()
    assert not M.DistMult(e_dim=320, r_dim=320, num_entities=N, num_relations=R, dp=0.0, loss_function="pointwize_logistic").grad_halves_supported()


@pytest.mark.parametrize("case", range(24))
def test_swap_step_random_shapes(case):
    """Seeded random shapes for the twin-table step: row widths 4..256 (every sub-group shape, widths that do not fill the last
    chunk), tiny and skewed tables (a single relation, three entities, every row on one entity), batches of one row -- against the
    three-launch step (fp32 rounding) and the oracle (1e-4), two steps each."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(1000 + case)
    model, kw = [("DistMult", {}), ("TransE", dict(norm=1, gamma=12)), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {})][case % 4]
    d = int(rs.choice([4, 8, 12, 20, 36, 64, 68, 100, 128, 132, 200, 256]))
    if model == "ComplEx":
        d = min(d, 128)
    N = int(rs.choice([3, 7, 50, 400]))
    R = int(rs.choice([1, 2, 9]))
    n_pos, n_neg = int(rs.choice([1, 5, 64])), int(rs.choice([1, 37, 900, 2500]))
    loss = ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss", "pointwize_cross_entropy"][int(rs.randint(4))]
    opt = ["adam", "sgd"][int(rs.randint(2))]
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    a, b = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                              loss_function=loss, margin=1.0, **kw) for _ in range(2))
    for m in (a, b):
        m.compile(opt)
        m._ensure_train_state(n_pos, n_neg)
    _state(a), _state(b)
    a._step_twins = (torch.empty_like(a.E), torch.empty_like(a.Rel))
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer(opt)
    skew = rs.rand() < 0.4
    for step in range(2):
        ent = (lambda n: np.where(rs.rand(n) < 0.8, 0, rs.randint(0, N, n))) if skew else (lambda n: rs.randint(0, N, n))
        pos = np.stack([ent(n_pos), rs.randint(0, R, n_pos), ent(n_pos)], 1).astype(np.int32)
        neg = np.stack([ent(n_neg), rs.randint(0, R, n_neg), ent(n_neg)], 1).astype(np.int32)
        La = float(a.train_step(a.plan_batch([pos, neg]))["loss"].item())
        Lb = float(b.train_step(b.plan_batch([pos, neg]))["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, loss, 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
        info = (case, model, d, N, R, n_pos, n_neg, loss, opt, skew, step)
        assert La == pytest.approx(Lb, rel=2e-6, abs=1e-6) and La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), info
        np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=5e-6, err_msg=str(info))
        np.testing.assert_allclose(a.Rel.cpu().numpy(), b.Rel.cpu().numpy(), atol=5e-6, err_msg=str(info))
        np.testing.assert_allclose(a.E.cpu().numpy(), Eo, atol=TOL_W, err_msg=str(info))
        np.testing.assert_allclose(a.Rel.cpu().numpy(), Ro, atol=TOL_W, err_msg=str(info))
    a._raise_on_flag()


@pytest.mark.parametrize("model,kw,d", [("DistMult", {}, 200), ("TransE", dict(norm=1, gamma=12), 24), ("TransE", dict(norm=2, gamma=12), 100),
                                        ("ComplEx", {}, 40), ("ComplEx", {}, 200), ("pRotatE", {}, 100)])
def test_swap_step_with_dropout(model, kw, d):
    """dp > 0 (models.py:122,144-148): the twin-table step regenerates the masks of the three gathered rows from (seed, batch row,
    role) in both roles of its launch -- equal to the three-launch step (same masks, fp32 rounding) and to the oracle with the same
    masks (1e-4); gradient halves likewise."""
    from kge_keras_amd import models as M
    if os.environ.get("KGE_GRAD_SELF") == "0":
        pytest.skip("KGE_GRAD_SELF=0 keeps the gradient half in one piece")
    N, R, dp = 150, 5, 0.3
    rs = np.random.RandomState(21)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    a, b, c = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=dp, init_entities=E0, init_relations=R0,
                                 loss_function="pointwize_logistic", margin=1.0, seed=5, **kw) for _ in range(3))
    for m in (a, b, c):
        m.compile("adam")
        m._ensure_train_state(40, 700)
    _state(a), _state(b)
    a._step_twins = (torch.empty_like(a.E), torch.empty_like(a.Rel))
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.make_optimizer("adam")
    for step in range(2):
        ent = lambda n: np.where(rs.rand(n) < 0.3, 3, rs.randint(0, N, n))
        pos = np.stack([ent(40), rs.randint(0, R, 40), ent(40)], 1).astype(np.int32)
        neg = np.stack([ent(700), rs.randint(0, R, 700), ent(700)], 1).astype(np.int32)
        La = float(a.train_step(a.plan_batch([pos, neg]))["loss"].item())
        Lb = float(b.train_step(b.plan_batch([pos, neg]))["loss"].item())
        assert a._seed == b._seed
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, a.modulus if model == "pRotatE" else kw.get("norm", 1),
                          kw.get("gamma", 12), dropout=dp, seed=a._seed)
        assert La == pytest.approx(Lb, rel=2e-6, abs=1e-6) and La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=5e-6)
        np.testing.assert_allclose(a.Rel.cpu().numpy(), b.Rel.cpu().numpy(), atol=5e-6)
        assert_tables(a.E.cpu().numpy(), Eo)
        assert_tables(a.Rel.cpu().numpy(), Ro)
    # gradient halves under dropout == the whole == the oracle's gradients with the same masks
    plan = c.plan_batch([pos, neg])
    c.train_grads(plan, parts=1)
    c.train_grads(plan, parts=2)
    Lg, _sp, _sn, gEo, gRo = O.dense_grads(model, E0.astype(np.float64), R0.astype(np.float64), pos, neg, "pointwize_logistic", 1.0, 1,
                                           c.modulus if model == "pRotatE" else kw.get("norm", 1), kw.get("gamma", 12), dropout=dp, seed=c._seed)
    np.testing.assert_allclose(c._gE.cpu().numpy(), gEo, atol=1e-5)
    np.testing.assert_allclose(c._gR.cpu().numpy(), gRo, atol=1e-5)
    assert float(c._loss[0].item()) == pytest.approx(Lg, rel=1e-5)


@pytest.mark.parametrize("opt", ["adam", "sgd"])
@pytest.mark.parametrize("model,kw,d,n_win", [("DistMult", {}, 200, 2), ("DistMult", {}, 24, 3), ("TransE", dict(norm=2, gamma=12), 64, 2),
                                              ("ComplEx", {}, 40, 4), ("pRotatE", {}, 24, 2)])
def test_row_windows_on_one_device_equal_the_unsharded_step(model, kw, d, n_win, opt):
    """kge_train_step_swap_window: the owner-computes step of SURVEY 8(e) rehearsed on ONE device.  `n_win` "ranks" -- each with its own plan
    of the SAME batch, its own step state and ONLY ITS WINDOW'S Adam moment rows (base pointers offset by -lo rows) -- write their
    entity rows and relation rows into shared out tables; together these must equal what kge_train_step_swap leaves on one GPU
    (<= 1e-6: the list order inside a row comes from each plan build's atomics), the partial losses must add up to the batch loss,
    every plan's counters must be left zero (a later build with parts | 8 still trains right), and rows outside a window must not be
    written by that window's call.  Hub entity (chunk waves + ticket inside one window), untouched entities, a dominant relation,
    windows that do not divide the tables; three steps, tables alternating."""
    from kge_keras_amd import _lib, models as M
    from kge_keras_amd import dist as D
    C, lib = _lib.C, _lib.lib
    N, R = 301, 7
    rs = np.random.RandomState(13)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    mk = lambda: getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,  # noqa: E731
                                   loss_function="pointwize_logistic", margin=1.0, **kw)
    n_pos, n_neg = 64, 1500
    ref = mk()
    ref.compile(opt)
    ref._ensure_train_state(n_pos, n_neg)
    _state(ref)
    ref._step_twins = (torch.empty_like(ref.E), torch.empty_like(ref.Rel))
    # the "ranks"
    ranks = []
    for r in range(n_win):
        m = mk()
        m.compile(opt)
        (lo, hi), (rlo, rhi) = D.shard_range(N, r, n_win), D.shard_range(R, r, n_win)
        sd = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device="cuda")
        o = m.optimizer
        od0 = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, 1, None, None, None, None, sd.data_ptr())
        _lib.check(lib.kge_opt_state_init(C.byref(od0), _lib.stream_ptr()))
        slots = [torch.zeros((max(1, hi - lo), de), device="cuda"), torch.zeros((max(1, hi - lo), de), device="cuda"),
                 torch.zeros((max(1, rhi - rlo), dr), device="cuda"), torch.zeros((max(1, rhi - rlo), dr), device="cuda")]
        ptrs = [slots[0].data_ptr() - 4 * lo * de, slots[1].data_ptr() - 4 * lo * de, slots[2].data_ptr() - 4 * rlo * dr, slots[3].data_ptr() - 4 * rlo * dr]
        md, ld = m._model_desc(True), m._loss_desc()
        ws = torch.zeros(int(_lib.check(lib.kge_train_step_workspace_bytes(C.byref(md), C.byref(ld), n_pos, n_neg))), dtype=torch.uint8, device="cuda")
        ranks.append(dict(m=m, win=_lib.RowWindow(lo, hi, rlo, rhi), sd=sd, slots=slots, ptrs=ptrs if opt == "adam" else [None] * 4, ws=ws,
                          loss=torch.zeros(1, device="cuda"), plan=None))
    cur = (torch.from_numpy(E0).cuda(), torch.from_numpy(R0).cuda())
    nxt = (torch.full_like(cur[0], 7.0), torch.full_like(cur[1], 7.0))  # (7: never a table value -- rows nobody wrote would show)
    for step in range(3):
        ent = lambda n: np.where(rs.rand(n) < 0.3, 7, rs.randint(0, N - 40, n))  # hub: entity 7; entities N-40.. never touched
        rel = lambda n: np.where(rs.rand(n) < 0.6, 2, rs.randint(0, R, n))
        pos = np.stack([ent(n_pos), rel(n_pos), ent(n_pos)], 1).astype(np.int32)
        neg = np.stack([ent(n_neg), rel(n_neg), ent(n_neg)], 1).astype(np.int32)
        L_ref = float(ref.train_step(ref.plan_batch([pos, neg]))["loss"].item())
        tot = 0.0
        for k, rk in enumerate(ranks):
            m = rk["m"]
            rk["plan"] = m.plan_batch([pos, neg], out=rk["plan"], parts=7 | (8 if step else 0))  # (8: counters known to be zero after a window step)
            md, ld = m._model_desc(True), m._loss_desc()
            o = m.optimizer
            od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, step + 1, *rk["ptrs"], rk["sd"].data_ptr())
            before = nxt[0].clone()
            _lib.check(lib.kge_train_step_swap_window(C.byref(md), C.byref(ld), C.byref(od), cur[0].data_ptr(), cur[1].data_ptr(), nxt[0].data_ptr(),
                                                      nxt[1].data_ptr(), rk["plan"].buf.data_ptr(), rk["plan"].buf.numel(), n_pos, n_neg,
                                                      rk["loss"].data_ptr(), 0, rk["ws"].data_ptr(), rk["ws"].numel(), C.byref(rk["win"]), _lib.stream_ptr()))
            tot += float(rk["loss"][0].item())
            w = rk["win"]
            outside = torch.ones(N, dtype=torch.bool, device="cuda")
            outside[w.ent_lo:w.ent_hi] = False
            assert torch.equal(nxt[0][outside], before[outside]), (step, k)  # rows of other windows untouched by this call
            assert int(rk["sd"][0].item()) == step + 1
        assert tot == pytest.approx(L_ref, rel=2e-6), step
        np.testing.assert_allclose(nxt[0].cpu().numpy(), ref.E.cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(nxt[1].cpu().numpy(), ref.Rel.cpu().numpy(), atol=1e-6)
        if opt == "adam":  # the moment shards, stitched together, are the unsharded moments
            for j in range(4):
                whole = torch.cat([rk["slots"][j][:((rk["win"].ent_hi - rk["win"].ent_lo) if j < 2 else (rk["win"].rel_hi - rk["win"].rel_lo))] for rk in ranks])
                np.testing.assert_allclose(whole.cpu().numpy(), ref.optimizer.slots[j].cpu().numpy(), atol=1e-7)
        cur, nxt = nxt, cur
        nxt[0].fill_(7.0)
        nxt[1].fill_(7.0)
    ref._raise_on_flag()
    # bad windows are refused
    bad = _lib.RowWindow(5, N + 1, 0, R)
    rk = ranks[0]
    md, ld = rk["m"]._model_desc(True), rk["m"]._loss_desc()
    o = rk["m"].optimizer
    od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, 4, *rk["ptrs"], rk["sd"].data_ptr())
    assert lib.kge_train_step_swap_window(C.byref(md), C.byref(ld), C.byref(od), cur[0].data_ptr(), cur[1].data_ptr(), nxt[0].data_ptr(), nxt[1].data_ptr(),
                                          rk["plan"].buf.data_ptr(), rk["plan"].buf.numel(), n_pos, n_neg, rk["loss"].data_ptr(), 0, rk["ws"].data_ptr(),
                                          rk["ws"].numel(), C.byref(bad), _lib.stream_ptr()) == -1


def test_row_shard_trainer_world_of_one_equals_twin_step():
    """dist.RowShardTrainer without a process group (one rank: the window is the whole table, the exchange a swap of the twin
    buffers): three steps equal the ordinary twin-table step, model.E / model.Rel follow the current buffers."""
    from kge_keras_amd import models as M
    from kge_keras_amd import dist as D
    N, R, d = 90, 5, 32
    rs = np.random.RandomState(2)
    E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
    a, b = (M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, loss_function="pointwize_hinge", margin=1.0)
            for _ in range(2))
    for m in (a, b):
        m.compile("adam")
    a._ensure_train_state(40, 400)
    _state(a)
    a._step_twins = (torch.empty_like(a.E), torch.empty_like(a.Rel))
    tr = D.RowShardTrainer(b)
    Eo, Ro, oo = E0.astype(np.float64), R0.astype(np.float64), O.make_optimizer("adam")
    for step in range(3):
        pos = np.stack([rs.randint(0, N, 40), rs.randint(0, R, 40), rs.randint(0, N, 40)], 1).astype(np.int32)
        neg = np.stack([rs.randint(0, N, 400), rs.randint(0, R, 400), rs.randint(0, N, 400)], 1).astype(np.int32)
        La = float(a.train_step(a.plan_batch([pos, neg]))["loss"].item())
        Lb = float(tr.step(b.plan_batch([pos, neg])).item())
        Lo = O.train_step("DistMult", Eo, Ro, oo, pos, neg, "pointwize_hinge", 1.0)
        assert La == pytest.approx(Lb, rel=1e-6) and Lb == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA)
        np.testing.assert_allclose(b.E.cpu().numpy(), a.E.cpu().numpy(), atol=1e-6)
        assert_tables(b.E.cpu().numpy(), Eo)
        assert_tables(b.Rel.cpu().numpy(), Ro)
    assert b.predict(pos).shape == (40,)


@pytest.mark.parametrize("opt", ["adam", "sgd"])
@pytest.mark.parametrize("model,kw,d", [("DistMult", {}, 200), ("DistMult", {}, 24), ("TransE", dict(norm=2, gamma=12), 64), ("ComplEx", {}, 40),
                                        ("pRotatE", {}, 24)])
def test_chained_steps_equal_the_same_steps_one_by_one(model, kw, d, opt):
    """kge_train_step_swap_chain: the relation rows, the loss and the step state of a step are produced by late workgroups of the step's
    ONE launch (they wait on per-workgroup flags of the relation-grouped forward: write-through stores, cache-bypassing loads).  Same
    sums in the same order: tables, moments, step state (count, next lr_t, loss sum) and the loss must be BIT-IDENTICAL to the same
    steps through kge_train_step_swap AFTER EVERY STEP (list order pinned by the deterministic mode, since the two models build their
    own plans).  Chains of 1, 2 and 5 steps back to back on the same workspace (seq = 0 refills the lr ring and clears the flags); a
    hub entity, a dominant relation, relations without rows, untouched entities; the plan buffers' counters must be left zero by a
    chained step (rebuilt with parts | 8)."""
    from kge_keras_amd import _lib, models as M
    C, lib = _lib.C, _lib.lib
    N, R = 301, 9
    rs = np.random.RandomState(21)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    mk = lambda: getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,  # noqa: E731
                                   loss_function="pointwize_logistic", margin=1.0, **kw)
    n_pos, n_neg = 64, 1500
    lib.kge_set_deterministic(1)
    try:
        ref, chn = mk(), mk()
        for m in (ref, chn):
            m.compile(opt)
            m._ensure_train_state(n_pos, n_neg)
            _state(m)
        ref._step_twins = (torch.empty_like(ref.E), torch.empty_like(ref.Rel))
        cur = [chn.E, chn.Rel]
        nxt = [torch.full_like(chn.E, 7.0), torch.full_like(chn.Rel, 7.0)]
        plans = [None, None, None]
        md, ld = chn._model_desc(True), chn._loss_desc()
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        k = 0
        for chain_len in (1, 2, 5):
            for seq in range(chain_len):
                ent = lambda n: np.where(rs.rand(n) < 0.3, 7, rs.randint(0, N - 40, n))  # hub: entity 7; entities N-40.. never touched
                rel = lambda n: np.where(rs.rand(n) < 0.6, 2, rs.randint(0, R - 2, n))   # relation 2 dominant; relations R-2.. without rows
                pos = np.stack([ent(n_pos), rel(n_pos), ent(n_pos)], 1).astype(np.int32)
                neg = np.stack([ent(n_neg), rel(n_neg), ent(n_neg)], 1).astype(np.int32)
                L_ref = ref.train_step(ref.plan_batch([pos, neg]))["loss"].clone()
                plans[k % 3] = chn.plan_batch([pos, neg], out=plans[k % 3], parts=7 | (8 if k >= 3 else 0))
                chn.optimizer.iterations += 1
                od = chn._opt_desc()
                _lib.check(lib.kge_train_step_swap_chain(C.byref(md), C.byref(ld), C.byref(od), cur[0].data_ptr(), cur[1].data_ptr(), nxt[0].data_ptr(),
                                                         nxt[1].data_ptr(), plans[k % 3].buf.data_ptr(), plans[k % 3].buf.numel(), n_pos, n_neg,
                                                         chn._loss.data_ptr(), 0, chn._ws.data_ptr(), chn._ws.numel(), seq, err.data_ptr(), _lib.stream_ptr()))
                cur, nxt = nxt, cur
                k += 1
                torch.cuda.synchronize()
                assert int(err.item()) == 0
                assert torch.equal(cur[0], ref.E), (chain_len, seq, float((cur[0] - ref.E).abs().max()))
                assert torch.equal(cur[1], ref.Rel), (chain_len, seq, float((cur[1] - ref.Rel).abs().max()))
                assert torch.equal(chn._step_dev, ref._step_dev), (chain_len, seq, chn._step_dev, ref._step_dev)
                assert float(chn._loss[0]) == float(L_ref)
                if opt == "adam":
                    for a, b in zip(chn.optimizer.slots, ref.optimizer.slots):
                        assert torch.equal(a, b)
                nxt[0].fill_(7.0)
                nxt[1].fill_(7.0)
    finally:
        lib.kge_set_deterministic(0)
