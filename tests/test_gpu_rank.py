"""GPU parity of the filtered 1-vs-all ranking (utils.py:199-264) through kge_rank_counts."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _mk(model, N, R, d, seed=0, quant=None, **kw):
    from kge_keras_amd import models as M
    rs = np.random.RandomState(seed)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-1, 1, (N, de)).astype(np.float32)
    R0 = rs.uniform(-1, 1, (R, dr)).astype(np.float32)
    if quant:
        E0 = (np.round(E0 * quant) / quant).astype(np.float32)
        R0 = (np.round(R0 * quant) / quant).astype(np.float32)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, **kw)
    return mdl, E0, R0, rs


def _ids(a):
    return [tuple(int(x) for x in t) for t in a]


def _check_ranks(model, mdl, E0, R0, test, filt, N, norm=1, gamma=12):
    fh, ft = O.build_filters(filt)
    tails, heads = mdl.rank(test, N, ft, fh)
    gaps = []
    to, ho = O.ranks_closed_form(model, E0, R0, test, filt, norm, gamma, margin_out=gaps)
    gaps = np.asarray(gaps).reshape(-1, 2)
    # fp32 rounding of a K-term sum is <~ 2^-23 * K * max|term|
    term = float(np.abs(E0).max() ** 2 * np.abs(R0).max()) if model != "TransE" else float(2 * np.abs(E0).max() + np.abs(R0).max()) ** norm
    if model == "HolE":
        term *= E0.shape[1]  # d^2 terms per score
    tol = 5e-7 * E0.shape[1] * term + (1e-5 if model == "TransE" else 0.0)
    for got, ref, gap in ((tails, to, gaps[:, 0]), (heads, ho, gaps[:, 1])):
        clear = gap > tol  # no candidate within float32 noise of the target: rank must be exact
        assert clear.mean() > 0.5
        assert np.array_equal(got[clear], ref[clear])
        assert np.all(np.abs(got[~clear] - ref[~clear]) <= 2)
    return tails, heads, to, ho


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("ComplEx", {}), ("HolE", {}), ("TransE", dict(norm=1, gamma=12)),
                                      ("TransE", dict(norm=2, gamma=12)), ("TransE", dict(norm=2, gamma=0))])
@pytest.mark.parametrize("N,d,nq", [(14, 32, 40), (300, 200, 257), (1000, 30, 64), (129, 8, 1)])
def test_rank_parity_random(model, kw, N, d, nq):
    mdl, E0, R0, rs = _mk(model, N, 7, d, **kw)
    test = list({(int(rs.randint(N)), int(rs.randint(7)), int(rs.randint(N))) for _ in range(nq)})
    # filters: random other triples sharing (s,p) or (p,o) with the test triples, never the test triples themselves
    tes = set(test)
    filt = set()
    for (s, p, o) in test:
        for _ in range(int(rs.randint(0, 6))):
            c = (s, p, int(rs.randint(N)))
            if c not in tes:
                filt.add(c)
            c = (int(rs.randint(N)), p, o)
            if c not in tes:
                filt.add(c)
    _check_ranks(model, mdl, E0, R0, test, sorted(filt), N, kw.get("norm", 1), kw.get("gamma", 12))


def test_rank_mfma_bit_exact_vs_chain():
    """The sweep's count must equal the count taken on the chain scores (the numbers the filter kernel and the target
    use): this is what makes un-counting filtered candidates exact.  Coarse-grid tables give many exact ties."""
    from kge_keras_amd import _lib
    for model, quant, kw in (("DistMult", None, {}), ("DistMult", 2, {}), ("ComplEx", 2, {}), ("HolE", None, {}), ("HolE", 2, {}),
                             ("TransE", None, dict(norm=1, gamma=12)),
                             ("TransE", 2, dict(norm=1, gamma=12)), ("TransE", 2, dict(norm=2, gamma=12)), ("TransE", None, dict(norm=2, gamma=0))):
        N, d, nq = 517, 40, 200
        mdl, E0, R0, rs = _mk(model, N, 5, d, seed=3, quant=quant, **kw)
        test = np.stack([rs.randint(0, N, nq), rs.randint(0, 5, nq), rs.randint(0, N, nq)], 1).astype(np.int32)
        for side in (0, 1):
            cnt = mdl.rank_counts(test, side, np.zeros(0, np.int32), np.zeros(0, np.int32)).cpu().numpy()
            t = torch.from_numpy(test).cuda()
            md = mdl._model_desc(False)
            out = torch.empty((nq, N), dtype=torch.float32, device="cuda")
            _lib.check(_lib.lib.kge_rank_chain_scores(C.byref(md), mdl.E.data_ptr(), mdl.Rel.data_ptr(), t.data_ptr(), nq, side,
                                                      out.data_ptr(), mdl._rank_ws.data_ptr(), mdl._rank_ws.numel(),
                                                      mdl._err.data_ptr(), _lib.stream_ptr()))
            sc = out.cpu().numpy()
            tgt = sc[np.arange(nq), test[:, 2] if side == 0 else test[:, 0]]
            ref = (sc > tgt[:, None]).sum(1)
            if model == "TransE" and kw.get("norm") == 2:
                # TransE-L2 ranks on the matrix cores by D = |e|^2 - 2<q,e> (csrc/rank.hip, rank_sumsq_kernel), the chain-score hook
                # reports the distance form: the two orders agree except for candidates within rounding of the target
                near = (np.abs(sc - tgt[:, None]) <= 2e-5 * np.maximum(1.0, np.abs(tgt[:, None]))).sum(1) - 1
                assert np.all(np.abs(cnt - ref) <= near), (model, quant, side, np.abs(cnt - ref).max())
            else:
                assert np.array_equal(cnt, ref), (model, quant, side, np.abs(cnt - ref).max())
            # and the chain scores are the model's scores to fp32 accuracy
            cols = np.arange(N)
            q0 = test[0]
            tri = np.stack([np.full(N, q0[0]) if side == 0 else cols, np.full(N, q0[1]), cols if side == 0 else np.full(N, q0[2])], 1)
            np.testing.assert_allclose(sc[0], O.call(model, E0, R0, tri, kw.get("norm", 1), kw.get("gamma", 12)), atol=1e-4, rtol=1e-5)


def test_rank_ties_are_optimistic():
    """Quantised tables: exact score ties between target and candidates must NOT count (strict >), as rankdata 'max' gives."""
    mdl, E0, R0, rs = _mk("DistMult", 64, 3, 8, seed=6, quant=1)  # entries in {-1,0,1}: integer scores, exact in fp32
    test = list({(int(rs.randint(64)), int(rs.randint(3)), int(rs.randint(64))) for _ in range(50)})
    tails, heads = mdl.rank(test, 64, {}, {})
    to, ho = O.ranks_closed_form("DistMult", E0, R0, test, [])
    assert np.array_equal(tails, to) and np.array_equal(heads, ho)
    assert (to < 64).all() and len(set(to.tolist())) > 3


def test_validate_golden_metrics_on_gpu(golden, datasets):
    """utils.validate with the GPU ranker reproduces the reference's metric dict for its own DistMult numpy model."""
    from kge_keras_amd import utils as U
    from kge_keras_amd.models import DistMult
    for c in golden("validate")["cases"]:
        m = c["model"]
        rs = np.random.RandomState(m["seed"])
        E = rs.uniform(-1, 1, (m["N"], m["d"]))
        Rl = rs.uniform(-1, 1, (m["R"], m["d"]))
        if m["quant"]:
            E, Rl = np.round(E * m["quant"]) / m["quant"], np.round(Rl * m["quant"]) / m["quant"]
        mdl = DistMult(e_dim=m["d"], r_dim=m["d"], num_entities=m["N"], num_relations=m["R"], dp=0, init_entities=E, init_relations=Rl)
        ds = c["dataset"]
        tes = set(_ids(datasets[ds + "_test"]))
        filt = [t for t in _ids(datasets[ds + "_train"]) + _ids(datasets[ds + "_valid"]) if t not in tes]
        met = U.validate(mdl, c["test"], m["N"], 32, filtering_triples=filt)
        assert list(met.keys()) == list(c["metrics"].keys())
        if m["quant"]:  # exact arithmetic on the coarse grid: identical ranks, identical metrics
            for k, v in c["metrics"].items():
                assert met[k] == pytest.approx(v, abs=1e-12), k
        else:  # fp32 tables vs the reference's fp64: ranks can move only on float-noise ties
            for k in ("mr", "tail_mr", "head_mr"):
                assert met[k] == pytest.approx(c["metrics"][k], rel=2e-3), k
            for k in ("mrr", "hits@10"):
                assert met[k] == pytest.approx(c["metrics"][k], abs=2e-2), k


def test_rank_error_behaviour():
    from kge_keras_amd import utils as U
    mdl, E0, R0, rs = _mk("DistMult", 14, 5, 8)
    with pytest.raises(ValueError):  # test triple inside its own filter (utils.py:168-171)
        U.validate(mdl, [(0, 1, 2)], 14, 32, filtering_triples=[(0, 1, 2)])
    with pytest.raises(TypeError):
        U.validate(mdl, [(0, 1, 2)], 14, 32)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=2, gamma=12)), ("TransE", dict(norm=1, gamma=12))])
def test_rank_sharded_candidates_sum_to_unsharded(model, kw):
    """Row-sharded sweep (SURVEY 8(e) schedule ii on one GPU): counts over candidate shards add up exactly."""
    mdl, E0, R0, rs = _mk(model, 1000, 4, 64, seed=9, **kw)
    nq = 300
    test = np.stack([rs.randint(0, 1000, nq), rs.randint(0, 4, nq), rs.randint(0, 1000, nq)], 1).astype(np.int32)
    fq = np.repeat(np.arange(nq, dtype=np.int32), 3)
    fi = rs.randint(0, 1000, 3 * nq).astype(np.int32)
    keep = fi != np.repeat(test[:, 2], 3)
    fq, fi = fq[keep], fi[keep]
    full = mdl.rank_counts(test, 0, fq, fi).cpu().numpy()
    acc = None
    for a, b in ((0, 125), (125, 500), (500, 1000)):
        acc = mdl.rank_counts(test, 0, fq, fi, cand_offset=a, cand_rows=b - a, counts=acc)
    assert np.array_equal(full, acc.cpu().numpy())


def test_rank_wn18rr_complex_slice(datasets):
    """BASELINE config 4 shape: WN18RR (N=40943, R=11), ComplEx d=400 -> K=800, real test triples, real filters."""
    from kge_keras_amd.models import ComplEx
    N, R, d = 40943, 11, 400
    rs = np.random.RandomState(1234)
    # Glorot tables scaled up (scores of order 1 instead of 1e-4), so that fp32 rounding is far below the gaps between candidates
    E0, R0 = (O.glorot_uniform(N, 2 * d, rs) * 30).astype(np.float32), (O.glorot_uniform(R, 2 * d, rs) * 10).astype(np.float32)
    mdl = ComplEx(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0, init_relations=R0)
    test = _ids(datasets["wn18rr_test"][:48])
    tes = set(_ids(datasets["wn18rr_test"]))
    filt = [t for t in _ids(datasets["wn18rr_train"]) + _ids(datasets["wn18rr_valid"]) if t not in tes]
    fh, ft = O.build_filters(filt)
    tails, heads = mdl.rank(test, N, ft, fh)
    gaps = []
    to, ho = O.ranks_closed_form("ComplEx", E0, R0, test, filt, margin_out=gaps)
    gaps = np.asarray(gaps).reshape(-1, 2)
    # 40 943 candidates on a score range of a few units lie ~1e-4 apart, fp32 rounding of an 800-term chain is ~1e-6 here: exact ranks
    # wherever no candidate lies within 1e-5 of the target's score, never more than one place off, and exact for nearly every query
    st, sh = gaps[:, 0] > 1e-5, gaps[:, 1] > 1e-5
    assert np.array_equal(tails[st], to[st]) and np.array_equal(heads[sh], ho[sh]), (tails[st] - to[st], heads[sh] - ho[sh])
    assert np.all(np.abs(tails - to) <= 1) and np.all(np.abs(heads - ho) <= 1), (np.abs(tails - to).max(), np.abs(heads - ho).max())
    assert (tails == to).mean() > 0.9 and (heads == ho).mean() > 0.9, ((tails == to).mean(), (heads == ho).mean(), st.mean(), sh.mean())


def test_nations_end_to_end_fit_then_validate(datasets):
    """The reference's workflow end to end on its smallest dataset: build_model -> fit(DataGenerator) -> validate, plus the
    working equivalent of KGEValidateCallback.  Link prediction must end up far better than chance (MR of a random scorer
    on 14 entities is ~7)."""
    import random
    from kge_keras_amd.harness import DataGenerator, build_model
    from kge_keras_amd.utils import validate, KGEValidateCallback
    tr, va, te = (_ids(datasets["nations_" + k]) for k in ("train", "valid", "test"))
    np.random.seed(0)
    random.seed(0)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic",
                           margin=1, dp=0.1, seed=1))
    tes = set(te) | set(va)
    filt = [t for t in tr if t not in tes]
    before = validate(mdl, te, 14, None, filtering_triples=filt)
    cb = KGEValidateCallback(va, train_data=tr, period=20)
    h = mdl.fit(DataGenerator(tr, ns=10, batch_size=32), epochs=40, verbose=0, callbacks=[cb])
    after = validate(mdl, te, 14, None, filtering_triples=filt)
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert after["mrr"] > before["mrr"] + 0.1 and after["mr"] < 4.0, (before, after)
    assert len(after) == 15


def test_rank_yago310_shape_row_sharded(datasets):
    """BASELINE config 5 shape: YAGO3-10 (N = 123 182 entities, 37 relations), DistMult d=200, the entity table cut into 8 row
    shards as on an 8-GPU node; per-shard counts add up to the unsharded sweep and match the oracle on real test triples."""
    from kge_keras_amd.models import DistMult
    from kge_keras_amd.dist import shard_range
    N, R, d = 123182, 37, 200
    rs = np.random.RandomState(1234)
    E0, R0 = O.glorot_uniform(N, d, rs), O.glorot_uniform(R, d, rs)
    mdl = DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0, init_entities=E0 * 30, init_relations=R0 * 10)
    Ef, Rf = (E0 * 30).astype(np.float32), (R0 * 10).astype(np.float32)
    test = np.asarray(datasets["yago310_test"][:64], dtype=np.int32)
    test[:, 0] %= N
    test[:, 2] %= N
    test[:, 1] %= R
    fq = np.repeat(np.arange(len(test), dtype=np.int32), 2)
    fi = rs.randint(0, N, 2 * len(test)).astype(np.int32)
    keep = fi != np.repeat(test[:, 2], 2)
    fq, fi = fq[keep], fi[keep]
    full = mdl.rank_counts(test, 0, fq, fi).cpu().numpy()
    acc = None
    for r in range(8):
        lo, hi = shard_range(N, r, 8)
        acc = mdl.rank_counts(test, 0, fq, fi, cand_offset=lo, cand_rows=hi - lo, counts=acc)
    assert np.array_equal(full, acc.cpu().numpy())
    # oracle on the first queries
    for i in range(8):
        s, p, o = (int(v) for v in test[i])
        sc = (Ef[s].astype(np.float64) * Rf[p]) @ Ef.astype(np.float64).T
        keepm = np.ones(N, bool)
        keepm[o] = False
        keepm[fi[fq == i]] = False
        ref = int(np.sum(sc[keepm] > sc[o]))
        gap = np.min(np.abs(sc[keepm] - sc[o]))
        assert full[i] == ref or gap < 1e-5 * np.abs(sc).max(), (i, full[i], ref)


@pytest.mark.parametrize("N,d,nq,gamma", [(14, 16, 40, 12), (300, 100, 257, 12), (1000, 15, 64, 0)])
def test_rank_parity_protate(N, d, nq, gamma):
    """pRotatE (models.py:461-489) ranks through the distance sweep with the |sin(x/2)| element function."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(11)
    E0 = rs.uniform(-2, 2, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-2, 2, (7, 2 * d)).astype(np.float32)
    mdl = M.pRotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=7, dp=0.0, gamma=gamma, modulus=4.0, init_entities=E0, init_relations=R0)
    test = list({(int(rs.randint(N)), int(rs.randint(7)), int(rs.randint(N))) for _ in range(nq)})
    tes, filt = set(test), set()
    for (s, p, o) in test:
        for _ in range(int(rs.randint(0, 5))):
            c = (s, p, int(rs.randint(N)))
            if c not in tes:
                filt.add(c)
            c = (int(rs.randint(N)), p, o)
            if c not in tes:
                filt.add(c)
    filt = sorted(filt)
    fh, ft = O.build_filters(filt)
    tails, heads = mdl.rank(test, N, ft, fh)
    gaps = []
    to, ho = O.ranks_closed_form("pRotatE", E0, R0, test, filt, mdl.modulus, gamma, margin_out=gaps)
    gaps = np.asarray(gaps).reshape(-1, 2)
    tol = 2e-6 * 2 * d * mdl.modulus + 1e-6
    for got, ref, gap in ((tails, to, gaps[:, 0]), (heads, ho, gaps[:, 1])):
        clear = gap > tol
        assert clear.mean() > 0.5
        assert np.array_equal(got[clear], ref[clear])
        assert np.all(np.abs(got[~clear] - ref[~clear]) <= 2)


@pytest.mark.parametrize("gamma", [12, 0])
@pytest.mark.parametrize("N,d,nq", [(700, 200, 300), (130, 12, 50), (2000, 100, 129)])
def test_rank_transe_l2_mfma_vs_distance_sweep(N, d, nq, gamma, monkeypatch):
    """TransE-L2 1-vs-all on the matrix cores (order by |e|^2 - 2<q,e>) against the VALU distance sweep (KGE_RANK_L2=0) and the
    fp64 oracle: identical ranks wherever no candidate lies within fp32 rounding of the target; filters un-count exactly."""
    mdl, E0, R0, rs = _mk("TransE", N, 7, d, seed=13, norm=2, gamma=gamma)
    test = list({(int(rs.randint(N)), int(rs.randint(7)), int(rs.randint(N))) for _ in range(nq)})
    tes, filt = set(test), set()
    for (s, p, o) in test:
        for _ in range(int(rs.randint(0, 6))):
            c = (s, p, int(rs.randint(N)))
            if c not in tes:
                filt.add(c)
            c = (int(rs.randint(N)), p, o)
            if c not in tes:
                filt.add(c)
    filt = sorted(filt)
    fh, ft = O.build_filters(filt)
    monkeypatch.setenv("KGE_RANK_L2", "1")
    t1, h1 = mdl.rank(test, N, ft, fh)
    monkeypatch.setenv("KGE_RANK_L2", "0")
    t0, h0 = mdl.rank(test, N, ft, fh)
    gaps = []
    to, ho = O.ranks_closed_form("TransE", E0, R0, test, filt, 2, gamma, margin_out=gaps)
    gaps = np.asarray(gaps).reshape(-1, 2)
    tol = 1e-4  # distances here are ~sqrt(d); fp32 rounding of either form stays well below this
    for got1, got0, ref, gap in ((t1, t0, to, gaps[:, 0]), (h1, h0, ho, gaps[:, 1])):
        clear = gap > tol
        assert clear.mean() > 0.3
        assert np.array_equal(got1[clear], ref[clear]) and np.array_equal(got0[clear], ref[clear])
        assert np.all(np.abs(got1[~clear] - ref[~clear]) <= 2)
    # quantised tables: every quantity is exact in fp32 on both paths -> identical ranks, ties optimistic
    mq, Eq, Rq, rs = _mk("TransE", 300, 5, 16, seed=4, quant=2, norm=2, gamma=gamma)
    tq = list({(int(rs.randint(300)), int(rs.randint(5)), int(rs.randint(300))) for _ in range(120)})
    monkeypatch.setenv("KGE_RANK_L2", "1")
    a = mq.rank(tq, 300, {}, {})
    monkeypatch.setenv("KGE_RANK_L2", "0")
    b = mq.rank(tq, 300, {}, {})
    ro = O.ranks_closed_form("TransE", Eq, Rq, tq, [], 2, gamma)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], ro[0]) and np.array_equal(a[1], ro[1])


def test_rank_protate_pair_tables_vs_direct(monkeypatch):
    """pRotatE ranks on (sin, cos) pair tables (3 VALU operations per element) by default; KGE_RANK_PHASE=0 keeps one sinf per element.
    Same ranks wherever no candidate lies within rounding of the target, with filters and candidate shards."""
    from kge_keras_amd import models as M
    N, d, nq = 900, 50, 200
    rs = np.random.RandomState(21)
    E0 = rs.uniform(-3, 3, (N, 2 * d)).astype(np.float32)
    R0 = rs.uniform(-3, 3, (5, 2 * d)).astype(np.float32)
    mdl = M.pRotatE(e_dim=d, r_dim=d, num_entities=N, num_relations=5, dp=0.0, gamma=12, modulus=2.0, init_entities=E0, init_relations=R0)
    test = np.stack([rs.randint(0, N, nq), rs.randint(0, 5, nq), rs.randint(0, N, nq)], 1).astype(np.int32)
    fq = np.repeat(np.arange(nq, dtype=np.int32), 2)
    fi = rs.randint(0, N, 2 * nq).astype(np.int32)
    keep = (fi != np.repeat(test[:, 2], 2)) & (fi != np.repeat(test[:, 0], 2))
    fq, fi = fq[keep], fi[keep]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("KGE_RANK_PHASE", mode)
        out[mode] = [mdl.rank_counts(test, side, fq, fi).cpu().numpy() for side in (0, 1)]
    for side in (0, 1):
        diff = np.abs(out["1"][side] - out["0"][side])
        assert (diff == 0).mean() > 0.9 and diff.max() <= 2, (side, diff.max(), (diff == 0).mean())
    monkeypatch.setenv("KGE_RANK_PHASE", "1")
    acc = None
    for a, b in ((0, 300), (300, 900)):
        acc = mdl.rank_counts(test, 0, fq, fi, cand_offset=a, cand_rows=b - a, counts=acc)
    assert np.array_equal(acc.cpu().numpy(), out["1"][0])


def test_rank_fuzz_shapes_vs_chain_scores():
    """Seeded sweep of awkward shapes (tables smaller than a tile, one query, K with and without chunk padding, candidate shards that cut
    tiles) through the persistent MFMA sweep: counts equal the counts taken on the chain scores bit for bit (DistMult / ComplEx)."""
    from kge_keras_amd import _lib
    rs = np.random.RandomState(77)
    for case in range(24):
        model = ("DistMult", "ComplEx")[case % 2]
        N = int(rs.choice([1, 2, 127, 128, 129, 255, 257, 640, 1111]))
        d = int(rs.choice([4, 8, 12, 36, 40, 44, 100, 132]))
        nq = int(rs.choice([1, 2, 63, 128, 129, 300, 515]))
        mdl, E0, R0, _ = _mk(model, N, 3, d, seed=case)
        test = np.stack([rs.randint(0, N, nq), rs.randint(0, 3, nq), rs.randint(0, N, nq)], 1).astype(np.int32)
        side = int(rs.randint(2))
        empty = np.zeros(0, np.int32)
        cnt = mdl.rank_counts(test, side, empty, empty).cpu().numpy()
        md = mdl._model_desc(False)
        out = torch.empty((nq, N), dtype=torch.float32, device="cuda")
        t = torch.from_numpy(test).cuda()
        _lib.check(_lib.lib.kge_rank_chain_scores(C.byref(md), mdl.E.data_ptr(), mdl.Rel.data_ptr(), t.data_ptr(), nq, side, out.data_ptr(),
                                                  mdl._rank_ws.data_ptr(), mdl._rank_ws.numel(), mdl._err.data_ptr(), _lib.stream_ptr()))
        sc = out.cpu().numpy()
        tgt = sc[np.arange(nq), test[:, 2] if side == 0 else test[:, 0]]
        assert np.array_equal(cnt, (sc > tgt[:, None]).sum(1)), (case, model, N, d, nq, side)
        if N >= 3:  # candidate shards with cuts inside tiles add up exactly
            cuts = sorted(set([0, N] + [int(x) for x in rs.randint(1, N, 2)]))
            acc = None
            for a, b in zip(cuts[:-1], cuts[1:]):
                acc = mdl.rank_counts(test, side, empty, empty, cand_offset=a, cand_rows=b - a, counts=acc)
            assert np.array_equal(acc.cpu().numpy(), cnt), (case, "shards", cuts)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("ComplEx", {}), ("TransE", dict(norm=1, gamma=12)), ("TransE", dict(norm=2, gamma=12))])
def test_resident_shard_schedule_equals_full_table(model, kw):
    """SURVEY 8(e) schedule (ii) on one GPU: the table cut into 3 row shards (the last one short, one EMPTY shard appended) that are
    never gathered; per shard kge_rank_counts with E = the rows the queries reference, E_cand = the shard, filters in global
    ids; counts summed over shards == the ranks of the full-table call, exactly (bit-identical comparisons)."""
    from kge_keras_amd import models as M
    from kge_keras_amd import dist as D
    N, R, d = 1000, 6, 32
    rs = np.random.RandomState(17)
    de, dr = O.table_dims(model, d, d)
    E0 = (rs.randint(-8, 9, (N, de)) / 8.0).astype(np.float32)   # quantised: exact arithmetic, true ties
    R0 = (rs.randint(-8, 9, (R, dr)) / 8.0).astype(np.float32)
    mdl = getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, **kw)
    test = np.stack([rs.randint(0, N, 300), rs.randint(0, R, 300), rs.randint(0, N, 300)], 1).astype(np.int32)
    fq = np.repeat(np.arange(300, dtype=np.int32), 4)
    fi = rs.randint(0, N, 1200).astype(np.int32)
    for side in (0, 1):
        keep = fi != np.repeat(test[:, 2 if side == 0 else 0], 4)
        q, f = torch.from_numpy(fq[keep]).cuda(), torch.from_numpy(fi[keep]).cuda()
        full = mdl.rank_counts(test, side, q, f).cpu().numpy()
        ids = D.referenced_rows(test)
        ref = mdl.E.index_select(0, torch.from_numpy(ids).cuda())
        local = test.copy()
        local[:, 0], local[:, 2] = np.searchsorted(ids, test[:, 0]), np.searchsorted(ids, test[:, 2])
        acc = None
        for lo, hi in ((0, 400), (400, 800), (800, 1000), (1000, 1000)):
            acc = mdl.rank_counts_shard(ref, local, side, q, f, mdl.E[lo:hi].clone(), lo, counts=acc)
        assert np.array_equal(acc.cpu().numpy(), full), (model, kw, side)
    mdl._raise_on_flag()


def test_rccl_wrappers_one_rank_communicator():
    """kge_allgather_rows / kge_allreduce_counts / kge_allreduce_f32 on a communicator the caller made: a one-rank RCCL communicator
    (all a one-GPU box allows), created through RCCL's own C API.  With one rank every collective is the identity -- what is under
    test is symbol resolution, argument marshalling, datatype codes and stream ordering."""
    import ctypes as C
    from kge_keras_amd import _lib
    import torch.distributed  # noqa: F401  (PyTorch-ROCm maps its own librccl; use that copy, as the library's run-time lookup does)
    path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln), "librccl.so.1")
    rccl = C.CDLL(path)
    class UniqueId(C.Structure):  # ncclUniqueId: passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rc = rccl.ncclGetUniqueId(C.byref(uid))
    assert rc == 0, ("ncclGetUniqueId", rc, path)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    assert rc == 0, ("ncclCommInitRank", rc, path)
    try:
        st = _lib.stream_ptr()
        shard = torch.arange(35 * 8, dtype=torch.float32, device="cuda").reshape(35, 8)
        full = torch.zeros_like(shard)
        _lib.check(_lib.lib.kge_allgather_rows(comm, shard.data_ptr(), 35, 8, full.data_ptr(), st))
        cnt = torch.arange(1000, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib.kge_allreduce_counts(comm, cnt.data_ptr(), cnt.numel(), st))
        g = torch.linspace(-1, 1, 4099, device="cuda")
        g0 = g.clone()
        _lib.check(_lib.lib.kge_allreduce_f32(comm, g.data_ptr(), g.numel(), st))
        torch.cuda.synchronize()
        assert torch.equal(full, shard) and torch.equal(cnt, torch.arange(1000, dtype=torch.int32, device="cuda")) and torch.equal(g, g0)
        assert _lib.lib.kge_allreduce_counts(None, cnt.data_ptr(), 4, st) == _lib.KGE_EINVAL
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("ComplEx", {}), ("HolE", {}), ("RotatE", {}), ("pRotatE", {}), ("TransE", dict(norm=1, gamma=12)),
                                      ("TransE", dict(norm=2, gamma=12)), ("TransE", dict(norm=3.5, gamma=12))])
@pytest.mark.parametrize("N,d,nq", [(300, 40, 257), (1000, 24, 513)])
def test_rank_both_sides_in_one_sweep(model, kw, N, d, nq):
    """kge_rank_counts with side = 2 (tail queries [0, n), head queries [n, 2n), one sweep of the table where the model ranks on the
    matrix cores, the two sides one after the other inside the call elsewhere) == side 0 and side 1 apart, bit for bit, with an
    UNSORTED filter list that mixes pairs of both sides, and when accumulating over two row shards."""
    mdl, E0, R0, rs = _mk(model, N, 5, d, **kw)
    test = np.stack([rs.randint(0, N, nq), rs.randint(0, 5, nq), rs.randint(0, N, nq)], 1).astype(np.int32)
    def pairs(col):
        q = rs.randint(0, nq, 4 * nq).astype(np.int32)
        ids = rs.randint(0, N, 4 * nq).astype(np.int32)
        keep = ids != test[q, col]
        return q[keep], ids[keep]
    (q0, i0), (q1, i1) = pairs(2), pairs(0)
    c0 = mdl.rank_counts(test, 0, q0, i0).cpu().numpy()
    c1 = mdl.rank_counts(test, 1, q1, i1).cpu().numpy()
    fq, fi = np.concatenate([q0, q1 + nq]), np.concatenate([i0, i1])
    perm = rs.permutation(len(fq))
    both = mdl.rank_counts(test, 2, fq[perm], fi[perm]).cpu().numpy()
    assert both.shape == (2 * nq,)
    assert np.array_equal(both[:nq], c0) and np.array_equal(both[nq:], c1)
    cut = N // 3
    acc = mdl.rank_counts(test, 2, fq[perm], fi[perm], cand_offset=0, cand_rows=cut)
    acc = mdl.rank_counts(test, 2, fq[perm], fi[perm], cand_offset=cut, cand_rows=N - cut, counts=acc).cpu().numpy()
    assert np.array_equal(acc, both)
    mdl._raise_on_flag()
