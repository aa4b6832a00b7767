"""GPU tests of the training pipeline (generator rounds | draw | plan | step; native driver and Python loop) against the host generator + oracle."""
import random

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
from tols import TOL, TOL_W, TOL_L, TOL_LA, assert_tables  # noqa: E402  (tests/tols.py: what each constant bounds)


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_pipeline_epochs_equal_host_generator_and_oracle(datasets, opt):
    """fit() over the DeviceDataGenerator runs TrainPipeline: device sampler -> folded batch plans -> planned steps with the
    device-resident optimizer state, every stage replayed from a hipGraph.  Same batches as the reference's generator
    (bit-exact streams) and, step by step, the oracle's tables.  Batches of 2 x 6400 rows, so the duplicate-row fold is on."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    hp = dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=9)
    np.random.seed(11)
    random.seed(12)
    mdl = build_model(hp)
    mdl.compile(opt)
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    h = mdl.fit(gen, epochs=2, verbose=0, shuffle=False)
    assert gen._pipe.use_graphs and gen._pipe._h is not None  # the stages really were graph replays, issued by the native driver
    assert mdl.optimizer.iterations == 2 * len(gen)
    after = (np.random.randint(0, 10**6, 2).tolist(), random.random())
    # reference path: host generator (global streams) + oracle steps
    np.random.seed(11)
    random.seed(12)
    hgen = DataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    oo = O.make_optimizer(opt)
    losses = []
    for _ in range(2):
        tot = 0.0
        for i in range(len(hgen)):
            (pos, neg), _y = hgen[i]
            tot += O.train_step("DistMult", E0, R0, oo, pos, neg, "pointwize_logistic", 1.0, 1)
        losses.append(tot / len(hgen))
        hgen.on_epoch_end()
    assert (np.random.randint(0, 10**6, 2).tolist(), random.random()) == after  # both leave the global streams in the same place
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    np.testing.assert_allclose(h.history["loss"], losses, rtol=1e-4)


def test_pipeline_stream_launches_with_dropout(datasets):
    """dropout > 0: no folding (every row has its own mask) and a new mask seed every step; the native driver then moves the batches
    as plain arrays (kge_pipeline_desc.pos / neg, kge_pipeline_set_seed) and must equal train_step on the same batches -- also with
    the Python loop (serial launches)."""
    from kge_keras_amd.harness import DeviceDataGenerator, build_model
    hp = dict(num_entities=14, num_relations=55, dim=16, embedding_model="TransE", loss_function="pointwize_hinge", margin=1, dp=0.25, seed=3)
    out = []
    for mode in ("pipeline", "python-loop", "steps"):
        np.random.seed(5)
        random.seed(6)
        mdl = build_model(hp)
        gen = DeviceDataGenerator(datasets["nations_train"], ns=20, batch_size=128)
        if mode == "pipeline":
            mdl.fit(gen, epochs=2, verbose=0, shuffle=False)
            assert not gen._pipe.use_graphs and gen._pipe._h is not None and not gen._pipe.counted  # native driver, arrays mode
        elif mode == "python-loop":
            from kge_keras_amd.pipeline import TrainPipeline
            gen._pipe = TrainPipeline(mdl, gen.kg, gen.n, gen.batch_size, gen.ns, gen.num_e, gen.sampler, graphs=True)
            gen._pipe.native_wanted = False
            mdl.fit(gen, epochs=2, verbose=0, shuffle=False)
            assert gen._pipe._h is None
        else:
            for _ in range(2):
                for i in range(len(gen)):
                    mdl.train_step(gen[i][0])
                gen.on_epoch_end()
        out.append((mdl.E.cpu().numpy(), mdl.Rel.cpu().numpy(), mdl.optimizer.iterations))
    for k in (1, 2):
        np.testing.assert_allclose(out[0][0], out[k][0], atol=2e-5)
        np.testing.assert_allclose(out[0][1], out[k][1], atol=2e-5)
        assert out[0][2] == out[k][2]


def test_indexed_batch_builder_walks_the_epoch(datasets):
    """kge_build_batch_device_indexed: positives read in place through (index, starts, cursor) == the (B,3) batch variant, and
    the device cursor advances by one per call."""
    from kge_keras_amd.sampler import DeviceSampler, NumpyStream, PythonStream
    kg = torch.from_numpy(np.ascontiguousarray(datasets["nations_train"], dtype=np.int32)).cuda()
    n, B, ns, N = kg.shape[0], 50, 7, 14
    index = torch.from_numpy(np.random.RandomState(2).permutation(n)).cuda()
    starts = torch.tensor([3, 0, 7], dtype=torch.int64, device="cuda")
    cursor = torch.zeros(1, dtype=torch.int64, device="cuda")
    a, b = DeviceSampler(NumpyStream(4), PythonStream(5)), DeviceSampler(NumpyStream(4), PythonStream(5))
    for k in range(3):
        pa, na = a.build_batch(None, ns, N, indexed=(kg, index, starts, cursor, B))
        s0 = int(starts[k]) * B
        batch = kg.index_select(0, index[s0:s0 + B]).contiguous()
        pb, nb = b.build_batch(batch, ns, N)
        assert torch.equal(pa, pb) and torch.equal(na, nb)
        assert int(cursor.item()) == k + 1
    assert torch.equal(a.np_state, b.np_state) and torch.equal(a.py_state, b.py_state)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=1, gamma=12)), ("ComplEx", {})])
def test_counted_plan_equals_plain_step(model, kw):
    """kge_batch_plan_build_counted: a batch given as candidate rows + multiplicities (what the device sampler emits next to the
    oversampled arrays; no hash table) trains exactly like the expanded arrays.  Plain count vector (pos_lanes = 1), rows with
    count 0, a row outside the tables."""
    from kge_keras_amd import models as M
    N, R, d = 40, 5, 16
    rs = np.random.RandomState(8)
    de, dr = O.table_dims(model, d, d)
    E0 = rs.uniform(-0.5, 0.5, (N, de)).astype(np.float32)
    R0 = rs.uniform(-0.5, 0.5, (R, dr)).astype(np.float32)
    a, b = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                              loss_function="pointwize_logistic", **kw) for _ in range(2))
    a.compile("adam")
    b.compile("adam")
    Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
    oo = O.AdamTF()
    for step in range(2):
        n_pos, n_neg = 20, 300
        cand = np.stack([rs.randint(0, N, n_pos + n_neg), rs.randint(0, R, n_pos + n_neg), rs.randint(0, N, n_pos + n_neg)], 1).astype(np.int32)
        cnt = rs.randint(0, 6, n_pos + n_neg).astype(np.int32)  # zeros included
        cnt[0] = max(cnt[0], 1)
        cnt[n_pos] = max(cnt[n_pos], 1)
        pos, neg = np.repeat(cand[:n_pos], cnt[:n_pos], axis=0), np.repeat(cand[n_pos:], cnt[n_pos:], axis=0)
        rs.shuffle(pos)
        plan = a.plan_batch_counted(torch.from_numpy(cand).cuda(), torch.from_numpy(cnt).cuda(), n_pos, len(pos), len(neg))
        La = float(a.train_step(plan)["loss"].item())
        Lb = float(b.train_step([pos, neg])["loss"].item())
        Lo = O.train_step(model, Eo, Ro, oo, pos, neg, "pointwize_logistic", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
        assert La == pytest.approx(Lb, rel=1e-5, abs=1e-6) and La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), step
        assert_tables(a.E.cpu().numpy(), Eo)
        assert_tables(a.Rel.cpu().numpy(), Ro)
        np.testing.assert_allclose(a.E.cpu().numpy(), b.E.cpu().numpy(), atol=2e-5)
    cand[3, 0] = N  # a picked row outside the tables: error flag, the row takes no part
    cnt[3] = 2
    a.train_step(a.plan_batch_counted(torch.from_numpy(cand).cuda(), torch.from_numpy(cnt).cuda(), n_pos, int(cnt[:n_pos].sum()), int(cnt[n_pos:].sum())))
    with pytest.raises(IndexError):
        a._raise_on_flag()


def test_sampler_candidates_and_pick_counts(datasets):
    """DeviceSampler.build_batch(..., counted=...): the candidate rows are the batch before oversampling and expanding them by
    their pick counts gives the same multiset of rows as the oversampled arrays."""
    from kge_keras_amd import _lib
    from kge_keras_amd.sampler import DeviceSampler, NumpyStream, PythonStream
    kg = torch.from_numpy(np.ascontiguousarray(datasets["nations_train"], dtype=np.int32)).cuda()
    n, B, ns, N, L = kg.shape[0], 64, 9, 14, _lib.CAND_POS_LANES
    m = B * ns
    index = torch.from_numpy(np.random.RandomState(3).permutation(n)).cuda()
    starts = torch.tensor([2, 5], dtype=torch.int64, device="cuda")
    cursor = torch.zeros(1, dtype=torch.int64, device="cuda")
    ds = DeviceSampler(NumpyStream(7), PythonStream(8))
    cand = torch.empty((B + m, 3), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(L * B + m, dtype=torch.int32, device="cuda")
    for k in range(2):
        pos, neg = ds.build_batch(None, ns, N, indexed=(kg, index, starts, cursor, B), counted=(cand, cnt))
        c = cnt.cpu().numpy()
        cp, cn = c[:L * B].reshape(L, B).sum(0), c[L * B:]
        assert cp.sum() == m and cn.sum() == m
        s0 = int(starts[k]) * B
        assert torch.equal(cand[:B], kg.index_select(0, index[s0:s0 + B]))
        rows = cand.cpu().numpy()
        key = lambda a: sorted(map(tuple, a.tolist()))  # noqa: E731
        assert key(np.repeat(rows[:B], cp, axis=0)) == key(pos.cpu().numpy())
        assert key(np.repeat(rows[B:], cn, axis=0)) == key(neg.cpu().numpy())


def test_randomised_paths_agree_with_oracle():
    """Seeded sweep over random shapes: table sizes, row widths (all float4-chunk counts of the kernels' dispatch tables), batch
    sizes around the folding threshold, duplicate-heavy and duplicate-free batches, every pointwise loss.  One step through a
    plan built ahead must match the oracle and the plain step."""
    from kge_keras_amd import models as M
    rs = np.random.RandomState(2024)
    losses = ["pointwize_hinge", "pointwize_logistic", "pointwize_square_loss"]
    for case in range(24):
        model = ["DistMult", "TransE", "ComplEx"][case % 3]
        kw = dict(norm=int(rs.randint(1, 3)), gamma=float(rs.choice([12.0, 0.0]))) if model == "TransE" else {}
        N, R = int(rs.randint(2, 400)), int(rs.randint(1, 25))
        d = int(rs.choice([4, 8, 12, 32, 36, 64, 100, 128, 200, 256]))
        m_pos, m_neg = int(rs.choice([1, 7, 300, 2500, 5000])), int(rs.choice([1, 9, 700, 2100, 6000]))
        loss = losses[case % 3]
        de, dr = O.table_dims(model, d, d)
        E0 = rs.uniform(-0.4, 0.4, (N, de)).astype(np.float32)
        R0 = rs.uniform(-0.4, 0.4, (R, dr)).astype(np.float32)
        a, b = (getattr(M, model)(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0,
                                  loss_function=loss, margin=0.7, **kw) for _ in range(2))
        a.compile("adam")
        b.compile("adam")

        def draw(n):
            if rs.rand() < 0.5:  # oversampled: few distinct rows
                base = np.stack([rs.randint(0, N, 17), rs.randint(0, R, 17), rs.randint(0, N, 17)], 1)
                return base[rs.randint(0, 17, n)]
            return np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)
        pos, neg = draw(m_pos), draw(m_neg)
        Eo, Ro = E0.astype(np.float64), R0.astype(np.float64)
        Lo = O.train_step(model, Eo, Ro, O.AdamTF(), pos, neg, loss, 0.7, 1, kw.get("norm", 1), kw.get("gamma", 12))
        La = float(a.train_step(a.plan_batch([pos, neg]))["loss"].item())
        Lb = float(b.train_step([pos, neg])["loss"].item())
        tag = (case, model, kw, N, R, d, m_pos, m_neg, loss)
        assert La == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA) and Lb == pytest.approx(Lo, rel=TOL_L, abs=TOL_LA), tag
        np.testing.assert_allclose(a.E.cpu().numpy(), Eo, atol=TOL_W, err_msg=str(tag))
        np.testing.assert_allclose(b.E.cpu().numpy(), Eo, atol=TOL_W, err_msg=str(tag))
        np.testing.assert_allclose(a.Rel.cpu().numpy(), Ro, atol=TOL_W, err_msg=str(tag))
        np.testing.assert_allclose(b.Rel.cpu().numpy(), Ro, atol=TOL_W, err_msg=str(tag))


def test_pipeline_with_a_hook_between_the_halves(datasets):
    """TrainPipeline(between=...): the data-parallel shape of the pipeline (draw and plan stages as graph replays, the step as
    gradient half -> hook (the all-reduce) -> optimizer half with the host step counter).  With a hook that does nothing but
    return the gradient scale 1 it must train exactly like the plain pipeline."""
    from kge_keras_amd.harness import DeviceDataGenerator, build_model
    from kge_keras_amd.pipeline import TrainPipeline
    hp = dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=4)
    out = []
    for hooked in (False, True):
        np.random.seed(21)
        random.seed(22)
        mdl = build_model(hp)
        gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
        calls = []
        pipe = TrainPipeline(mdl, gen.kg, gen.n, gen.batch_size, gen.ns, gen.num_e, gen.sampler,
                             between=(lambda: calls.append(1) or 1.0) if hooked else None)
        gen._upload_global()
        tot = pipe.run_epoch(gen.index, list(range(len(gen))))
        gen.release_global()
        assert pipe.use_graphs and (pipe._h is not None) == (not hooked) and (("S", 0) in pipe.graphs) == hooked and ("T", 0) not in pipe.graphs
        assert len(calls) == (len(gen) if hooked else 0) and mdl.optimizer.iterations == len(gen)
        out.append((mdl.E.cpu().numpy(), float(tot.item())))
    np.testing.assert_allclose(out[0][0], out[1][0], atol=2e-5)
    assert out[0][1] == pytest.approx(out[1][1], rel=1e-5)


def test_device_generator_end_to_end_with_validation_and_callback(datasets):
    """The reference's workflow on the GPU-resident pipeline: build_model -> fit(DeviceDataGenerator, validation_data, callback)
    over many epochs (graph replays for the steps, plain steps for the validation loss, ranking in the callback) -> validate."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    from kge_keras_amd.utils import validate, KGEValidateCallback
    tr, va, te = ([tuple(int(v) for v in t) for t in datasets["nations_" + k].tolist()] for k in ("train", "valid", "test"))
    np.random.seed(0)
    random.seed(0)
    mdl = build_model(dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic",
                           margin=1, dp=0.0, seed=1))
    tes = set(te) | set(va)
    filt = [t for t in tr if t not in tes]
    before = validate(mdl, te, 14, None, filtering_triples=filt)
    cb = KGEValidateCallback(va, train_data=tr, period=10)
    gen = DeviceDataGenerator(tr, ns=10, batch_size=32)
    h = mdl.fit(gen, epochs=30, verbose=0, callbacks=[cb], validation_data=DataGenerator(va, ns=10, batch_size=32))
    after = validate(mdl, te, 14, None, filtering_triples=filt)
    assert gen._pipe.use_graphs and gen._pipe._h is not None and mdl.optimizer.iterations == 30 * len(gen) and getattr(mdl, "_step_dev", None) is None
    assert len(h.history["loss"]) == 30 and len(h.history["val_loss"]) == 30
    assert h.history["loss"][-1] < h.history["loss"][0]
    assert after["mrr"] > before["mrr"] + 0.1 and after["mr"] < 4.0, (before, after)


def test_device_generator_fit_hole_frequency_domain(datasets):
    """HolE without dropout takes batch plans but its step is a composite of library calls: fit() over the DeviceDataGenerator runs it
    batch by batch (no TrainPipeline) on the device-drawn batches -- same batches as the host generator, same tables as the oracle."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    hp = dict(num_entities=14, num_relations=55, dim=16, embedding_model="HolE", loss_function="pointwize_hinge", margin=1, dp=0.0, seed=5)
    np.random.seed(3)
    random.seed(4)
    mdl = build_model(hp)
    assert mdl.plan_supported() and not mdl.pipeline_supported()
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=10, batch_size=64)
    h = mdl.fit(gen, epochs=1, verbose=0, shuffle=False)
    np.random.seed(3)
    random.seed(4)
    hgen = DataGenerator(datasets["nations_train"], ns=10, batch_size=64)
    oo = O.make_optimizer("adam")
    tot = 0.0
    for i in range(len(hgen)):
        (pos, neg), _y = hgen[i]
        tot += O.train_step("HolE", E0, R0, oo, pos, neg, "pointwize_hinge", 1.0, 1)
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    np.testing.assert_allclose(h.history["loss"], [tot / len(hgen)], rtol=1e-4)


def test_pipeline_protate(datasets):
    """pRotatE (TransE kernels, norm id 3) through the pipeline: same tables as the host generator + oracle."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    hp = dict(num_entities=14, num_relations=55, dim=16, embedding_model="pRotatE", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=2)
    np.random.seed(7)
    random.seed(8)
    mdl = build_model(hp)
    assert mdl.pipeline_supported()
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    h = mdl.fit(gen, epochs=1, verbose=0, shuffle=False)
    assert gen._pipe.use_graphs and gen._pipe._h is not None
    np.random.seed(7)
    random.seed(8)
    hgen = DataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    oo = O.make_optimizer("adam")
    tot = 0.0
    for i in range(len(hgen)):
        (pos, neg), _y = hgen[i]
        tot += O.train_step("pRotatE", E0, R0, oo, pos, neg, "pointwize_logistic", 1.0, 1, mdl.modulus, 12)
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    np.testing.assert_allclose(h.history["loss"], [tot / len(hgen)], rtol=1e-4)


@pytest.mark.parametrize("model", ["DistMult", "TransE"])
def test_benched_path_at_benched_size(model):
    """The exact code path bench.py times, at its size: TrainPipeline with the device sampler, counted plans and one hipGraph per
    stage on the BASELINE configs[1] / configs[2] workload (N=14 541, R=237, d=200, B=2048, ns=100, 272 115 synthetic triples,
    bench.py's seeds).  Every drawn batch bit for bit against the host generator (rows before oversampling + pick counts, and the
    generator states left behind); loss and both tables after each of 3 steps against the oracle's C restatement AND the NumPy
    fp64 oracle: tables at 1e-6 absolute, loss at 1e-6 relative (tests.py:55-61,130-134)."""
    import bench
    res = bench.check_benched_path(model, steps=3, fp64=True, tol=2 * TOL_W, tol_loss_rel=TOL_L)  # (2e-6: TransE-L2 at d = 200 measures 9.5e-7)
    assert res["checked"] and res["check_batches_bit_exact"] == [0, 1, 2, 3, 4]
    assert max(res["max_abs_err_by"]["E"], res["max_abs_err_by"]["Rel"]) <= 2 * TOL_W


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_native_driver_hands_the_epoch_over_to_the_python_loop(datasets, k):
    """k steps issued by the native driver (four plan buffers: batch j sits in buffer j % 4), then `serial` flips mid-epoch and the
    Python loop finishes the epoch from the driver's buffers, with the step count and Adam's lr_t taken from the device step state.
    Tables, epoch loss and optimizer.iterations == host generator + oracle for every hand-over phase (draw buffers: k mod 2, plan buffers: k mod 4; odd and even
    chain lengths of the native driver's chained steps)."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    from kge_keras_amd.pipeline import TrainPipeline
    hp = dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_logistic", margin=1, dp=0.0, seed=9)
    np.random.seed(31)
    random.seed(32)
    mdl = build_model(hp)
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    n = 12
    # one plain step first: the hand-over must also be right when the optimizer has updates behind it (lr_t of update k + 2, not k + 1)
    first = gen[0][0]
    mdl.train_step(first)
    pipe = TrainPipeline(mdl, gen.kg, gen.n, gen.batch_size, gen.ns, gen.num_e, gen.sampler)
    gen._upload_global()
    pipe.load_epoch(gen.index, list(range(n)))
    pipe.start()
    assert pipe._native_active
    pipe.run(k)
    pipe.serial = True          # what a profiling pass does: the rest of the epoch on the Python loop, kernel by kernel
    pipe.run(n - k)
    assert not pipe._native_active and pipe._taken_over and pipe._np == len(pipe.plans) == 4
    tot = float(pipe.finish().item())
    gen.release_global()
    assert mdl.optimizer.iterations == n + 1
    mdl._raise_on_flag()
    np.random.seed(31)
    random.seed(32)
    hgen = DataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    oo = O.make_optimizer("adam")
    (p0, n0), _y = hgen[0]
    assert np.array_equal(p0, first[0].cpu().numpy())
    O.train_step("DistMult", E0, R0, oo, p0, n0, "pointwize_logistic", 1.0, 1)
    ref = 0.0
    for i in range(n):
        (pos, neg), _y = hgen[i]
        ref += O.train_step("DistMult", E0, R0, oo, pos, neg, "pointwize_logistic", 1.0, 1)
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    assert tot == pytest.approx(ref, rel=1e-5)


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_python_loop_with_captured_twin_step(datasets, opt):
    """The Python loop with every stage replayed from a hipGraph (native driver switched off) and the step between twin tables inside
    the captured step stage: Adam's lr_t must move on from replay to replay -- it is read from the device step state, not baked into
    the graph.  Two epochs (the second reuses the captured graphs) == host generator + oracle."""
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    from kge_keras_amd.pipeline import TrainPipeline
    hp = dict(num_entities=14, num_relations=55, dim=32, embedding_model="DistMult", loss_function="pointwize_hinge", margin=1, dp=0.0, seed=9)
    np.random.seed(41)
    random.seed(42)
    mdl = build_model(hp)
    mdl.compile(opt)
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    gen._pipe = TrainPipeline(mdl, gen.kg, gen.n, gen.batch_size, gen.ns, gen.num_e, gen.sampler, graphs=True)
    gen._pipe.native_wanted = False
    h = mdl.fit(gen, epochs=2, verbose=0, shuffle=False)
    pipe = gen._pipe
    assert pipe._h is None and pipe.use_graphs and ("T", 0) in pipe.graphs and pipe._twins is not None
    assert mdl.optimizer.iterations == 2 * len(gen)
    np.random.seed(41)
    random.seed(42)
    hgen = DataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    oo = O.make_optimizer(opt)
    losses = []
    for _ in range(2):
        tot = 0.0
        for i in range(len(hgen)):
            (pos, neg), _y = hgen[i]
            tot += O.train_step("DistMult", E0, R0, oo, pos, neg, "pointwize_hinge", 1.0, 1)
        losses.append(tot / len(hgen))
        hgen.on_epoch_end()
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    np.testing.assert_allclose(h.history["loss"], losses, rtol=1e-4)


@pytest.mark.parametrize("halves,loop", [(True, "native"), (False, "native"), (True, "python")])
@pytest.mark.parametrize("model,opt", [("DistMult", "adam"), ("TransE", "sgd")])
def test_native_data_parallel_step_on_a_one_rank_communicator(datasets, model, opt, halves, loop, monkeypatch):
    """kge_pipeline_desc.comm: the data-parallel step inside the native driver -- entity half of the gradients, all-reduce of its
    table on the communicator's own stream beside the relation half, all-reduce of the rest, optimizer half with scale 1 / world.
    One GPU allows a one-rank RCCL communicator (every collective the identity), which exercises everything but the wire: with it an
    epoch must equal the host generator + the oracle's steps, with the gradient halves apart and in one piece.  loop = "python": the
    same communicator under the Python loop (serial launches) -- the comm branch of its step stage, which must never be captured into a
    graph (a collective inside a hipGraph): graph_stages leaves the step out whenever comm is set."""
    from kge_keras_amd import dist as D
    from kge_keras_amd.harness import DataGenerator, DeviceDataGenerator, build_model
    from kge_keras_amd.pipeline import TrainPipeline
    if not halves:
        monkeypatch.setenv("KGE_PIPE_DP_HALVES", "0")
    hp = dict(num_entities=14, num_relations=55, dim=32, embedding_model=model, loss_function="pointwize_logistic", margin=1, dp=0.0, seed=9)
    np.random.seed(51)
    random.seed(52)
    mdl = build_model(hp)
    mdl.compile(opt)
    E0, R0 = mdl.E.cpu().numpy().astype(np.float64), mdl.Rel.cpu().numpy().astype(np.float64)
    gen = DeviceDataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    comm, destroy = D.make_rccl_comm()
    try:
        pipe = TrainPipeline(mdl, gen.kg, gen.n, gen.batch_size, gen.ns, gen.num_e, gen.sampler, comm=comm, world=1)
        assert "T" not in pipe.graph_stages
        if loop == "python":
            pipe.serial = True
        gen._upload_global()
        tot = float(pipe.run_epoch(gen.index, list(range(len(gen)))).item())
        gen.release_global()
        assert (pipe._h is not None) == (loop == "native") and pipe._twins is None and mdl.optimizer.iterations == len(gen)
        pipe.close()
    finally:
        destroy()
    mdl._raise_on_flag()
    np.random.seed(51)
    random.seed(52)
    hgen = DataGenerator(datasets["nations_train"], ns=100, batch_size=64)
    oo = O.make_optimizer(opt)
    ref = 0.0
    for i in range(len(hgen)):
        (pos, neg), _y = hgen[i]
        ref += O.train_step(model, E0, R0, oo, pos, neg, "pointwize_logistic", 1.0, 1)
    assert_tables(mdl.E.cpu().numpy(), E0)
    assert_tables(mdl.Rel.cpu().numpy(), R0)
    assert tot == pytest.approx(ref, rel=1e-5)


@pytest.mark.parametrize("model,kw", [("DistMult", {}), ("TransE", dict(norm=2, gamma=12)), ("ComplEx", {})])
def test_deterministic_mode_two_seeded_runs_are_bit_identical(model, kw):
    """kge_set_deterministic(1) (KGE_DETERMINISTIC=1): a row's place in its relation / entity lists is its rank BY BATCH ROW (two stable
    sorts per plan) instead of the arrival order of plan_hist_kernel's atomics, so two runs of fit(DeviceDataGenerator) from the same
    seeds -- the native pipeline, device sampler, steps between twin tables, hub entities (chunk waves + ticket) included -- leave
    bit-identical tables and epoch losses; and the deterministic run agrees with the default one to fp32 rounding (same sums, another
    order).  SURVEY 7 / VERDICT r3 #7."""
    import random
    from kge_keras_amd import _lib
    from kge_keras_amd.harness import DeviceDataGenerator, build_model
    N, R, n_train = 600, 9, 6000
    rs = np.random.RandomState(11)
    ent = lambda n: np.where(rs.rand(n) < 0.2, 5, rs.randint(0, N, n))  # a hub entity: a list far longer than one chunk
    train = np.stack([ent(n_train), rs.randint(0, R, n_train), ent(n_train)], 1)

    def run():
        np.random.seed(1)
        random.seed(2)
        mdl = build_model(dict(num_entities=N, num_relations=R, dim=64, embedding_model=model, loss_function="pointwize_logistic", margin=1, dp=0.0, seed=3, **kw))
        h = mdl.fit(DeviceDataGenerator(train, ns=20, batch_size=256), epochs=2, verbose=0)
        torch.cuda.synchronize()
        return mdl.E.clone(), mdl.Rel.clone(), list(h.history["loss"])
    assert _lib.lib.kge_get_deterministic() == 0
    base = run()
    _lib.check(_lib.lib.kge_set_deterministic(1))
    try:
        a, b = run(), run()
    finally:
        _lib.check(_lib.lib.kge_set_deterministic(0))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    assert float((a[0] - base[0]).abs().max()) < 2e-5 and a[2][-1] == pytest.approx(base[2][-1], rel=1e-5)
    assert a[2][-1] < a[2][0]
