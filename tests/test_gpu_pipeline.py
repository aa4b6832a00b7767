"""GPU tests of the three-stage training pipeline (draw | plan | step, hipGraph replays) against the host generator + oracle."""
import random

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


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
    assert gen._pipe.use_graphs and gen._pipe.graphs  # the stages really were graph replays
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
    np.testing.assert_allclose(mdl.E.cpu().numpy(), E0, atol=TOL)
    np.testing.assert_allclose(mdl.Rel.cpu().numpy(), R0, atol=TOL)
    np.testing.assert_allclose(h.history["loss"], losses, rtol=1e-4)


def test_pipeline_stream_launches_with_dropout(datasets):
    """dropout > 0: no graphs (the mask seed changes every step), no folding (every row has its own mask); the pipeline then
    launches its stages kernel by kernel and must equal train_step on the same batches."""
    from kge_keras_amd.harness import DeviceDataGenerator, build_model
    hp = dict(num_entities=14, num_relations=55, dim=16, embedding_model="TransE", loss_function="pointwize_hinge", margin=1, dp=0.25, seed=3)
    out = []
    for mode in ("pipeline", "steps"):
        np.random.seed(5)
        random.seed(6)
        mdl = build_model(hp)
        gen = DeviceDataGenerator(datasets["nations_train"], ns=20, batch_size=128)
        if mode == "pipeline":
            mdl.fit(gen, epochs=1, verbose=0, shuffle=False)
            assert not gen._pipe.use_graphs
        else:
            for i in range(len(gen)):
                mdl.train_step(gen[i][0])
        out.append(mdl.E.cpu().numpy())
    np.testing.assert_allclose(out[0], out[1], atol=2e-5)


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
