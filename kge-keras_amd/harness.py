"""Counterpart of the reference's training script pieces (tests.py:29-61 DataGenerator, tests.py:63-89
build_model): the caller of the hot path.  Batches come out of the host-side bit-exact samplers, and
are handed to EmbeddingModel.train_step as int32 (m,3) arrays."""
import numpy as np

from . import models as _models
from .sampler import NumpyStream, PythonStream, build_batch
from .utils import loss_function_lookup

models = {'DistMult': _models.DistMult, 'TransE': _models.TransE, 'HolE': _models.HolE, 'ComplEx': _models.ComplEx,
          'ConvE': _models.ConvE, 'pRotatE': _models.pRotatE, 'RotatE': _models.RotatE, 'HAKE': _models.HAKE}  # tests.py:18-27, restricted to the path's model families


class DataGenerator:
    """tests.py:29-61.  Consumes NumPy's and Python's *global* generators exactly like the reference
    (epoch shuffle -> randint heads -> randint tails -> choices positives -> choices negatives), so that
    `np.random.seed(a); random.seed(b)` gives the same batches as the reference's generator."""

    def __init__(self, kg, ns=10, batch_size=32, shuffle=True):
        self.batch_size = min(batch_size, len(kg))
        self.kg = np.ascontiguousarray(np.asarray(kg, dtype=np.int32).reshape(-1, 3))
        self.ns = ns
        self.num_e = len(set(self.kg[:, 0].tolist()) | set(self.kg[:, 2].tolist()))  # tests.py:34
        self.shuffle = shuffle
        self.indices = np.arange(len(self.kg))
        self.on_epoch_end()

    def __len__(self):
        return len(self.kg) // self.batch_size

    def __getitem__(self, index):
        index = self.index[index * self.batch_size:(index + 1) * self.batch_size]
        batch = self.kg[self.indices[index]]
        nps, pys = NumpyStream().adopt_global(), PythonStream().adopt_global()
        pos, neg = build_batch(nps, pys, batch, self.ns, self.num_e)  # generate_negative + oversample_data
        nps.release_global()
        pys.release_global()
        return [pos, neg], None

    def on_epoch_end(self):
        self.index = np.arange(len(self.indices), dtype=np.int64)
        if self.shuffle:
            st = NumpyStream().adopt_global()
            st.shuffle(self.index)  # np.random.shuffle (tests.py:53)
            st.release_global()


class DeviceDataGenerator:
    """Same batches as DataGenerator, built on the GPU: the training triples and the epoch permutation live in HBM, every
    batch is produced by kge_build_batch_device (bit-exact MT19937 on the device) and never touches the host.  Like the
    reference's generator it draws from NumPy's / Python's GLOBAL generators: their states are uploaded when an epoch's
    iteration starts (`batches`) and written back when it ends, so `np.random.seed(a); random.seed(b)` + fit() consumes
    the global streams exactly like the host generator does.  The epoch shuffle (sequential Fisher-Yates, tests.py:50-53)
    runs on the host."""

    def __init__(self, kg, ns=10, batch_size=32, shuffle=True, device=None):
        import torch
        from .sampler import DeviceSampler
        self.batch_size = min(batch_size, len(kg))
        kg = np.ascontiguousarray(np.asarray(kg, dtype=np.int32).reshape(-1, 3))
        self.n = len(kg)
        self.ns = ns
        self.num_e = len(set(kg[:, 0].tolist()) | set(kg[:, 2].tolist()))  # tests.py:34
        self.shuffle = shuffle
        self.sampler = DeviceSampler(NumpyStream().adopt_global(), PythonStream().adopt_global(), device)
        self.device = self.sampler.device
        self.kg = torch.from_numpy(kg).to(self.device)
        self.on_epoch_end()

    def __len__(self):
        return self.n // self.batch_size

    def on_epoch_end(self):
        import torch
        index = np.arange(self.n, dtype=np.int64)
        if self.shuffle:
            st = NumpyStream().adopt_global()
            st.shuffle(index)  # np.random.shuffle (tests.py:53)
            st.release_global()
        self.index = torch.from_numpy(index).to(self.device)

    def _upload_global(self):
        self.sampler.load_streams(NumpyStream().adopt_global(), PythonStream().adopt_global())

    def release_global(self):
        """Write the generator states back into np.random / random (where the reference's generator would stand).  Raises if
        the device sampler flagged a short rejection buffer since the last check (its draws would not be the reference's)."""
        err = int(self.sampler.err.item())
        if err:
            self.sampler.err.zero_()
            raise RuntimeError(f"device sampler error flag {err}: the masked-rejection buffer ran short (kge_build_batch_device)")
        nps, pys = self.sampler.to_streams()
        nps.release_global()
        pys.release_global()

    def __getitem__(self, i):
        """One batch, synchronously (global states up, batch, global states down): the compatible slow path."""
        self._upload_global()
        sl = self.index[i * self.batch_size:(i + 1) * self.batch_size]
        batch = self.kg.index_select(0, sl).contiguous()
        pos, neg = self.sampler.build_batch(batch, self.ns, self.num_e)
        self.release_global()
        return [pos, neg], None

    def run_epoch(self, model, order, graphs=True):
        """One epoch of model training over the batches `order` through the GPU pipeline (pipeline.TrainPipeline:
        draw batch t+2 | plan batch t+1 | train step t, steady state replayed from hipGraphs).  Same draws and the same
        updates as `for i in order: model.train_step(self[i][0])`.  Returns the summed loss (device tensor)."""
        from .pipeline import TrainPipeline
        pipe = getattr(self, "_pipe", None)
        if pipe is None or pipe.model is not model or pipe.use_graphs != (bool(graphs) and float(model.dp or 0.0) == 0.0):
            pipe = self._pipe = TrainPipeline(model, self.kg, self.n, self.batch_size, self.ns, self.num_e, self.sampler, graphs=graphs)
        self._upload_global()
        try:
            return pipe.run_epoch(self.index, order)
        finally:
            self.release_global()

    def batches(self, order, plan=None):
        """Iterate the batches `order` with the sampler running on a side stream ahead of the consumer (double-buffered in
        HBM, event fork/join): `fit` uses this, so drawing a batch overlaps the previous train steps.  Same draws, in the
        same order, as calling self[i] for i in order.  `plan`: optional callable ([pos, neg], out=previous) -> BatchPlan
        (EmbeddingModel.plan_batch); it runs on a third stream right behind the sampler and the iterator then yields the
        plans instead of the arrays."""
        import torch
        main, side = torch.cuda.current_stream(), torch.cuda.Stream()
        pstream = torch.cuda.Stream() if plan is not None else None
        m = self.batch_size * self.ns
        bufs = [tuple(torch.empty((m, 3), dtype=torch.int32, device=self.device) for _ in range(2)) for _ in range(2)]
        plans = [None, None]
        ready = [torch.cuda.Event() for _ in range(2)]     # batch arrays drawn
        freed = [torch.cuda.Event() for _ in range(2)]     # batch arrays no longer needed
        planned = [torch.cuda.Event() for _ in range(2)]   # plan built
        pfreed = [torch.cuda.Event() for _ in range(2)]    # plan consumed
        order = list(order)
        self._upload_global()
        side.wait_stream(main)
        if pstream is not None:
            pstream.wait_stream(main)

        def produce(k):
            j = k % 2
            with torch.cuda.stream(side):
                side.wait_event(freed[j])
                i = order[k]
                sl = self.index[i * self.batch_size:(i + 1) * self.batch_size]
                batch = self.kg.index_select(0, sl).contiguous()
                self.sampler.build_batch(batch, self.ns, self.num_e, *bufs[j])
                ready[j].record(side)
            if pstream is not None:
                with torch.cuda.stream(pstream):
                    pstream.wait_event(ready[j])
                    pstream.wait_event(pfreed[j])
                    plans[j] = plan(list(bufs[j]), out=plans[j])
                    planned[j].record(pstream)
                    freed[j].record(pstream)

        for k in range(min(2, len(order))):
            produce(k)
        for k in range(len(order)):
            j = k % 2
            if pstream is not None:
                main.wait_event(planned[j])
                plans[j].buf.record_stream(main)
                yield plans[j]
                pfreed[j].record(main)
            else:
                main.wait_event(ready[j])
                yield list(bufs[j])
                freed[j].record(main)
            if k + 2 < len(order):
                produce(k + 2)
        main.wait_stream(side)
        if pstream is not None:
            main.wait_stream(pstream)
        self.release_global()


def build_model(hp):
    """tests.py:63-89: instantiate the embedding model from the hp dict (all keys passed as kwargs), attach the
    looked-up loss with `margin or 1` and negative_samples=1, compile with Adam."""
    params = hp.copy()
    params['e_dim'] = params['dim']
    params['r_dim'] = params['dim']
    params['name'] = 'embedding_model'
    loss_function_lookup(params['loss_function'])  # KeyError on unknown names, as at tests.py:80
    model = models[params['embedding_model']](**params)
    model.margin = params.get('margin') or 1
    model.loss_negative_samples = 1
    model.compile(optimizer='adam', loss=None)
    return model


def main(train, valid, test, dim=100, bs=2048, epochs=100, verbose=2, device_generator=False, **hp):
    """tests.py:96-134 (the reference's training script): id maps over the three splits, random 5-d literals behind a tanh
    Dense, DistMult, pairwize_hinge, Adam, `fit` over the generator with the validation split.  train / valid / test: paths
    (utils.load_kg) or lists of (s, p, o) string triples.  Keyword arguments override the script's hyper-parameters;
    device_generator=True draws the batches on the GPU (same draws).  Returns (model, history)."""
    from .utils import index_kg, load_indexed, load_kg
    if all(isinstance(x, str) for x in (train, valid, test)):  # three files: read and indexed by the native front-end
        ents, rels, (train, valid, test) = load_indexed(train, valid, test)
    else:
        ents, rels, (train, valid, test) = index_kg(*[load_kg(x) if isinstance(x, str) else list(x) for x in (train, valid, test)])
    literals = np.random.rand(len(ents), 5)  # tests.py:113
    params = {'num_entities': len(ents), 'num_relations': len(rels), 'dim': dim, 'embedding_model': 'DistMult', 'literals': literals,
              'literal_activation': 'tanh', 'loss_function': 'pairwize_hinge', 'margin': 1}
    params.update(hp)
    if params.get('literal_activation') is None:
        params.pop('literals', None)
    model = build_model(params)
    gen = DeviceDataGenerator if device_generator else DataGenerator
    history = model.fit(gen(train, batch_size=bs), validation_data=DataGenerator(valid, batch_size=bs), epochs=epochs, batch_size=bs,
                        verbose=verbose)
    return model, history
