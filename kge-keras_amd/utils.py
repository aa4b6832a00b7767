"""Counterpart of KGEkeras/utils.py for the hot path: batch construction (bit-exact with the reference's
NumPy / `random` draws), the loss-name table, and filtered link-prediction evaluation.

Reference lines (relative to the reference repo) are cited per function.  LiteralConverter (utils.py:23-104)
is preprocessing outside the hot path and is not provided.
"""
from collections import defaultdict

import numpy as np

from . import _lib
from .sampler import NumpyStream, PythonStream

EPSILON = 1e-6  # utils.py:12


def load_kg(path):
    """utils.py:106-112: whitespace-split triples of strings."""
    out = []
    with open(path, "r") as f:
        for l in f:
            out.append(l.strip().split())
    return out


def index_kg(*splits):
    """Deterministic replacement for tests.py:102-117 (whose entity / relation ids come from enumerating Python sets of strings
    and so change with PYTHONHASHSEED): ids are the ranks of the sorted strings over all given splits.
    Returns (entity_map, relation_map, [int32 (n,3) array per split])."""
    ents = sorted({x for sp in splits for (s, _, o) in sp for x in (s, o)})
    rels = sorted({p for sp in splits for (_, p, _) in sp})
    emap = {e: i for i, e in enumerate(ents)}
    rmap = {r: i for i, r in enumerate(rels)}
    arrs = [np.asarray([(emap[s], rmap[p], emap[o]) for s, p, o in sp], dtype=np.int32).reshape(-1, 3) for sp in splits]
    return emap, rmap, arrs


def load_indexed(*paths):
    """The data front-end in one native call (csrc/frontend.cpp: kge_kg_open): reads the split files like load_kg and maps the names
    to ids like index_kg -- same maps, same arrays.  Returns (entity_map, relation_map, [int32 (n,3) array per split]).
    FileNotFoundError / ValueError where the reference's script would raise them (open(); `for a, b, c in train`)."""
    import ctypes as C
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    h = C.c_void_p()
    rc = _lib.lib.kge_kg_open(arr, n, C.byref(h))
    if rc < 0:
        msg = _lib.lib.kge_last_error().decode("utf-8", "replace")
        raise (FileNotFoundError if msg.startswith("[Errno 2]") else ValueError)(msg)
    try:
        ne, nr = C.c_int64(0), C.c_int64(0)
        rows = (C.c_int64 * n)()
        _lib.check(_lib.lib.kge_kg_sizes(h, C.byref(ne), C.byref(nr), rows))
        arrs = []
        for i in range(n):
            a = np.empty((int(rows[i]), 3), dtype=np.int32)
            _lib.check(_lib.lib.kge_kg_triples(h, i, a.ctypes.data))
            arrs.append(a)
        maps = []
        for which, cnt in ((0, ne.value), (1, nr.value)):
            offs = np.zeros(cnt + 1, dtype=np.int64)
            _lib.check(_lib.lib.kge_kg_names(h, which, None, 0, offs.ctypes.data))
            buf = C.create_string_buffer(int(offs[-1]) + 1)
            _lib.check(_lib.lib.kge_kg_names(h, which, buf, int(offs[-1]), offs.ctypes.data))
            raw = buf.raw
            maps.append({raw[offs[i]:offs[i + 1]].decode("utf-8"): i for i in range(cnt)})
        return maps[0], maps[1], arrs
    finally:
        _lib.lib.kge_kg_close(h)


def generate_negative(kg, N, negative=2, check_kg=False, corrupt_head=True, corrupt_tail=True):
    """utils.py:114-121.  Consumes NumPy's *global* legacy stream exactly like the reference
    (all `negative*len(kg)` heads, then all tails) and returns the same (negative*len(kg), 3, 1) int64 array."""
    assert corrupt_head or corrupt_tail  # utils.py:116
    kg = np.asarray(kg)
    n = negative * len(kg)
    R = np.repeat(kg[:, 1].reshape((-1, 1)), negative, axis=0)
    st = NumpyStream().adopt_global()
    fs = st.randint(0, N, (n, 1))
    fo = st.randint(0, N, (n, 1))
    st.release_global()
    return np.stack([fs, R.astype(fs.dtype, copy=False), fo], axis=1)


def oversample_data(kgs, x=None, y=None, testing=False):
    """utils.py:123-147.  Every array is resampled with replacement to the common length with the draws
    `random.choices` would make on Python's *global* generator, in list order."""
    if testing:
        kgs = [np.asarray(kg)[:len(y)] for kg in kgs]
    else:
        kgs = [np.asarray(kg) for kg in kgs]
    if y is not None:
        m = max(max(map(len, kgs)), len(y))
    else:
        m = max(map(len, kgs))
    st = PythonStream().adopt_global()
    out = []
    for kg in kgs:
        if len(kg) == 0:
            raise IndexError("Cannot choose from an empty sequence")  # random.choices
        out.append(kg[st.choices_idx(len(kg), m)])
    st.release_global()
    if x is not None and y is not None:
        k = int(np.ceil(m / len(y)))
        y = np.repeat(y, k, axis=0)[:m]
        x = np.repeat(x, k, axis=0)[:m, :]
        for s in np.split(x, 3, axis=1):
            out.append(s.reshape((-1,)))
        return [np.squeeze(np.asarray(o)) for o in out], np.asarray(y)
    return [np.squeeze(np.asarray(o)) for o in out]


def pad(kg, bs):
    """utils.py:149-153: append random.choice(kg) until len % bs == 0 (one getrandbits draw loop per pick)."""
    import random
    kg = list(kg)
    while len(kg) % bs != 0:
        kg.append(random.choice(kg))
    return np.asarray(kg)


# ---- loss names (utils.py:288-330).  The arithmetic lives in the HIP train step; the table maps the reference's
# names onto it.  A looked-up entry is a small descriptor object that the models accept wherever the reference
# passes the function around, and that can also be called on two score vectors (device) to get the loss value.
class LossFunction:
    def __init__(self, name):
        self.name = name
        self.__name__ = name

    def __call__(self, true, false, margin=1, negative_samples=1, reduce_mean=True):
        from .models import loss_value, loss_pairs
        if not reduce_mean and self.name.startswith("pairwize"):  # the pointwize ones ignore the flag (utils.py:288-298)
            return loss_pairs(self.name, true, false, margin, negative_samples)
        return loss_value(self.name, true, false, margin, negative_samples)

    def __repr__(self):
        return f"<kge loss {self.name}>"


_LOSSES = ("pointwize_hinge", "pointwize_logistic", "pointwize_cross_entropy", "pointwize_square_loss",
           "pairwize_hinge", "pairwize_logistic", "pairwize_square_loss")
pointwize_hinge, pointwize_logistic, pointwize_cross_entropy, pointwize_square_loss, pairwize_hinge, \
    pairwize_logistic, pairwize_square_loss = (LossFunction(n) for n in _LOSSES)


def loss_function_lookup(name):
    """utils.py:321-330 (KeyError on unknown names, like the reference's dict lookup)."""
    return {n: globals()[n] for n in _LOSSES}[name]


# ---- filtered ranking (utils.py:165-264)
def gen_tail_data(test_data, num_entities, bs, filter_t):
    """utils.py:165-179: per test triple the (n_cand+1, 3) candidate list, target first, then every entity that
    is neither the target nor filtered, ascending.  ValueError (list.remove) when a filtered id is absent,
    e.g. when the test triple itself is in the filter."""
    for s, p, o in test_data:
        yield _candidates(int(s), int(p), int(o), num_entities, filter_t[(s, p)], tail=True)


def gen_head_data(test_data, num_entities, bs, filter_h):
    """utils.py:181-196."""
    for s, p, o in test_data:
        yield _candidates(int(s), int(p), int(o), num_entities, filter_h[(p, o)], tail=False)


def _candidates(s, p, o, num_entities, filtered, tail):
    target = o if tail else s
    keep = np.ones(num_entities, dtype=bool)
    if not 0 <= target < num_entities:
        raise ValueError("list.remove(x): x not in list")
    keep[target] = False
    for x in filtered:
        if not (0 <= x < num_entities) or not keep[x]:
            raise ValueError("list.remove(x): x not in list")
        keep[x] = False
    ids = np.concatenate(([target], np.nonzero(keep)[0])).astype(np.int64)
    out = np.empty((len(ids), 3), dtype=np.int64)
    out[:, 0] = s if tail else ids
    out[:, 1] = p
    out[:, 2] = ids if tail else o
    return out


def metrics_from_ranks(tail_ranks, head_ranks):
    """utils.py:216-262: hits@k <=> real_rank <= k; the 15 keys of the reference's dict."""
    def agg(r):
        r = np.asarray(r, dtype=np.float64)
        n = float(len(r))
        return (np.sum(r <= 1) / n, np.sum(r <= 3) / n, np.sum(r <= 10) / n, np.mean(r), np.mean(1.0 / r))
    t1, t3, t10, tmr, tmrr = agg(tail_ranks)
    h1, h3, h10, hmr, hmrr = agg(head_ranks)
    return {'tail_hits@1': t1, 'tail_hits@3': t3, 'tail_hits@10': t10, 'tail_mr': tmr, 'tail_mrr': tmrr,
            'head_hits@1': h1, 'head_hits@3': h3, 'head_hits@10': h10, 'head_mr': hmr, 'head_mrr': hmrr,
            'hits@1': (t1 + h1) / 2, 'hits@3': (t3 + h3) / 2, 'hits@10': (t10 + h10) / 2,
            'mr': (tmr + hmr) / 2, 'mrr': (tmrr + hmrr) / 2}


def filter_coo(test, side, filtering_triples, num_entities):
    """COO (query index, excluded id) of the reference's filter_t[(s,p)] / filter_h[(p,o)] sets for every test triple
    (side 0 = tails, 1 = heads), built by the C++ front-end (sort + binary search).  ValueError where the reference's
    list.remove would raise."""
    import ctypes as C
    test = np.ascontiguousarray(np.asarray(test, dtype=np.int32).reshape(-1, 3))
    filt = np.ascontiguousarray(np.asarray(filtering_triples, dtype=np.int32).reshape(-1, 3))
    n = C.c_int64(0)
    _lib.check(_lib.lib.kge_filter_coo_host(test.ctypes.data, len(test), side, filt.ctypes.data, len(filt), int(num_entities), None, None, C.byref(n)))
    q = np.empty(n.value, dtype=np.int32)
    ids = np.empty(n.value, dtype=np.int32)
    _lib.check(_lib.lib.kge_filter_coo_host(test.ctypes.data, len(test), side, filt.ctypes.data, len(filt), int(num_entities),
                                           q.ctypes.data, ids.ctypes.data, C.byref(n)))
    return q, ids


def validate(model, test_data, num_entities, bs, filtering_triples=None):
    """utils.py:199-264.  `bs` is unused, as in the reference; `filtering_triples=None` raises TypeError, as there.

    For the models of this package the 1-vs-all sweep runs on the GPU (`model.rank`); any other object with a
    `.predict` is driven exactly like the reference drives it (candidate list per query, target in row 0)."""
    if hasattr(model, "rank_triples"):  # GPU ranker + C++ filter index: no per-query Python work
        if filtering_triples is None:
            raise TypeError("'NoneType' object is not iterable")  # the reference iterates over it (utils.py:203)
        filt = np.asarray(filtering_triples if isinstance(filtering_triples, np.ndarray) else list(filtering_triples), dtype=np.int32).reshape(-1, 3)
        tail_ranks, head_ranks = model.rank_triples(np.asarray(test_data, dtype=np.int32).reshape(-1, 3), num_entities, filt)
        return metrics_from_ranks(tail_ranks, head_ranks)
    filter_h = defaultdict(set)
    filter_t = defaultdict(set)
    for s, p, o in filtering_triples:  # TypeError on None, like the reference
        filter_h[(p, o)].add(s)
        filter_t[(s, p)].add(o)
    test_data = [tuple(int(v) for v in t) for t in test_data]
    if hasattr(model, "rank"):
        tail_ranks, head_ranks = model.rank(test_data, num_entities, filter_t, filter_h)
    else:
        tail_ranks = [_rank_from_scores(model.predict(t)) for t in gen_tail_data(test_data, num_entities, bs, filter_t)]
        head_ranks = [_rank_from_scores(model.predict(t)) for t in gen_head_data(test_data, num_entities, bs, filter_h)]
    return metrics_from_ranks(tail_ranks, head_ranks)


def _rank_from_scores(res):
    """utils.py:211-215: rankdata(res,'max')[0] = #{j: res_j <= res_0}; real_rank = n - that + 1."""
    res = np.asarray(res).reshape((-1,))
    return len(res) - int(np.sum(res <= res[0])) + 1


class KGEValidateCallback:
    """Working equivalent of utils.py:267-285 (the reference's passes train_data in the `bs` slot and no filter,
    so it cannot run): every `period` epochs and at train end, `validate` with train_data as the filter; results
    go to logs['val_<key>']."""

    def __init__(self, validation_data, train_data=None, period=5):
        self.validation_data = validation_data
        self.train_data = train_data
        self.period = period
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        if epoch % self.period == 0:
            logs = logs if logs is not None else {}
            vs = set(tuple(int(v) for v in t) for t in self.validation_data)
            filt = [t for t in (tuple(int(v) for v in t) for t in (self.train_data or [])) if t not in vs]
            tmp = validate(self.model, self.validation_data, self.model.num_entities, None, filt)
            for k in tmp:
                logs['val_' + k] = tmp[k]

    def on_train_end(self, logs=None):
        self.on_epoch_end(100, logs=logs)
