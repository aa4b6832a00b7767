#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the *reference itself*.

Runs ONLY in the build container (needs /root/reference).  Nothing under /root/reference is
copied: the reference's `KGEkeras/utils.py` is loaded by file path with *inert* placeholder
modules registered for the packages it imports at module scope but that are not installed here
(tensorflow, rdflib, spacy).  The placeholders compute nothing; every value written below comes out of
the reference's own NumPy / SciPy / `random` code:

  * `generate_negative`  (utils.py:114-121)   -> golden/generate_negative.json
  * `oversample_data`    (utils.py:123-147)   -> golden/oversample.json
  * `np.random.shuffle` as used by the harness (tests.py:50-53) -> golden/shuffle.json
  * the harness batch order `DataGenerator.__get_data` (tests.py:55-61), restated here as the two
    reference calls in the same order                         -> golden/batch_stream.json
  * `gen_tail_data` / `gen_head_data` / `validate` (utils.py:165-264) driven by a deterministic
    NumPy `.predict` object                                   -> golden/validate.json
  * `load_kg` (utils.py:106-112), `pad` (utils.py:149-153)     -> golden/misc.json

The arithmetic of the scoring / loss / optimizer path lives in TensorFlow, which is absent, so no
vector for it can be produced here ("parity unpinned" at that boundary).

Usage:  python tests/golden/make_golden.py
"""
import hashlib
import importlib.util
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference_utils():
    def inert(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Inert:  # placeholder base / symbol; never called for a value
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Inert()

    tf = inert("tensorflow")
    keras = inert("tensorflow.keras")
    tf.keras = keras
    keras.callbacks = inert("tensorflow.keras.callbacks", Callback=_Inert)
    keras.losses = inert("tensorflow.keras.losses", binary_crossentropy=_Inert())
    rdflib = inert("rdflib", Graph=_Inert, URIRef=_Inert, Literal=_Inert, Namespace=_Inert)
    rdflib.namespace = inert("rdflib.namespace", XSD=_Inert(), RDF=_Inert())
    inert("spacy")
    spec = importlib.util.spec_from_file_location("_ref_utils", os.path.join(REF, "KGEkeras", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def sorted_id_maps(*splits):
    """Deterministic ids: sorted strings (the harness's set-enumeration order is hash-seed dependent)."""
    ents = sorted({x for sp in splits for (s, _, o) in sp for x in (s, o)})
    rels = sorted({p for sp in splits for (_, p, _) in sp})
    return {e: i for i, e in enumerate(ents)}, {r: i for i, r in enumerate(rels)}


class DistMultPredict:
    """Deterministic NumPy stand-in for `model.predict` (DistMult with fixed tables)."""

    def __init__(self, N, R, d, seed, quant=None):
        rs = np.random.RandomState(seed)
        self.E = rs.uniform(-1, 1, (N, d))
        self.Rel = rs.uniform(-1, 1, (R, d))
        if quant:  # coarse grid -> many exact score ties
            self.E = np.round(self.E * quant) / quant
            self.Rel = np.round(self.Rel * quant) / quant

    def predict(self, t):
        t = np.asarray(t)
        return np.sum(self.E[t[:, 0]] * self.Rel[t[:, 1]] * self.E[t[:, 2]], axis=-1)


def main():
    U = _load_reference_utils()
    out = {}

    # ---------------------------------------------------------------- generate_negative
    cases = []
    for N in (14, 14541, 40943, 123182):
        for seed in (0, 1, 12345, 2**31 + 7):
            for B, ns in ((3, 2), (32, 10)):
                kg = np.stack([np.arange(B) % N, (np.arange(B) * 7) % 11, (np.arange(B) * 3 + 1) % N], axis=1)
                np.random.seed(seed)
                neg = U.generate_negative(kg, N=N, negative=ns)
                assert neg.shape == (B * ns, 3, 1)
                # two further raw draws pin where the stream stands after the call
                after = np.random.randint(0, 2**31 - 1, size=2)
                cases.append(dict(N=N, seed=seed, B=B, ns=ns, kg=kg.tolist(), shape=list(neg.shape),
                                  dtype=str(neg.dtype), neg=neg.reshape(-1, 3).tolist(), after=after.tolist()))
    # the BASELINE size, by checksum only
    big = []
    for N, seed in ((14541, 0), (14541, 99), (123182, 0), (40943, 5)):
        B, ns = 2048, 100
        rs = np.random.RandomState(4242)
        kg = np.stack([rs.randint(0, N, B), rs.randint(0, 237, B), rs.randint(0, N, B)], axis=1)
        np.random.seed(seed)
        neg = U.generate_negative(kg, N=N, negative=ns).reshape(-1, 3)
        after = np.random.randint(0, 2**31 - 1, size=2)
        big.append(dict(N=N, seed=seed, B=B, ns=ns, kg_seed=4242, sha256_int64=sha(neg.astype(np.int64)),
                        head=neg[:8].tolist(), tail=neg[-8:].tolist(), after=after.tolist(),
                        colsum=[int(x) for x in neg.sum(0)]))
    json.dump(dict(cases=cases, big=big), open(os.path.join(HERE, "generate_negative.json"), "w"))

    # ---------------------------------------------------------------- oversample_data
    cases = []
    for seed in (0, 7, 2**40 + 3, -5):
        for B, ns, N in ((3, 2, 14), (32, 10, 14541), (5, 7, 40943)):
            kg = np.stack([np.arange(B) % N, (np.arange(B) * 7) % 11, (np.arange(B) * 3 + 1) % N], axis=1)
            np.random.seed(3)
            neg = U.generate_negative(kg, N=N, negative=ns)
            random.seed(seed)
            pos_os, neg_os = U.oversample_data(kgs=[kg, neg])
            after = [random.getrandbits(32) for _ in range(2)]
            cases.append(dict(seed=seed, B=B, ns=ns, N=N, kg=kg.tolist(), neg=neg.reshape(-1, 3).tolist(),
                              pos_os=np.asarray(pos_os).tolist(), neg_os=np.asarray(neg_os).tolist(), after=after))
    big = []
    for seed in (0, 31337):
        B, ns, N = 2048, 100, 14541
        rs = np.random.RandomState(4242)
        kg = np.stack([rs.randint(0, N, B), rs.randint(0, 237, B), rs.randint(0, N, B)], axis=1)
        np.random.seed(seed)
        neg = U.generate_negative(kg, N=N, negative=ns)
        random.seed(seed)
        pos_os, neg_os = U.oversample_data(kgs=[kg, neg])
        after = [random.getrandbits(32) for _ in range(2)]
        big.append(dict(seed=seed, B=B, ns=ns, N=N, kg_seed=4242,
                        sha_pos=sha(np.asarray(pos_os).astype(np.int64)), sha_neg=sha(np.asarray(neg_os).astype(np.int64)),
                        pos_head=np.asarray(pos_os)[:6].tolist(), neg_head=np.asarray(neg_os)[:6].tolist(), after=after))
    json.dump(dict(cases=cases, big=big), open(os.path.join(HERE, "oversample.json"), "w"))

    # ---------------------------------------------------------------- shuffle (tests.py:50-53)
    cases = []
    for seed in (0, 1, 2024):
        for n in (1, 2, 14, 199, 1592, 4097):
            np.random.seed(seed)
            idx = np.arange(n)
            np.random.shuffle(idx)
            after = np.random.randint(0, 2**31 - 1, size=2)
            cases.append(dict(seed=seed, n=n, perm=idx.tolist() if n <= 1592 else None, sha256_int64=sha(idx.astype(np.int64)),
                              after=after.tolist()))
    np.random.seed(0)
    idx = np.arange(272115)
    np.random.shuffle(idx)
    cases.append(dict(seed=0, n=272115, perm=None, sha256_int64=sha(idx.astype(np.int64)),
                      after=np.random.randint(0, 2**31 - 1, size=2).tolist(), head=idx[:8].tolist()))
    json.dump(dict(cases=cases), open(os.path.join(HERE, "shuffle.json"), "w"))

    # ---------------------------------------------------------------- harness batch stream on nations
    train = U.load_kg(os.path.join(REF, "data/nations/train.txt"))
    valid = U.load_kg(os.path.join(REF, "data/nations/valid.txt"))
    test = U.load_kg(os.path.join(REF, "data/nations/test.txt"))
    emap, rmap = sorted_id_maps(train, valid, test)
    tr = [(emap[s], rmap[p], emap[o]) for s, p, o in train]
    va = [(emap[s], rmap[p], emap[o]) for s, p, o in valid]
    te = [(emap[s], rmap[p], emap[o]) for s, p, o in test]
    out["nations"] = dict(n_train=len(tr), n_valid=len(va), n_test=len(te), N=len(emap), R=len(rmap),
                          first_train=tr[:5], sha_train=sha(np.asarray(tr, dtype=np.int64)))
    B, ns = 32, 10
    np.random.seed(11)
    random.seed(11)
    num_e = len(set(s for s, _, _ in tr) | set(o for _, _, o in tr))  # tests.py:34
    index = np.arange(len(tr))
    np.random.shuffle(index)  # tests.py:50-53
    batches = []
    for b in range(3):
        sl = index[b * B:(b + 1) * B]  # tests.py:44
        tmp_kg = np.asarray([tr[i] for i in sl])  # tests.py:56
        neg = U.generate_negative(tmp_kg, N=num_e, negative=ns)  # tests.py:58
        X = U.oversample_data(kgs=[tmp_kg, neg])  # tests.py:59
        batches.append(dict(index=sl.tolist(), pos=np.asarray(X[0]).tolist(), neg=np.asarray(X[1]).tolist()))
    json.dump(dict(np_seed=11, py_seed=11, B=B, ns=ns, num_e=num_e, steps_per_epoch=len(tr) // B, batches=batches),
              open(os.path.join(HERE, "batch_stream.json"), "w"))

    # ---------------------------------------------------------------- validate
    vals = []
    N, R = len(emap), len(rmap)
    for quant, seed in ((None, 5), (2, 6)):  # second model: coarse grid => score ties
        fm = DistMultPredict(N, R, 4, seed, quant)
        # filters must not contain the evaluated triples (utils.py:168-171 double-remove -> ValueError)
        filt = [t for t in tr + va if t not in set(te)]
        tset = te[:40]
        m = U.validate(fm, tset, N, 32, filtering_triples=filt)
        # per-query candidate sets, to pin gen_tail_data / gen_head_data
        ft, fh = {}, {}
        for s, p, o in filt:
            fh.setdefault((p, o), set()).add(s)
            ft.setdefault((s, p), set()).add(o)
        from collections import defaultdict
        dft, dfh = defaultdict(set, ft), defaultdict(set, fh)
        tails = [t.tolist() for t in U.gen_tail_data(tset[:3], N, 32, dft)]
        heads = [t.tolist() for t in U.gen_head_data(tset[:3], N, 32, dfh)]
        vals.append(dict(dataset="nations", model=dict(kind="distmult_numpy", N=N, R=R, d=4, seed=seed, quant=quant),
                         test=tset, filt_sha=sha(np.asarray(filt, dtype=np.int64)), n_filt=len(filt),
                         metrics={k: float(v) for k, v in m.items()}, tails=tails, heads=heads))
    # WN18RR slice
    wtrain = U.load_kg(os.path.join(REF, "data/WN18RR/train.txt"))
    wvalid = U.load_kg(os.path.join(REF, "data/WN18RR/valid.txt"))
    wtest = U.load_kg(os.path.join(REF, "data/WN18RR/test.txt"))
    wemap, wrmap = sorted_id_maps(wtrain, wvalid, wtest)
    wtr = [(wemap[s], wrmap[p], wemap[o]) for s, p, o in wtrain]
    wva = [(wemap[s], wrmap[p], wemap[o]) for s, p, o in wvalid]
    wte = [(wemap[s], wrmap[p], wemap[o]) for s, p, o in wtest]
    out["wn18rr"] = dict(n_train=len(wtr), n_valid=len(wva), n_test=len(wte), N=len(wemap), R=len(wrmap),
                         first_train=wtr[:5], sha_train=sha(np.asarray(wtr, dtype=np.int64)))
    fm = DistMultPredict(len(wemap), len(wrmap), 8, 9)
    tes = set(wte)
    filt = [t for t in wtr + wva if t not in tes]
    tset = wte[:12]
    m = U.validate(fm, tset, len(wemap), 32, filtering_triples=filt)
    vals.append(dict(dataset="wn18rr", model=dict(kind="distmult_numpy", N=len(wemap), R=len(wrmap), d=8, seed=9, quant=None),
                     test=tset, filt_sha=sha(np.asarray(filt, dtype=np.int64)), n_filt=len(filt),
                     metrics={k: float(v) for k, v in m.items()}))
    json.dump(dict(cases=vals), open(os.path.join(HERE, "validate.json"), "w"))

    # ---------------------------------------------------------------- misc: load_kg, pad, error behaviour
    random.seed(3)
    padded = U.pad(np.asarray(tr[:10]), 4)
    errs = {}
    try:
        U.validate(DistMultPredict(N, R, 4, 5), te[:2], N, 32, filtering_triples=te[:2])
    except Exception as e:  # noqa
        errs["validate_filter_contains_test"] = type(e).__name__
    try:
        U.validate(DistMultPredict(N, R, 4, 5), te[:2], N, 32)
    except Exception as e:  # noqa
        errs["validate_no_filter"] = type(e).__name__
    try:
        U.loss_function_lookup("nope")
    except Exception as e:  # noqa
        errs["loss_lookup_unknown"] = type(e).__name__
    out["misc"] = dict(load_kg_first=train[:3], pad_seed=3, pad_in=[list(map(int, t)) for t in tr[:10]], pad_bs=4,
                       pad_out=padded.tolist(), errors=errs,
                       loss_names=sorted(["pointwize_hinge", "pointwize_logistic", "pointwize_cross_entropy",
                                          "pointwize_square_loss", "pairwize_hinge", "pairwize_logistic",
                                          "pairwize_square_loss"]),
                       loss_lookup_ok=[n for n in ["pointwize_hinge", "pointwize_logistic", "pointwize_cross_entropy",
                                                   "pointwize_square_loss", "pairwize_hinge", "pairwize_logistic",
                                                   "pairwize_square_loss"] if callable(U.loss_function_lookup(n))],
                       EPSILON=U.EPSILON)
    json.dump(out, open(os.path.join(HERE, "misc.json"), "w"))

    # ---------------------------------------------------------------- datasets as int32 id arrays (sorted-string id maps)
    ds = dict(nations_train=tr, nations_valid=va, nations_test=te, wn18rr_train=wtr, wn18rr_valid=wva, wn18rr_test=wte)
    for name, d in (("fb15k237", "FB15k-237"), ("yago310", "YAGO3-10")):  # train splits are missing blobs
        v = U.load_kg(os.path.join(REF, "data", d, "valid.txt"))
        t = U.load_kg(os.path.join(REF, "data", d, "test.txt"))
        em, rm = sorted_id_maps(v, t)
        ds[name + "_valid"] = [(em[s], rm[p], em[o]) for s, p, o in v]
        ds[name + "_test"] = [(em[s], rm[p], em[o]) for s, p, o in t]
        ds[name + "_dims"] = [len(em), len(rm)]
    np.savez_compressed(os.path.join(HERE, "datasets.npz"), **{k: np.asarray(v, dtype=np.int32) for k, v in ds.items()})
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".json"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
