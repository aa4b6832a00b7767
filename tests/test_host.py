"""CPU-only tests of the product's host side: the C ABI loads and exports what include/kge_hip.h declares, the
MT19937 samplers reproduce the reference-generated golden vectors bit for bit, and the batch builder /
evaluator glue mirrors the reference (no compute calls that need a GPU)."""
import ctypes
import hashlib
import os
import random
import re
import sys

import numpy as np
import pytest

import kge_keras_amd as K
from kge_keras_amd import _lib, utils as U
from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
from kge_keras_amd.harness import DataGenerator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ ABI
def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "kge_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(kge_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in kge_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.kge_abi_version() == int(re.search(r"#define KGE_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "kge_hip.h")).read()).group(1))


def test_abi_struct_layouts():
    assert ctypes.sizeof(_lib.ModelDesc) == 52 and _lib.ModelDesc.modulus.offset == 40 and _lib.ModelDesc.conv_kw.offset == 48
    assert ctypes.sizeof(_lib.LossDesc) == 16
    assert ctypes.sizeof(_lib.OptDesc) == 72
    assert _lib.OptDesc.step.offset == 24 and _lib.OptDesc.mE.offset == 32


def test_argument_errors_without_gpu():
    md = _lib.ModelDesc(0, 10, 3, 8, 4, 1, 0.0, 0.0, 0, 0.0)  # e_dim != r_dim
    rc = _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None)
    assert rc == _lib.KGE_ESHAPE and b"broadcast" in _lib.lib.kge_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    md = _lib.ModelDesc(9, 10, 3, 8, 8, 1, 0.0, 0.0, 0, 0.0)
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None) == _lib.KGE_EINVAL
    md = _lib.ModelDesc(2, 10, 3, 7, 7, 1, 0.0, 0.0, 0, 0.0)  # ComplEx odd width
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None) == _lib.KGE_ESHAPE
    md = _lib.ModelDesc(0, 10, 3, 8, 8, 1, 0.0, 0.0, 0, 0.0)
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 0, 1, 1, None) == 0  # empty input is a no-op
    assert _lib.lib.kge_score(ctypes.byref(md), None, 1, 1, 4, 1, 1, None) == _lib.KGE_EINVAL
    # HAKE (models.py:362-400): rows [phase | modulus] against [phase | modulus | bias]; entry points of its own
    hp = _lib.HakeParams(0.5, 1.0, 0.1)
    md = _lib.ModelDesc(_lib.MODEL_IDS["HAKE"], 10, 3, 8, 8, 1, 12.0, 0.0, 0, 0.0)  # r_dim must be 3 * e_dim / 2
    assert _lib.lib.kge_hake_score(ctypes.byref(md), ctypes.byref(hp), 1, 1, 1, 4, 0, None, 0, 1, 1, 1 << 20, 1, None) == _lib.KGE_ESHAPE
    md = _lib.ModelDesc(_lib.MODEL_IDS["HAKE"], 10, 3, 8, 12, 1, 12.0, 0.0, 0, 0.0)
    assert _lib.lib.kge_hake_score(ctypes.byref(md), ctypes.byref(hp), 1, 1, 1, 0, 0, None, 0, 1, 1, 1 << 20, 1, None) == 0  # empty: a no-op
    assert _lib.lib.kge_hake_score(ctypes.byref(md), None, 1, 1, 1, 4, 0, None, 0, 1, 1, 1 << 20, 1, None) == _lib.KGE_EINVAL
    assert _lib.lib.kge_score(ctypes.byref(md), 1, 1, 1, 4, 1, 1, None) == _lib.KGE_EUNSUPPORTED and b"kge_score" in _lib.lib.kge_last_error()
    assert _lib.lib.kge_rank_counts(ctypes.byref(md), 1, 1, None, 0, 10, 1, 4, 0, None, None, 0, 1, 0, None, 1, 1 << 20, 1, None) == _lib.KGE_EUNSUPPORTED
    md.model = 0
    assert _lib.lib.kge_hake_score(ctypes.byref(md), ctypes.byref(hp), 1, 1, 1, 4, 0, None, 0, 1, 1, 1 << 20, 1, None) in (_lib.KGE_EINVAL, _lib.KGE_ESHAPE)
    assert _lib.lib.kge_hake_workspace_bytes(1000, 0) >= _lib.lib.kge_hake_workspace_bytes(1000, 1000) > 0


# ------------------------------------------------------------------ samplers vs reference-generated vectors
def test_randint_golden(golden):
    g = golden("generate_negative")
    for c in g["cases"]:
        st = NumpyStream(c["seed"])
        n = c["B"] * c["ns"]
        fs = st.randint(0, c["N"], n)
        fo = st.randint(0, c["N"], n)
        neg = np.asarray(c["neg"])
        assert fs.tolist() == neg[:, 0].tolist() and fo.tolist() == neg[:, 2].tolist()
        assert st.randint(0, 2**31 - 1, 2).tolist() == c["after"]


def test_generate_negative_global_stream_golden(golden):
    g = golden("generate_negative")
    for c in g["cases"]:
        np.random.seed(c["seed"])
        neg = U.generate_negative(np.asarray(c["kg"]), c["N"], c["ns"])
        assert list(neg.shape) == c["shape"] and str(neg.dtype) == c["dtype"]
        assert neg.reshape(-1, 3).tolist() == c["neg"]
        # the global numpy stream stands where the reference leaves it
        assert np.random.randint(0, 2**31 - 1, size=2).tolist() == c["after"]
    for c in g["big"]:
        rs = np.random.RandomState(c["kg_seed"])
        kg = np.stack([rs.randint(0, c["N"], c["B"]), rs.randint(0, 237, c["B"]), rs.randint(0, c["N"], c["B"])], axis=1)
        np.random.seed(c["seed"])
        neg = U.generate_negative(kg, c["N"], c["ns"]).reshape(-1, 3)
        assert sha(neg.astype(np.int64)) == c["sha256_int64"]
        assert neg.sum(0).tolist() == c["colsum"]
        assert np.random.randint(0, 2**31 - 1, size=2).tolist() == c["after"]


def test_oversample_golden(golden):
    g = golden("oversample")
    for c in g["cases"]:
        random.seed(c["seed"])
        neg = np.asarray(c["neg"]).reshape(-1, 3, 1)
        pos_os, neg_os = U.oversample_data([np.asarray(c["kg"]), neg])
        assert pos_os.tolist() == c["pos_os"] and neg_os.tolist() == c["neg_os"]
        assert [random.getrandbits(32) for _ in range(2)] == c["after"]
        # explicit seeding path (no global state)
        st = PythonStream(c["seed"])
        ia = st.choices_idx(c["B"], c["B"] * c["ns"])
        assert np.asarray(c["kg"])[ia].tolist() == c["pos_os"]


def test_fused_batch_builder_big_golden(golden):
    for c in golden("oversample")["big"]:
        rs = np.random.RandomState(c["kg_seed"])
        kg = np.stack([rs.randint(0, c["N"], c["B"]), rs.randint(0, 237, c["B"]), rs.randint(0, c["N"], c["B"])], axis=1)
        pos, neg = build_batch(NumpyStream(c["seed"]), PythonStream(c["seed"]), kg, c["ns"], c["N"])
        assert pos.dtype == np.int32 and pos.shape == (c["B"] * c["ns"], 3)
        assert sha(pos.astype(np.int64)) == c["sha_pos"] and sha(neg.astype(np.int64)) == c["sha_neg"]
        assert pos[:6].tolist() == c["pos_head"] and neg[:6].tolist() == c["neg_head"]


def test_shuffle_golden(golden):
    for c in golden("shuffle")["cases"]:
        st = NumpyStream(c["seed"])
        idx = st.shuffle(np.arange(c["n"], dtype=np.int64))
        assert sha(idx) == c["sha256_int64"]
        if c["perm"] is not None:
            assert idx.tolist() == c["perm"]
        assert st.randint(0, 2**31 - 1, 2).tolist() == c["after"]


def test_streams_vs_live_numpy_and_python():
    """Beyond the committed vectors: random seeds / ranges against the installed NumPy and CPython."""
    rs = np.random.RandomState(7)
    for _ in range(20):
        seed = int(rs.randint(0, 2**32 - 1, dtype=np.int64))
        hi = int(rs.choice([1, 2, 3, 14, 255, 256, 257, 14541, 40943, 123182, 2**31 - 1, 2**32]))
        np.random.seed(seed)
        want = np.random.randint(0, hi, size=50)
        got = NumpyStream(seed).randint(0, hi, 50)
        assert got.tolist() == want.tolist(), (seed, hi)
        pseed = int(rs.randint(-2**40, 2**40, dtype=np.int64))
        random.seed(pseed)
        pop = int(rs.randint(1, 10**6))
        want = random.choices(range(pop), k=40)
        assert PythonStream(pseed).choices_idx(pop, 40).tolist() == want
    assert NumpyStream(5).raw(3).tolist() == [int(x) for x in np.random.RandomState(5).randint(0, 2**32, 3, dtype=np.uint64)]
    with pytest.raises(ValueError):
        NumpyStream(1).randint(5, 5, 3)  # numpy: low >= high
    with pytest.raises(ValueError):
        PythonStream(1).choices_idx(0, 3)  # random.choices on an empty population raises
    assert NumpyStream(1).randint(3, 4, 4).tolist() == [3, 3, 3, 3]  # range of one: no draw
    assert NumpyStream(1).randint(0, 10, 0).shape == (0,)


def test_adopt_and_release_global_state():
    np.random.seed(123)
    a = np.random.randint(0, 1000, 5)
    st = NumpyStream().adopt_global()
    b = st.randint(0, 1000, 5)
    st.release_global()
    c = np.random.randint(0, 1000, 5)
    np.random.seed(123)
    assert np.concatenate([a, b, c]).tolist() == np.random.randint(0, 1000, 15).tolist()
    random.seed(9)
    x = random.random()
    ps = PythonStream().adopt_global()
    i = ps.choices_idx(1000, 3)
    ps.release_global()
    y = random.random()
    random.seed(9)
    assert x == random.random() and random.choices(range(1000), k=3) == i.tolist() and y == random.random()


# ------------------------------------------------------------------ harness batch order (tests.py:29-61)
def test_datagenerator_matches_reference_stream(golden, datasets):
    g = golden("batch_stream")
    np.random.seed(g["np_seed"])
    random.seed(g["py_seed"])
    gen = DataGenerator(datasets["nations_train"], ns=g["ns"], batch_size=g["B"])
    assert gen.num_e == g["num_e"] and len(gen) == g["steps_per_epoch"]
    for b, ref in enumerate(g["batches"]):
        (pos, neg), y = gen[b]
        assert y is None
        assert pos.tolist() == ref["pos"] and neg.tolist() == ref["neg"]


# ------------------------------------------------------------------ evaluator glue with a foreign .predict object
class _DM:
    def __init__(self, N, R, d, seed, quant):
        rs = np.random.RandomState(seed)
        self.E = rs.uniform(-1, 1, (N, d))
        self.Rel = rs.uniform(-1, 1, (R, d))
        if quant:
            self.E = np.round(self.E * quant) / quant
            self.Rel = np.round(self.Rel * quant) / quant

    def predict(self, t):
        t = np.asarray(t)
        return np.sum(self.E[t[:, 0]] * self.Rel[t[:, 1]] * self.E[t[:, 2]], -1)


def _ids(a):
    return [tuple(int(x) for x in t) for t in a]


def test_validate_generic_predict_golden(golden, datasets):
    for c in golden("validate")["cases"]:
        if c["dataset"] != "nations":
            continue
        m = c["model"]
        fm = _DM(m["N"], m["R"], m["d"], m["seed"], m["quant"])
        tes = set(_ids(datasets["nations_test"]))
        filt = [t for t in _ids(datasets["nations_train"]) + _ids(datasets["nations_valid"]) if t not in tes]
        met = U.validate(fm, c["test"], m["N"], 32, filtering_triples=filt)
        assert list(met.keys()) == list(c["metrics"].keys())
        for k, v in c["metrics"].items():
            assert met[k] == pytest.approx(v, abs=1e-12), k
        ft = {}
        for s, p, o in filt:
            ft.setdefault((s, p), set()).add(o)
        from collections import defaultdict
        rows = list(U.gen_tail_data([tuple(t) for t in c["test"][:3]], m["N"], 32, defaultdict(set, ft)))
        assert [r.tolist() for r in rows] == c["tails"]


def test_validate_error_behaviour(golden):
    errs = golden("misc")["misc"]["errors"]
    fm = _DM(14, 55, 4, 5, None)
    with pytest.raises(ValueError):  # test triple inside the filter: list.remove twice
        U.validate(fm, [(0, 1, 2)], 14, 32, filtering_triples=[(0, 1, 2)])
    assert errs["validate_filter_contains_test"] == "ValueError"
    with pytest.raises(TypeError):
        U.validate(fm, [(0, 1, 2)], 14, 32)
    assert errs["validate_no_filter"] == "TypeError"
    with pytest.raises(KeyError):
        U.loss_function_lookup("nope")
    assert errs["loss_lookup_unknown"] == "KeyError"
    for n in golden("misc")["misc"]["loss_lookup_ok"]:
        assert U.loss_function_lookup(n).name == n


def test_load_kg_and_pad(golden, tmp_path):
    g = golden("misc")["misc"]
    p = tmp_path / "kg.txt"
    p.write_text("\n".join("\t".join(t) for t in g["load_kg_first"]) + "\n")
    assert U.load_kg(str(p)) == g["load_kg_first"]
    random.seed(g["pad_seed"])
    assert U.pad(np.asarray(g["pad_in"]), g["pad_bs"]).tolist() == g["pad_out"]
    assert U.EPSILON == g["EPSILON"]


def test_no_product_import_of_oracle():
    """The product path must never route through the oracle."""
    pk = os.path.join(ROOT, "kge-keras_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_filter_coo_frontend_matches_reference_sets(golden, datasets):
    """C++ filter index == the reference's filter_t / filter_h dict-of-sets, for every query of nations and a WN18RR slice."""
    from collections import defaultdict
    for ds, n_ent in (("nations", 14), ("wn18rr", 40943)):
        tes = set(_ids(datasets[ds + "_test"]))
        filt = [t for t in _ids(datasets[ds + "_train"]) + _ids(datasets[ds + "_valid"]) if t not in tes]
        filt = filt + filt[:50]  # duplicates must collapse (sets)
        test = _ids(datasets[ds + "_test"])[:300]
        fh, ft = defaultdict(set), defaultdict(set)
        for s, p, o in filt:
            fh[(p, o)].add(s)
            ft[(s, p)].add(o)
        for side in (0, 1):
            q, ids = U.filter_coo(test, side, filt, n_ent)
            got = defaultdict(set)
            for a, b in zip(q.tolist(), ids.tolist()):
                assert b not in got[a]
                got[a].add(b)
            for i, (s, p, o) in enumerate(test):
                want = ft.get((s, p), set()) if side == 0 else fh.get((p, o), set())
                assert got.get(i, set()) == want, (ds, side, i)
    with pytest.raises(ValueError):
        U.filter_coo([(0, 1, 2)], 0, [(0, 1, 2)], 14)      # evaluated triple inside the filter
    with pytest.raises(ValueError):
        U.filter_coo([(0, 1, 2)], 0, [(0, 1, 99)], 14)     # filtered id outside range(num_entities)
    with pytest.raises(ValueError):
        U.filter_coo([(0, 1, 14)], 0, [], 14)              # target outside range(num_entities)
    q, ids = U.filter_coo(np.zeros((0, 3), np.int32), 1, [(0, 1, 2)], 14)
    assert len(q) == 0


def test_mt19937_jump_ahead_polynomial():
    """GF(2) jump-ahead (csrc/mtjump.cpp): the state advanced by J words through x^J mod phi continues the stream exactly
    where stepping the generator J words does."""
    for seed, J in ((1, 1), (5, 624), (7, 1000), (11, 20592), (2024, 624 * 165), (99, 819200)):
        a = NumpyStream(seed)
        a.raw(624)                      # make the state block a regenerated one (every word in the recurrence regime)
        b = NumpyStream()
        b._buf[:] = a._buf
        _lib.check(_lib.lib.kge_mt_jump_host(b._buf.ctypes.data, J))
        b._buf[624] = 624               # jumped block = raw words [J, J+624): continue with the block after it
        got = b.raw(700)
        ref = NumpyStream()
        ref._buf[:] = a._buf
        ref._buf[624] = 0               # pos 0: the state block's own words are words 0..623 of the stream
        ref.raw(J + 624)
        assert got.tolist() == ref.raw(700).tolist(), (seed, J)
    poly = np.zeros(312, dtype=np.uint64)
    _lib.check(_lib.lib.kge_mt_jump_poly(5, poly.ctypes.data))
    assert int(poly[0]) == 32 and not poly[1:].any()   # x^5 mod phi = x^5


def test_index_kg_matches_golden_id_arrays(golden, datasets, tmp_path):
    """utils.index_kg (sorted-string ids) reproduces the id arrays the golden vectors were generated with."""
    g = golden("misc")
    rows = g["misc"]["load_kg_first"]
    p = tmp_path / "kg.txt"
    p.write_text("\n".join("\t".join(t) for t in rows) + "\n")
    kg = U.load_kg(str(p))
    emap, rmap, (arr,) = U.index_kg(kg)
    assert arr.dtype == np.int32 and arr.shape == (len(rows), 3)
    assert sorted(emap, key=emap.get) == sorted(emap) and sorted(rmap, key=rmap.get) == sorted(rmap)
    for (s, p_, o), (a, b, c) in zip(rows, arr.tolist()):
        assert emap[s] == a and rmap[p_] == b and emap[o] == c


def test_graft_entry_build_contract():
    """__graft_entry__.build() is the driver's "does it build" check: run it here (incremental make, import, ABI version)."""
    import __graft_entry__ as g
    g.build()


def test_native_loader_matches_load_kg_and_index_kg(golden, datasets, tmp_path):
    """kge_kg_open (csrc/frontend.cpp) == load_kg (utils.py:106-112) + the sorted-string id maps, on files with tabs, runs of blanks,
    CRLF and bare-CR line ends, non-ASCII names, Unicode whitespace and a missing final newline; the golden rows of the reference's
    own loader; and the reference script's errors (a short row, a missing file)."""
    from kge_keras_amd import utils as U
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_bytes("x\tr1\ty\n  z   r2 \t x  \r\nÅse r1 zebra \nété r3　x\rq r1 q".encode("utf-8"))
    b.write_bytes(b"y r2 Z\nZ r0 x\n")
    em, rm, (ta, tb) = U.load_indexed(str(a), str(b))
    pa, pb = U.load_kg(str(a)), U.load_kg(str(b))
    assert pa == [["x", "r1", "y"], ["z", "r2", "x"], ["Åse", "r1", "zebra"], ["été", "r3", "x"], ["q", "r1", "q"]]
    em2, rm2, (ta2, tb2) = U.index_kg(pa, pb)
    assert em == em2 and rm == rm2 and np.array_equal(ta, ta2) and np.array_equal(tb, tb2)
    assert list(em) == sorted(em) and list(rm) == ["r0", "r1", "r2", "r3"] and ta.dtype == np.int32
    # the reference loader's golden rows
    rows = golden("misc")["misc"]["load_kg_first"]
    p = tmp_path / "g.txt"
    p.write_text("\n".join("\t".join(t) for t in rows) + "\n")
    em3, rm3, (tg,) = U.load_indexed(str(p))
    e4, r4, (tg4,) = U.index_kg(rows)
    assert em3 == e4 and rm3 == r4 and np.array_equal(tg, tg4)
    bad = tmp_path / "bad.txt"
    bad.write_text("a r b\nc d\n")
    with pytest.raises(ValueError, match="line 2: not enough values to unpack"):
        U.load_indexed(str(bad))
    with pytest.raises(FileNotFoundError):
        U.load_indexed(str(tmp_path / "nope.txt"))


def test_host_objects_under_asan_ubsan():
    """SURVEY 5 (sanitizers): sampler.cpp, frontend.cpp and mtjump.cpp -- raw pointer arithmetic on caller buffers -- built with
    -fsanitize=address,undefined (`make -C kge-keras_amd/csrc asan`) and driven through every host entry point at its edges by
    tests/host_asan_driver.py in a child process with the sanitizer runtime preloaded (CPU only: the build container, never the GPU box).
    A sanitizer report aborts the child; the final line must be HOST-ASAN-OK."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++ here")
    csrc = os.path.join(ROOT, "kge-keras_amd", "csrc")
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not rt or not os.path.exists(rt):
        pytest.skip("no AddressSanitizer runtime here")
    b = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_asan_driver.py"), os.path.join(csrc, "libkge_host_asan.so")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("HOST-ASAN-OK"), (r.stdout[-2000:], r.stderr[-4000:])
