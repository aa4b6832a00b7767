"""Driver of the sanitizer build of the host objects (kge-keras_amd/csrc/libkge_host_asan.so: `make -C kge-keras_amd/csrc asan`;
sampler.cpp, frontend.cpp, mtjump.cpp under -fsanitize=address,undefined).  Run by tests/test_host.py in a child process with the
AddressSanitizer runtime preloaded -- pure ctypes + NumPy, no torch, no GPU, never on the GPU box.  Every entry point is exercised at its
edges (empty inputs, sizes that are not multiples of the 624-word block, ragged files, error returns) and checked against NumPy's /
CPython's own generators (the reference's dependencies: utils.py:118-119,136, tests.py:50-53) and the committed golden vectors.
Exit code 0 + a final "HOST-ASAN-OK" line: no sanitizer report, all checks passed."""
import ctypes as C
import hashlib
import json
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "kge-keras_amd", "csrc", "libkge_host_asan.so"))
lib.kge_last_error.restype = C.c_char_p
P, I64, I32, U32 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32
sig = {"kge_mt_seed_numpy": [P, U32], "kge_mt_seed_python": [P, P, C.c_int], "kge_mt_raw": [P, P, I64], "kge_mt_randint": [P, I64, I64, P, I64],
       "kge_mt_shuffle": [P, P, I64], "kge_mt_choices": [P, I64, P, I64], "kge_build_batch_host": [P, P, P, I64, I64, I64, P, P],
       "kge_filter_coo_host": [P, I64, I32, P, I64, I64, P, P, P], "kge_kg_open": [P, I32, P], "kge_kg_sizes": [P, P, P, P],
       "kge_kg_triples": [P, I32, P], "kge_kg_names": [P, I32, P, I64, P], "kge_kg_close": [P], "kge_mt_jump_poly": [I64, P], "kge_mt_jump_host": [P, I64]}
for k, v in sig.items():
    getattr(lib, k).argtypes = v
    getattr(lib, k).restype = C.c_int


def ok(rc):
    assert rc == 0, (rc, lib.kge_last_error())


def state():
    return np.zeros(626, dtype=np.uint32)  # KGE_MT_STATE_BYTES = 2504


def np_state(seed):
    st = state()
    ok(lib.kge_mt_seed_numpy(st.ctypes.data, seed))
    return st


def py_state(seed):
    st = state()
    key = np.array([seed & 0xFFFFFFFF] if seed < 2**32 else [seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    ok(lib.kge_mt_seed_python(st.ctypes.data, key.ctypes.data, len(key)))
    return st


sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731

# ---- NumPy twin: randint with masked rejection, shuffle, raw words; sizes around the 624-word block edge
for seed in (0, 1, 12345, 2**32 - 1):
    for N in (1, 2, 14, 14541, 40943, 123182, 2**31 - 1):
        for n in (0, 1, 623, 624, 625, 5000):
            st, rs = np_state(seed), np.random.RandomState(seed)
            out = np.empty(max(n, 1), dtype=np.int64)
            ok(lib.kge_mt_randint(st.ctypes.data, 0, N, out.ctypes.data, n))
            assert out[:n].tolist() == rs.randint(0, N, n).tolist(), (seed, N, n)
            out2 = np.empty(3, dtype=np.int64)   # the stream stands where NumPy's does
            ok(lib.kge_mt_randint(st.ctypes.data, 5, 1000, out2.ctypes.data, 3))
            assert out2.tolist() == rs.randint(5, 1000, 3).tolist()
    for n in (0, 1, 2, 1592, 10007):
        st, rs = np_state(seed), np.random.RandomState(seed)
        a, b = np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64)
        ok(lib.kge_mt_shuffle(st.ctypes.data, a.ctypes.data, n))
        rs.shuffle(b)
        assert a.tolist() == b.tolist(), (seed, n)
_st, _o = np_state(1), np.empty(1, np.int64)   # (arrays handed over by address must outlive the call: keep the references)
assert lib.kge_mt_randint(_st.ctypes.data, 5, 5, _o.ctypes.data, 1) != 0   # empty range: low >= high is an error
# ---- CPython twin: random.choices indices
for seed in (0, 7, 2**31, 2**40 + 3):
    for pop, k in ((1, 5), (32, 320), (2048, 4097), (204800, 1000)):
        st = py_state(seed)
        random.seed(seed)
        idx = np.empty(k, dtype=np.int64)
        ok(lib.kge_mt_choices(st.ctypes.data, pop, idx.ctypes.data, k))
        assert idx.tolist() == random.choices(range(pop), k=k), (seed, pop, k)
# ---- fused batch builder against the reference-generated golden hashes (tests/golden/oversample.json) and tiny shapes
g = json.load(open(os.path.join(ROOT, "tests", "golden", "oversample.json")))
for c in g["big"]:
    rs = np.random.RandomState(c["kg_seed"])
    kg = np.stack([rs.randint(0, c["N"], c["B"]), rs.randint(0, 237, c["B"]), rs.randint(0, c["N"], c["B"])], axis=1).astype(np.int32)
    m = c["B"] * c["ns"]
    pos, neg = np.empty((m, 3), np.int32), np.empty((m, 3), np.int32)
    sn, sp_, kgc = np_state(c["seed"]), py_state(c["seed"]), np.ascontiguousarray(kg)
    ok(lib.kge_build_batch_host(sn.ctypes.data, sp_.ctypes.data, kgc.ctypes.data, c["B"], c["ns"], c["N"], pos.ctypes.data, neg.ctypes.data))
    assert sha(pos.astype(np.int64)) == c["sha_pos"] and sha(neg.astype(np.int64)) == c["sha_neg"]
for B, ns in ((1, 1), (3, 7), (32, 10)):
    kg = np.stack([np.arange(B), np.arange(B) % 3, (np.arange(B) * 5) % 14], 1).astype(np.int32)
    pos, neg = np.empty((B * ns, 3), np.int32), np.empty((B * ns, 3), np.int32)
    sn, sp_ = np_state(3), py_state(3)
    ok(lib.kge_build_batch_host(sn.ctypes.data, sp_.ctypes.data, kg.ctypes.data, B, ns, 14, pos.ctypes.data, neg.ctypes.data))
    assert pos.min() >= 0 and neg[:, [0, 2]].max() < 14
# ---- filter index: two-pass protocol, duplicates, error returns
test = np.array([[0, 1, 2], [3, 0, 4], [0, 1, 5]], dtype=np.int32)
filt = np.array([[0, 1, 7], [0, 1, 7], [9, 0, 4], [0, 1, 5 + 3], [3, 0, 1]], dtype=np.int32)
for side in (0, 1):
    n_out = C.c_int64(0)
    ok(lib.kge_filter_coo_host(test.ctypes.data, 3, side, filt.ctypes.data, len(filt), 14, None, None, C.byref(n_out)))
    q, ids = np.empty(max(1, n_out.value), np.int32), np.empty(max(1, n_out.value), np.int32)
    ok(lib.kge_filter_coo_host(test.ctypes.data, 3, side, filt.ctypes.data, len(filt), 14, q.ctypes.data, ids.ctypes.data, C.byref(n_out)))
    want = {}
    for s, p, o in filt.tolist():
        for i, (ts, tp, to) in enumerate(test.tolist()):
            if side == 0 and (s, p) == (ts, tp):
                want.setdefault(i, set()).add(o)
            if side == 1 and (p, o) == (tp, to):
                want.setdefault(i, set()).add(s)
    got = {}
    for a, b in zip(q[:n_out.value].tolist(), ids[:n_out.value].tolist()):
        got.setdefault(a, set()).add(b)
    assert got == want, (side, got, want)
n_out = C.c_int64(0)
assert lib.kge_filter_coo_host(test.ctypes.data, 3, 0, test.ctypes.data, 3, 14, None, None, C.byref(n_out)) != 0      # evaluated triple inside its filter
_bad = np.array([[0, 1, 99]], np.int32)
assert lib.kge_filter_coo_host(test.ctypes.data, 3, 0, _bad.ctypes.data, 1, 14, None, None, C.byref(n_out)) != 0
ok(lib.kge_filter_coo_host(test.ctypes.data, 0, 0, filt.ctypes.data, len(filt), 14, None, None, C.byref(n_out)))
assert n_out.value == 0
# ---- dataset front-end: ragged whitespace, CRLF / bare CR, non-ASCII names, no final newline; a short row; a missing file
with tempfile.TemporaryDirectory() as td:
    a, b = os.path.join(td, "a.txt"), os.path.join(td, "b.txt")
    open(a, "wb").write("x\tr1\ty\n  z   r2 \t x  \r\nÅse r1 zebra \nété r3　x\rq r1 q".encode("utf-8"))
    open(b, "wb").write(b"y r2 Z\nZ r0 x\n")
    paths = (C.c_char_p * 2)(a.encode(), b.encode())
    h = C.c_void_p()
    ok(lib.kge_kg_open(paths, 2, C.byref(h)))
    ne, nr, rows = C.c_int64(0), C.c_int64(0), (C.c_int64 * 2)()
    ok(lib.kge_kg_sizes(h, C.byref(ne), C.byref(nr), rows))
    assert (ne.value, nr.value, list(rows)) == (8, 4, [5, 2]), (ne.value, nr.value, list(rows))
    for sp in (0, 1):
        t = np.empty((rows[sp], 3), np.int32)
        ok(lib.kge_kg_triples(h, sp, t.ctypes.data))
        assert t.min() >= 0 and t[:, [0, 2]].max() < ne.value and t[:, 1].max() < nr.value
    for which, n in ((0, ne.value), (1, nr.value)):
        off = (C.c_int64 * (n + 1))()
        ok(lib.kge_kg_names(h, which, None, 0, off))
        buf = C.create_string_buffer(off[n])
        ok(lib.kge_kg_names(h, which, buf, off[n], off))
        names = [buf.raw[off[i]:off[i + 1]].decode("utf-8") for i in range(n)]
        assert names == sorted(names, key=lambda s: s.encode("utf-8")), names
    ok(lib.kge_kg_close(h))
    bad = os.path.join(td, "bad.txt")
    open(bad, "w").write("a r b\nc d\n")
    h2 = C.c_void_p()
    assert lib.kge_kg_open((C.c_char_p * 1)(bad.encode()), 1, C.byref(h2)) != 0 and b"line 2" in lib.kge_last_error()
    assert lib.kge_kg_open((C.c_char_p * 1)(os.path.join(td, "nope.txt").encode()), 1, C.byref(h2)) != 0
# ---- GF(2) jump-ahead against stepping the generator
for seed, J in ((1, 1), (5, 624), (7, 1000), (11, 20592), (99, 819200)):
    a = np_state(seed)
    tmp = np.empty(624, np.uint32)
    ok(lib.kge_mt_raw(a.ctypes.data, tmp.ctypes.data, 624))
    b, ref = a.copy(), a.copy()
    ok(lib.kge_mt_jump_host(b.ctypes.data, J))
    b[624] = 624
    got = np.empty(700, np.uint32)
    ok(lib.kge_mt_raw(b.ctypes.data, got.ctypes.data, 700))
    ref[624] = 0
    skip = np.empty(J + 624, np.uint32)
    ok(lib.kge_mt_raw(ref.ctypes.data, skip.ctypes.data, J + 624))
    want = np.empty(700, np.uint32)
    ok(lib.kge_mt_raw(ref.ctypes.data, want.ctypes.data, 700))
    assert got.tolist() == want.tolist(), (seed, J)
poly = np.zeros(312, dtype=np.uint64)
ok(lib.kge_mt_jump_poly(5, poly.ctypes.data))
assert int(poly[0]) == 32 and not poly[1:].any()
print("HOST-ASAN-OK")
