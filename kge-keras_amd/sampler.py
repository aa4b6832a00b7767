"""Host-side MT19937 streams that reproduce NumPy's legacy global RandomState and CPython's `random`
bit for bit (csrc/sampler.cpp).  A stream can adopt the live global state and hand it back, so that
`generate_negative` / `oversample_data` here consume exactly the draws the reference's calls would
(KGEkeras/utils.py:118-119,136; tests.py:50-53) and leave the global generators where the reference
would have left them."""
import ctypes as C
import random as _pyrandom

import numpy as np

from ._lib import lib, check, MT_STATE_BYTES


class _Stream:
    def __init__(self):
        self._buf = np.zeros(MT_STATE_BYTES // 4, dtype=np.uint32)  # 624 words + pos + pad

    @property
    def ptr(self):
        return self._buf.ctypes.data

    def raw(self, n):
        out = np.empty(int(n), dtype=np.uint32)
        check(lib.kge_mt_raw(self.ptr, out.ctypes.data, int(n)))
        return out


class NumpyStream(_Stream):
    """Twin of `np.random` (legacy RandomState, MT19937 seeded by init_genrand)."""

    def __init__(self, seed=None):
        super().__init__()
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        seed = int(seed)
        if not 0 <= seed <= 0xFFFFFFFF:
            raise ValueError("Seed must be between 0 and 2**32 - 1")  # numpy's message
        check(lib.kge_mt_seed_numpy(self.ptr, C.c_uint32(seed)))
        return self

    def adopt_global(self):
        """Copy the state of NumPy's global generator."""
        kind, key, pos, _, _ = np.random.get_state()
        assert kind == "MT19937"
        self._buf[:624] = key
        self._buf[624] = pos
        return self

    def release_global(self):
        """Write this stream's state back into NumPy's global generator."""
        np.random.set_state(("MT19937", self._buf[:624].copy(), int(self._buf[624]), 0, 0.0))

    def randint(self, low, high, size):
        n = int(np.prod(size))
        out = np.empty(n, dtype=np.int64)
        check(lib.kge_mt_randint(self.ptr, int(low), int(high), out.ctypes.data, n))
        return out.reshape(size)

    def shuffle(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.int64)
        check(lib.kge_mt_shuffle(self.ptr, a.ctypes.data, a.shape[0]))
        if a is not arr:
            arr[...] = a
        return arr


class PythonStream(_Stream):
    """Twin of the module-level `random` generator (MT19937 seeded by init_by_array)."""

    def __init__(self, seed=None):
        super().__init__()
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        a = abs(int(seed))  # random.seed(int) hashes nothing: key = 32-bit words of |seed|, little-endian
        words = []
        while True:
            words.append(a & 0xFFFFFFFF)
            a >>= 32
            if not a:
                break
        key = np.asarray(words, dtype=np.uint32)
        check(lib.kge_mt_seed_python(self.ptr, key.ctypes.data, len(words)))
        return self

    def adopt_global(self):
        ver, st, _ = _pyrandom.getstate()
        assert ver == 3
        self._buf[:625] = np.asarray(st, dtype=np.uint32)
        return self

    def release_global(self):
        _pyrandom.setstate((3, tuple(int(x) for x in self._buf[:625]), None))

    def choices_idx(self, population, k):
        out = np.empty(int(k), dtype=np.int64)
        check(lib.kge_mt_choices(self.ptr, int(population), out.ctypes.data, int(k)))
        return out


def build_batch(np_stream, py_stream, batch, ns, num_e):
    """generate_negative + oversample_data of one (B,3) batch in one host call (tests.py:55-61).
    Returns int32 (m,3) pos and neg, m = B*ns."""
    b = np.ascontiguousarray(batch, dtype=np.int32).reshape(-1, 3)
    m = b.shape[0] * int(ns)
    pos = np.empty((m, 3), dtype=np.int32)
    neg = np.empty((m, 3), dtype=np.int32)
    check(lib.kge_build_batch_host(np_stream.ptr, py_stream.ptr, b.ctypes.data, b.shape[0], int(ns), int(num_e),
                                   pos.ctypes.data, neg.ctypes.data))
    return pos, neg


class DeviceSampler:
    """The two generator states resident in HBM; build_batch() is kge_build_batch_device: the same draws as the host
    builder (and as the reference), produced by HIP kernels, with pos/neg left on the device."""

    def __init__(self, np_stream, py_stream, device=None):
        import torch
        from . import _lib
        _lib.require_gpu()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.np_state = torch.from_numpy(np_stream._buf.view(np.uint8).copy()).to(self.device)
        self.py_state = torch.from_numpy(py_stream._buf.view(np.uint8).copy()).to(self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ws = None
        self._plans = {}  # (B, ns, num_e) -> device tensor of jump polynomials (segmented generation)
        self.segmented = True

    def to_streams(self):
        """Copy the device states back into host twins (e.g. for the epoch shuffle, which is sequential by nature)."""
        nps, pys = NumpyStream(), PythonStream()
        nps._buf[:] = self.np_state.cpu().numpy().view(np.uint32)
        pys._buf[:] = self.py_state.cpu().numpy().view(np.uint32)
        return nps, pys

    def load_streams(self, np_stream=None, py_stream=None):
        import torch
        if np_stream is not None:
            self.np_state.copy_(torch.from_numpy(np_stream._buf.view(np.uint8).copy()))
        if py_stream is not None:
            self.py_state.copy_(torch.from_numpy(py_stream._buf.view(np.uint8).copy()))

    def _workspace(self, B, ns, num_e):
        """The library's scratch: zero-filled when allocated and again whenever the batch shape changes (the layout depends on
        it, and the batch epoch that tags in-flight tile counts must not be read from stale bytes)."""
        import torch
        need = check(lib.kge_sampler_workspace_bytes(B, ns, num_e))
        shape = (B, ns, num_e)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        elif getattr(self, "_ws_shape", shape) != shape:
            self._ws.zero_()
        self._ws_shape = shape

    def stream_buffer(self, B, ns, num_e):
        """The device buffer of the streamed form (kge_sampler_stream_*): the two generators' word rings + the draw's scratch,
        zero-filled when allocated; one per batch shape."""
        import torch
        key = (int(B), int(ns), int(num_e))
        if getattr(self, "_stream_key", None) != key:
            need = check(lib.kge_sampler_stream_bytes(*key))
            self._stream = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
            self._stream_key = key
        return self._stream

    # ---- streamed form: word generation ahead of the batches (kge_sampler_stream_*; what pipeline.TrainPipeline drives natively)
    def stream_init(self, B, ns, num_e):
        """Seed the word rings from the current generator states."""
        from . import _lib
        buf = self.stream_buffer(B, ns, num_e)
        check(lib.kge_sampler_stream_init(buf.data_ptr(), buf.numel(), self.np_state.data_ptr(), self.py_state.data_ptr(), int(B), int(ns), int(num_e),
                                          _lib.stream_ptr()))

    def stream_plan(self, B, ns, num_e):
        """Jump polynomials of the streamed form's production rounds for this batch shape, on the device (None: sequential rounds)."""
        import torch
        if not self.segmented:
            return None
        key = ("stream", int(B), int(ns), int(num_e))
        if key not in self._plans:
            buf = np.empty(int(lib.kge_sampler_plan_bytes()) // 8, dtype=np.uint64)
            check(lib.kge_sampler_stream_plan_host(int(B), int(ns), int(num_e), buf.ctypes.data))
            self._plans[key] = torch.from_numpy(buf.view(np.int64)).to(self.device)
        return self._plans[key]

    def stream_produce(self, B, ns, num_e):
        """One production round on the current stream (kge_sampler_stream_round() batches' worth of words; skipped on the device
        while enough words lie ahead)."""
        from . import _lib
        buf = self.stream_buffer(B, ns, num_e)
        plan = self.stream_plan(B, ns, num_e)
        check(lib.kge_sampler_stream_produce(buf.data_ptr(), buf.numel(), int(B), int(ns), int(num_e), plan.data_ptr() if plan is not None else None,
                                             _lib.stream_ptr()))

    def stream_draw(self, batch, ns, num_e, pos_out=None, neg_out=None, indexed=None, counted=None):
        """build_batch() from the rings: same arguments, same outputs, same generator states left behind."""
        import torch
        from . import _lib
        B = batch.shape[0] if indexed is None else int(indexed[4])
        m = B * int(ns)
        buf = self.stream_buffer(B, ns, num_e)
        arrays = not (counted is not None and pos_out is False)
        if not arrays:
            pos_out = neg_out = None
        elif pos_out is None:
            pos_out = torch.empty((m, 3), dtype=torch.int32, device=self.device)
            neg_out = torch.empty((m, 3), dtype=torch.int32, device=self.device)
        kg, index, starts, cursor = (indexed[:4] if indexed is not None else (None,) * 4)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        check(lib.kge_sampler_stream_draw(buf.data_ptr(), buf.numel(), self.np_state.data_ptr(), self.py_state.data_ptr(), ptr(batch) if indexed is None else None,
                                          ptr(kg), ptr(index), ptr(starts), ptr(cursor), B, int(ns), int(num_e), ptr(pos_out) if arrays else None,
                                          ptr(neg_out) if arrays else None, counted[0].data_ptr() if counted else None,
                                          counted[1].data_ptr() if counted else None, self.err.data_ptr(), _lib.stream_ptr()))
        return pos_out, neg_out

    def prepare(self, B, ns, num_e):
        """Allocate the workspace and compute / upload the jump polynomials for this batch shape (what build_batch does lazily
        on its first call), so that later calls allocate nothing -- needed before capturing build_batch into a hipGraph."""
        import torch
        self._workspace(int(B), int(ns), int(num_e))
        key = (int(B), int(ns), int(num_e))
        if self.segmented and key not in self._plans:
            buf = np.empty(int(lib.kge_sampler_plan_bytes()) // 8, dtype=np.uint64)
            check(lib.kge_sampler_plan_host(int(B), int(ns), int(num_e), buf.ctypes.data))
            self._plans[key] = torch.from_numpy(buf.view(np.int64)).to(self.device)

    def build_batch(self, batch, ns, num_e, pos_out=None, neg_out=None, indexed=None, counted=None):
        """batch: (B,3) int32 device rows -- or None with indexed=(kg, index, starts, cursor, B): the positives are read in
        place as kg[index[starts[cursor] * B + i]] and the device cursor moves on by one (kge_build_batch_device_indexed).
        counted=(cand_rows ((B + B*ns), 3) int32, cand_counts (B + B*ns) int32), with `indexed`: also emit the batch before
        oversampling and how often each of its rows was picked (the batch in folded form); pos_out=False then skips the
        oversampled arrays altogether."""
        import torch
        from . import _lib
        B = batch.shape[0] if indexed is None else int(indexed[4])
        m = B * int(ns)
        self._workspace(B, int(ns), int(num_e))
        arrays = not (counted is not None and pos_out is False)  # pos_out=False with counted=...: folded form only
        if not arrays:
            pos_out = neg_out = None
        elif pos_out is None:
            pos_out = torch.empty((m, 3), dtype=torch.int32, device=self.device)
            neg_out = torch.empty((m, 3), dtype=torch.int32, device=self.device)
        plan_ptr = None
        if self.segmented:
            key = (B, int(ns), int(num_e))
            if key not in self._plans:  # GF(2) jump polynomials of the segment starts (host, once per batch shape)
                buf = np.empty(int(lib.kge_sampler_plan_bytes()) // 8, dtype=np.uint64)
                check(lib.kge_sampler_plan_host(B, int(ns), int(num_e), buf.ctypes.data))
                self._plans[key] = torch.from_numpy(buf.view(np.int64)).to(self.device)
            plan_ptr = self._plans[key].data_ptr()
        if indexed is not None:
            kg, index, starts, cursor = indexed[:4]
            check(lib.kge_build_batch_device_indexed(self.np_state.data_ptr(), self.py_state.data_ptr(), kg.data_ptr(), index.data_ptr(),
                                                     starts.data_ptr(), cursor.data_ptr(), B, int(ns), int(num_e),
                                                     pos_out.data_ptr() if arrays else None, neg_out.data_ptr() if arrays else None,
                                                     counted[0].data_ptr() if counted else None,
                                                     counted[1].data_ptr() if counted else None, plan_ptr, self._ws.data_ptr(),
                                                     self._ws.numel(), self.err.data_ptr(), _lib.stream_ptr()))
            return pos_out, neg_out
        check(lib.kge_build_batch_device(self.np_state.data_ptr(), self.py_state.data_ptr(), batch.data_ptr(), B, int(ns), int(num_e),
                                         pos_out.data_ptr(), neg_out.data_ptr(), plan_ptr, self._ws.data_ptr(), self._ws.numel(),
                                         self.err.data_ptr(), _lib.stream_ptr()))
        return pos_out, neg_out
