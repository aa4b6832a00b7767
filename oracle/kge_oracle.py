"""NumPy restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All `file:line` citations are relative to /root/reference.  "[TF]" marks semantics that live in
TensorFlow/Keras (absent here): they are restated from the published behaviour of those ops, and are
the part of the oracle whose parity is *unpinned* (no reference-produced vector exists for them).

Default dtype is float64 ("ground truth"); pass dtype=np.float32 to mimic the reference's fp32 graph.
"""
from collections import defaultdict
import random as _pyrandom

import numpy as np

EPSILON = 1e-6  # utils.py:12

MODELS = ("DistMult", "TransE", "ComplEx", "HolE")
LOSSES = ("pointwize_hinge", "pointwize_logistic", "pointwize_square_loss", "pointwize_cross_entropy",
          "pairwize_hinge", "pairwize_logistic", "pairwize_square_loss")  # utils.py:321-330


# ----------------------------------------------------------------------------------------------
# tables (models.py:87-112)
# ----------------------------------------------------------------------------------------------
def glorot_uniform(rows, cols, rs, dtype=np.float32):
    """[TF] keras GlorotUniform on an Embedding matrix (rows, cols): U(-l, l), l = sqrt(6/(rows+cols))."""
    lim = np.sqrt(6.0 / (rows + cols))
    return rs.uniform(-lim, lim, size=(rows, cols)).astype(dtype)


def table_dims(model, e_dim, r_dim):
    """ComplEx doubles both widths (models.py:208-209); everything else keeps them."""
    if model in ("ComplEx", "pRotatE"):  # pRotatE: models.py:469-470
        return 2 * e_dim, 2 * r_dim
    if model == "RotatE":  # models.py:428-429: the entity rows are doubled, the relation rows (phases) are not
        return 2 * e_dim, r_dim
    if model == "HAKE":  # models.py:371-372: entity rows [phase | modulus], relation rows [phase | modulus | bias]
        return 2 * e_dim, 3 * r_dim
    return e_dim, r_dim


# ----------------------------------------------------------------------------------------------
# score functions on gathered rows (models.py `func`)
# ----------------------------------------------------------------------------------------------
def func_distmult(s, p, o):
    """models.py:171-172  reduce_sum(s*p*o, -1)"""
    return np.sum(s * p * o, axis=-1)


def func_transe(s, p, o, norm=1, gamma=12):
    """models.py:186-190.  [TF] tf.norm ord=1 -> sum|x|, ord=2 -> sqrt(sum x^2).  Sign flips on gamma<=0."""
    x = s + p - o
    if norm == 1:
        n = np.sum(np.abs(x), axis=1)
    elif norm == 2:
        n = np.sqrt(np.sum(x * x, axis=1))
    else:
        n = np.sum(np.abs(x) ** norm, axis=1) ** (1.0 / norm)
    return gamma - n if gamma > 0 else n


def func_complex(s, p, o):
    """models.py:212-222.  Rows are [real | imag] halves (tf.split num=2)."""
    h = s.shape[-1] // 2
    sr, si = s[..., :h], s[..., h:]
    pr, pi = p[..., :h], p[..., h:]
    orr, oi = o[..., :h], o[..., h:]
    s1 = sr * pr * orr
    s2 = pr * si * oi
    s3 = pi * sr * oi
    s4 = pi * si * orr
    return np.sum(s1 + s2 + s3 - s4, axis=-1)


def ccorr(a, b):
    """models.py:232-234: real(ifft(conj(fft(a)) * fft(b)))  ==  x_k = sum_i a_i b_{(i+k) mod d}."""
    return np.real(np.fft.ifft(np.conj(np.fft.fft(a, axis=-1)) * np.fft.fft(b, axis=-1), axis=-1))


def func_hole(s, p, o):
    """models.py:231-237."""
    return np.sum(p * ccorr(s, o), axis=-1)


def protate_modulus(e_dim_doubled, gamma=12, epsilon=2, modulus=0.5):
    """models.py:475-476: embedding_range = (gamma + epsilon) / e_dim / 2 on the doubled e_dim; modulus * embedding_range."""
    return modulus * ((gamma + epsilon) / e_dim_doubled / 2)


def func_protate(s, p, o, modulus, gamma=12):
    """models.py:479-489.  `modulus` is the product of :476.  Phases are wrapped as the reference does (atan2(sin, cos))."""
    ph = lambda x: np.arctan2(np.sin(x), np.cos(x))  # noqa: E731
    r = np.sum(np.abs(np.sin((ph(s) + ph(p) - ph(o)) / 2)), axis=1) * modulus
    return gamma - r if gamma > 0 else r


def func_rotate(s, p, o, gamma=12):
    """models.py:439-459, exactly as written: the relation row is a phase vector (half as wide as the entity rows), wrapped by
    atan2(sin, cos); the rotated subject minus the object is SUMMED with its signs (the constructor's `norm` is never applied),
    so the score is linear in s and in o; gamma - sum if gamma > 0 else the sum."""
    h = s.shape[-1] // 2
    re_s, im_s = s[..., :h], s[..., h:]
    re_o, im_o = o[..., :h], o[..., h:]
    phase = np.arctan2(np.sin(p), np.cos(p))
    re_r, im_r = np.cos(phase), np.sin(phase)
    re = re_s * re_r - im_s * im_r - re_o
    im = re_s * im_r + im_s * re_r - im_o
    sc = np.sum(np.concatenate([re, im], axis=-1), axis=-1)
    return gamma - sc if gamma > 0 else sc


def hake_params(e_dim_doubled, gamma=12, epsilon=2, phase_weight=0.5, mod_weight=1.0):
    """models.py:364-381: (phase_weight, mod_weight, embedding_range) with embedding_range = (gamma + epsilon) / e_dim / 2 on the
    doubled e_dim -- what the `norm` slot of score_rows / train_step carries for model == "HAKE"."""
    return (phase_weight, mod_weight, (gamma + epsilon) / e_dim_doubled / 2)


def _hake_parts(s, p, o, prm):
    pw, mw, rng = prm
    h = s.shape[-1] // 2
    sc = np.pi / rng  # phase / (embedding_range / pi), models.py:391-393
    phs, ms = s[..., :h] * sc, s[..., h:]
    pho, mo = o[..., :h] * sc, o[..., h:]
    php, mp, bp = p[..., :h] * sc, p[..., h:2 * h], p[..., 2 * h:]
    b1 = np.minimum(bp, 1.0)                       # K.clip(bias_p, -inf, 1)            models.py:395
    cond = b1 < -np.abs(mp)
    b = np.where(cond, -np.abs(mp), b1)            # tf.where(bias < -|mod_p|, -|mod_p|, bias)   models.py:396
    M = ms * (mp + b) - np.abs(mo) * (1 - b)
    y = (phs + php - pho) / 2
    return pw, mw, sc, h, ms, mo, mp, bp, b, cond, M, y


def func_hake(s, p, o, prm, gamma=12):
    """models.py:383-400, exactly as written: the phase part is a per-row L1 norm (axis=-1), the modulus part is tf.norm(..., ord=2)
    WITHOUT an axis -- [TF] one vector norm over the entire tensor -- i.e. ONE scalar over all rows handed to this call (the whole
    training batch side, or one predict chunk of 32 rows).  The score of a row therefore depends on the rows it is scored with."""
    pw, mw, _sc, _h, _ms, _mo, _mp, _bp, _b, _cond, M, y = _hake_parts(s, p, o, prm)
    r_score = mw * np.sqrt(np.sum(M * M))
    p_score = pw * np.sum(np.abs(np.sin(y)), axis=-1)
    return gamma - (p_score + r_score)


def score_rows(model, s, p, o, norm=1, gamma=12):
    """For model == "pRotatE" the `norm` slot carries the modulus product of models.py:476; for "HAKE" hake_params()."""
    if model == "HAKE":
        return func_hake(s, p, o, norm, gamma)
    if model == "pRotatE":
        return func_protate(s, p, o, norm, gamma)
    if model == "RotatE":
        return func_rotate(s, p, o, gamma)
    if model == "DistMult":
        return func_distmult(s, p, o)
    if model == "TransE":
        return func_transe(s, p, o, norm, gamma)
    if model == "ComplEx":
        return func_complex(s, p, o)
    if model == "HolE":
        return func_hole(s, p, o).astype(s.dtype)
    raise KeyError(model)


def call(model, E, Rel, triples, norm=1, gamma=12, dtype=np.float64, chunk=None):
    """EmbeddingModel.call (models.py:134-162) with dropout inactive (predict / dp=0):
    unstack ids (:140) -> gather E[s], Rel[p], E[o] (:144-150) -> func (:160).  chunk: model.predict scores its input in batches of
    32 rows ([TF] Keras' default predict batch size) -- one `call` per batch, which matters for HAKE alone (its score depends on the
    rows of the call); None = all rows in one call."""
    t = np.asarray(triples).reshape(-1, 3).astype(np.int64)
    E = np.asarray(E, dtype=dtype)
    Rel = np.asarray(Rel, dtype=dtype)
    if chunk and model == "HAKE":
        return np.concatenate([score_rows(model, E[c[:, 0]], Rel[c[:, 1]], E[c[:, 2]], norm, gamma) for c in (t[a:a + chunk] for a in range(0, len(t), chunk))])
    return score_rows(model, E[t[:, 0]], Rel[t[:, 1]], E[t[:, 2]], norm, gamma)


def score_row_grads(model, s, p, o, g, norm=1, gamma=12):
    """d(sum_i g_i * score_i)/d(s,p,o) -- what autodiff hands to the three Embedding gathers [TF]."""
    if model == "HAKE":  # one scalar norm over all rows: every row's modulus gradient carries the SUM of the g's
        pw, mw, sc, h, ms, mo, mp, bp, b, cond, M, y = _hake_parts(s, p, o, norm)
        dph = -pw * g[:, None] * 0.5 * np.sign(np.sin(y)) * np.cos(y) * sc
        r = np.sqrt(np.sum(M * M))
        dM = (-mw * np.sum(g) / r) * M if r > 0 else np.zeros_like(M)  # ([TF] 0/0 = NaN at r == 0; the kernels return 0 there)
        db = dM * (ms + np.abs(mo))
        dms = dM * (mp + b)
        dmo = -dM * np.sign(mo) * (1 - b)
        dmp = dM * ms + np.where(cond, -np.sign(mp) * db, 0.0)
        dbp = np.where(cond | (bp > 1.0), 0.0, db)  # [TF] clip_by_value passes the gradient on [min, max]; where() to the taken branch
        return np.concatenate([dph, dms], 1), np.concatenate([dph, dmp, dbp], 1), np.concatenate([-dph, dmo], 1)
    g = g[:, None]
    if model == "pRotatE":  # autodiff through atan2(sin x, cos x) is the identity; d|u|/du = sign(u) [TF sign(0) = 0]
        ph = lambda x: np.arctan2(np.sin(x), np.cos(x))  # noqa: E731
        y = (ph(s) + ph(p) - ph(o)) / 2
        dx = (-1.0 if gamma > 0 else 1.0) * g * norm * 0.5 * np.sign(np.sin(y)) * np.cos(y)
        return dx, dx, -dx
    if model == "RotatE":  # linear in s and o; d atan2(sin p, cos p)/dp = 1
        h = s.shape[-1] // 2
        re_s, im_s = s[:, :h], s[:, h:]
        c, sn = np.cos(p), np.sin(p)
        gg = (-1.0 if gamma > 0 else 1.0) * g
        ds = np.concatenate([gg * (c + sn), gg * (c - sn)], 1)
        dp = gg * (re_s * (c - sn) - im_s * (c + sn))
        do = -gg * np.ones_like(o)
        return ds, dp, do
    if model == "DistMult":
        return g * p * o, g * s * o, g * s * p
    if model == "TransE":
        x = s + p - o
        sgn = -1.0 if gamma > 0 else 1.0
        if norm == 1:
            dx = np.sign(x)  # [TF] sign(0) = 0
        elif norm == 2:
            n = np.sqrt(np.sum(x * x, axis=1, keepdims=True))
            with np.errstate(invalid="ignore", divide="ignore"):
                dx = x / n  # [TF] NaN at x == 0
        else:  # general p: d (sum |x|^p)^(1/p) / dx_k = |x_k|^(p-1) sign(x_k) n^(1-p)
            n = np.sum(np.abs(x) ** norm, axis=1, keepdims=True) ** (1.0 / norm)
            with np.errstate(invalid="ignore", divide="ignore"):
                dx = np.sign(x) * np.abs(x) ** (norm - 1) * n ** (1 - norm)
        dx = sgn * g * dx
        return dx, dx, -dx
    if model == "ComplEx":
        h = s.shape[-1] // 2
        sr, si = s[:, :h], s[:, h:]
        pr, pi = p[:, :h], p[:, h:]
        orr, oi = o[:, :h], o[:, h:]
        ds = np.concatenate([pr * orr + pi * oi, pr * oi - pi * orr], 1)
        dp = np.concatenate([sr * orr + si * oi, sr * oi - si * orr], 1)
        do = np.concatenate([sr * pr - pi * si, pr * si + pi * sr], 1)
        return g * ds, g * dp, g * do
    if model == "HolE":
        d = s.shape[-1]
        dp = ccorr(s, o)
        ds = ccorr(p, o)  # ds_i = sum_k p_k o_{(i+k)}
        # do_j = sum_k p_k s_{(j-k)}  (circular convolution)
        do = np.real(np.fft.ifft(np.fft.fft(p, axis=-1) * np.fft.fft(s, axis=-1), axis=-1))
        assert do.shape[-1] == d
        return g * ds, g * dp, g * do
    raise KeyError(model)


# ----------------------------------------------------------------------------------------------
# losses (utils.py:288-319): value and d/d(true), d/d(false)
# ----------------------------------------------------------------------------------------------
def _unbroadcast(x, shape):
    while x.ndim > len(shape):
        x = x.sum(axis=0)
    for ax, n in enumerate(shape):
        if n == 1 and x.shape[ax] != 1:
            x = x.sum(axis=ax, keepdims=True)
    return x


def loss_pairs(name, true, false, margin=1, negative_samples=1):
    """reduce_mean=False branch of the pairwize losses (utils.py:303-305, 310-312, 317-319): the raw pair tensor."""
    t = np.asarray(true)
    F = np.asarray(false).reshape(-1, negative_samples)  # utils.py:301/308/315
    z = F - t
    if name == "pairwize_hinge":
        return np.maximum(t.dtype.type(margin) + z, 0)
    if name == "pairwize_logistic":
        return np.log(t.dtype.type(EPSILON) + 1 + np.exp(z))
    if name == "pairwize_square_loss":
        return -(z * z)
    raise KeyError(name)


def loss_value_and_grads(name, true, false, margin=1, negative_samples=1):
    """Returns (loss, dL/dtrue, dL/dfalse) for the reduce_mean=True form the harness uses (tests.py:80-81).

    [TF] relu'(0) = 0; broadcasting is NumPy-style, which is what produces the (m, m) quirk of
    pairwize_* when negative_samples=1 and `true` is (m,) (utils.py:300-312, tests.py:81)."""
    t = np.asarray(true)
    f = np.asarray(false)
    dt = t.dtype
    margin = dt.type(margin)
    eps = dt.type(EPSILON)
    one = dt.type(1)
    if name == "pointwize_hinge":  # utils.py:288-289
        a = margin - t
        b = margin + f
        loss = np.mean(np.maximum(a, 0)) + np.mean(np.maximum(b, 0))
        return loss, -(a > 0).astype(dt) / t.size, (b > 0).astype(dt) / f.size
    if name == "pointwize_logistic":  # utils.py:291-292
        ea = np.exp(-t)
        eb = np.exp(f)
        loss = np.mean(np.log(eps + one + ea)) + np.mean(np.log(eps + one + eb))
        return loss, -(ea / (eps + one + ea)) / t.size, (eb / (eps + one + eb)) / f.size
    if name == "pointwize_square_loss":  # utils.py:294-295
        a = margin - t
        b = margin + f
        loss = np.mean(a * a) + np.mean(b * b)
        return loss, -2 * a / t.size, 2 * b / f.size
    if name == "pointwize_cross_entropy":
        # utils.py:297-298  [TF] keras.backend.binary_crossentropy: clip to [eps, 1 - eps], eps added inside both logs.  TF
        # evaluates this in float32, and the saturated branches depend on that: fp32(1 - 1e-7) = 1 - 2**-23, so a score >= 1
        # costs -log(2**-23 + 1e-7) as a negative, not -log(2e-7).  The constants are therefore the float32 ones in every dtype.
        k = dt.type(np.float32(1e-7))
        hi = dt.type(np.float32(1) - np.float32(1e-7))
        ct = np.clip(t, k, hi)
        cf = np.clip(f, k, hi)
        loss = -np.mean(np.log(ct + k)) - np.mean(np.log(one - cf + k))
        gt = np.where((t >= k) & (t <= hi), -one / (ct + k), 0) / t.size  # clip_by_value passes the gradient on [lo, hi]
        gf = np.where((f >= k) & (f <= hi), one / (one - cf + k), 0) / f.size
        return loss, gt.astype(dt), gf.astype(dt)
    if name in ("pairwize_hinge", "pairwize_logistic", "pairwize_square_loss"):
        F = f.reshape(-1, negative_samples)  # utils.py:301/308/315
        z = F - t  # NumPy broadcast == TF broadcast
        if name == "pairwize_hinge":  # utils.py:300-305
            y = margin + z
            tmp = np.maximum(y, 0)
            dz = (y > 0).astype(dt)
        elif name == "pairwize_logistic":  # utils.py:307-312 (margin unused)
            e = np.exp(z)
            tmp = np.log(eps + one + e)
            dz = e / (eps + one + e)
        else:  # utils.py:314-319
            tmp = -(z * z)
            dz = -2 * z
        dz = dz / tmp.size
        dF = _unbroadcast(dz, F.shape).reshape(f.shape)
        dT = -_unbroadcast(dz, t.shape if t.ndim == dz.ndim else (1,) * (dz.ndim - t.ndim) + t.shape).reshape(t.shape)
        return np.mean(tmp), dT.astype(dt), dF.astype(dt)
    raise KeyError(name)  # utils.py:322-330 dict lookup


def loss_value(name, true, false, margin=1, negative_samples=1):
    return loss_value_and_grads(name, true, false, margin, negative_samples)[0]


# ----------------------------------------------------------------------------------------------
# optimizers [TF] (tests.py:86 compiles with 'adam'; 'sgd' named in the north star)
# ----------------------------------------------------------------------------------------------
class AdamTF:
    """[TF] Keras Adam with IndexedSlices gradients: duplicates summed, then DENSE moment decay and DENSE
    variable update over every row; epsilon=1e-7 sits outside the bias correction:
        lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m = b1*m+(1-b1)*g;  v = b2*v+(1-b2)*g^2;  w -= lr_t*m/(sqrt(v)+eps)"""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        # [TF] OptimizerV2 keeps its hyper-parameters as tensors of the VARIABLE's dtype (`_get_hyper(name, var_dtype)`, float32 for
        # the embedding tables): beta_2 is 0.999 rounded to float32 (0.99900001287...), and 1 - beta_2^t feels that rounding at 1.3e-5
        # relative for small t (lr_t at 6e-6).  The restatement therefore holds the float32-rounded values (as doubles).
        self.lr, self.b1, self.b2, self.eps = (float(np.float32(x)) for x in (lr, b1, b2, eps))
        self.t = 0
        self.slots = {}

    def lr_t(self, t=None):
        t = self.t if t is None else t
        return self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def begin_step(self):
        self.t += 1

    def apply(self, key, w, g):
        m, v = self.slots.setdefault(key, (np.zeros_like(w), np.zeros_like(w)))
        dt = w.dtype.type
        m *= dt(self.b1)
        m += dt(1 - self.b1) * g
        v *= dt(self.b2)
        v += dt(1 - self.b2) * (g * g)
        w -= dt(self.lr_t()) * m / (np.sqrt(v) + dt(self.eps))


class SGDTF:
    """[TF] Keras SGD, lr=0.01, no momentum: w[ids] -= lr*g with duplicate ids accumulated."""

    def __init__(self, lr=0.01):
        self.lr = float(np.float32(lr))  # (float32 hyper-parameter tensor, as in AdamTF)
        self.t = 0

    def begin_step(self):
        self.t += 1

    def apply(self, key, w, g):
        w -= w.dtype.type(self.lr) * g


def make_optimizer(name, **kw):
    return {"adam": AdamTF, "sgd": SGDTF}[name.lower()](**kw)


# ----------------------------------------------------------------------------------------------
# dropout on the gathered rows (models.py:122,144-148).  [TF] inverted dropout: kept entries scaled by 1/(1-rate).
# TensorFlow's mask stream is not reproducible; the checker regenerates the product's counter-based masks
# (kge-keras_amd/csrc/dropout.hpp: splitmix64 key of (seed, batch row, role) + a 32-bit mix per element pair).
# ----------------------------------------------------------------------------------------------
def dropout_multipliers(seed, n_rows, role, width, rate, row_offset=0):
    """(n_rows, width) array of 0 / 1/(1-rate) for batch rows row_offset..row_offset+n_rows-1 in role 0/1/2 (s/p/o)."""
    M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        t = np.arange(row_offset, row_offset + n_rows, dtype=np.uint64)
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (np.uint64(3) * t + np.uint64(role + 1))) & M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        z = z ^ (z >> np.uint64(31))
        lo = (z & np.uint64(0xFFFFFFFF)).astype(np.uint32)[:, None]
        hi = (z >> np.uint64(32)).astype(np.uint32)[:, None]
        nw = (width + 1) // 2
        w = np.arange(nw, dtype=np.uint32)[None, :]
        x = lo + w * (hi | np.uint32(1))  # Weyl sequence: start lo, odd stride hi | 1 (uint32 wrap-around)
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> np.uint32(15))
    bits = np.empty((n_rows, 2 * nw), dtype=np.uint32)
    bits[:, 0::2] = x & np.uint32(0xFFFF)
    bits[:, 1::2] = x >> np.uint32(16)
    thr = int(np.float32(rate) * np.float32(65536.0) + np.float32(0.5))
    thr = max(0, min(65535, thr))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return np.where(bits[:, :width] >= thr, scale, np.float32(0.0)).astype(np.float64)


# ----------------------------------------------------------------------------------------------
# one training batch: the default Keras train_step over add_loss (tests.py:63-89)
# ----------------------------------------------------------------------------------------------
def _scatter_add(G, idx, rows):
    """G[idx] += rows with duplicates summed ([TF] IndexedSlices dedup).  Sparse mat-vec for big batches."""
    if len(idx) < 4096:
        np.add.at(G, idx, rows)
        return
    from scipy.sparse import csr_matrix
    m = len(idx)
    S = csr_matrix((np.ones(m, dtype=rows.dtype), (idx, np.arange(m))), shape=(G.shape[0], m))
    G += S @ rows


def dense_grads(model, E, Rel, pos, neg, loss, margin=1, negative_samples=1, norm=1, gamma=12, regularization=0.0,
                dropout=0.0, seed=0):
    """Forward on pos and neg with the SAME tables (tests.py:77-78), loss (tests.py:80-81), and the dense
    equivalents of the IndexedSlices gradients (duplicates summed) [TF].  Returns (loss, scores_pos,
    scores_neg, gE, gRel).  `regularization` adds w*sum|E|^3 over the whole entity table (models.py:10-11,78-92)."""
    dt = E.dtype
    pos = np.asarray(pos).reshape(-1, 3).astype(np.int64)
    neg = np.asarray(neg).reshape(-1, 3).astype(np.int64)
    rows, masks = [], []
    for t, off in ((pos, 0), (neg, len(pos))):
        s_, p_, o_ = E[t[:, 0]], Rel[t[:, 1]], E[t[:, 2]]
        if dropout:  # batch row index = position in [pos ; neg]
            ms = [dropout_multipliers(seed, len(t), r, (Rel if r == 1 else E).shape[1], dropout, off).astype(dt) for r in range(3)]
            s_, p_, o_ = s_ * ms[0], p_ * ms[1], o_ * ms[2]
            masks.append(ms)
        rows.append((s_, p_, o_))
    sp = score_rows(model, *rows[0], norm, gamma).astype(dt)
    sn = score_rows(model, *rows[1], norm, gamma).astype(dt)
    L, gt, gf = loss_value_and_grads(loss, sp, sn, margin, negative_samples)
    gE = np.zeros_like(E)
    gR = np.zeros_like(Rel)
    for i, (t, (s, p, o), g) in enumerate(((pos, rows[0], gt), (neg, rows[1], gf))):
        ds, dp, do = score_row_grads(model, s, p, o, g.astype(dt), norm, gamma)
        if dropout:
            ds, dp, do = ds * masks[i][0], dp * masks[i][1], do * masks[i][2]
        _scatter_add(gE, t[:, 0], ds.astype(dt))
        _scatter_add(gR, t[:, 1], dp.astype(dt))
        _scatter_add(gE, t[:, 2], do.astype(dt))
    if regularization:
        w = dt.type(regularization)
        L = L + w * np.sum(np.abs(E) ** 3)
        gE += 3 * w * E * np.abs(E)
    return L, sp, sn, gE, gR


def train_step(model, E, Rel, opt, pos, neg, loss="pointwize_hinge", margin=1, negative_samples=1,
               norm=1, gamma=12, regularization=0.0, dropout=0.0, seed=0):
    """In-place update of E and Rel; returns the batch loss (what Keras logs as `loss`)."""
    L, _, _, gE, gR = dense_grads(model, E, Rel, pos, neg, loss, margin, negative_samples, norm, gamma, regularization,
                                  dropout, seed)
    opt.begin_step()
    opt.apply("E", E, gE)
    opt.apply("R", Rel, gR)
    return L


# ----------------------------------------------------------------------------------------------
# batch construction (utils.py:114-153, tests.py:29-61).  NumPy's legacy global RandomState and
# Python's `random` ARE the reference's dependencies, so the oracle calls them exactly as it does.
# ----------------------------------------------------------------------------------------------
def generate_negative(kg, N, negative=2, np_random=np.random):
    """utils.py:114-121: R = repeat(p, negative); fs = randint(0,N,(negative*len,1)); fo likewise (fs first)."""
    kg = np.asarray(kg)
    R = np.repeat(kg[:, 1].reshape((-1, 1)), negative, axis=0)
    fs = np_random.randint(0, N, size=(negative * len(kg), 1))
    fo = np_random.randint(0, N, size=(negative * len(kg), 1))
    return np.stack([fs, R, fo], axis=1)  # (negative*len, 3, 1)


def oversample_data(kgs, py_random=_pyrandom):
    """utils.py:123-147 (the branch the harness uses): every array independently resampled with
    replacement to the longest length with random.choices, in list order."""
    kgs = [list(kg) for kg in kgs]
    m = max(map(len, kgs))
    out = [py_random.choices(kg, k=m) for kg in kgs]
    return [np.squeeze(np.asarray(o)) for o in out]


def epoch_permutation(n, np_random=np.random):
    """tests.py:50-53."""
    idx = np.arange(n)
    np_random.shuffle(idx)
    return idx


def harness_batch(kg, index, b, B, ns, num_e, np_random=np.random, py_random=_pyrandom):
    """tests.py:43-61: slice of the epoch permutation -> generate_negative -> oversample_data."""
    sl = index[b * B:(b + 1) * B]
    tmp_kg = np.asarray([kg[i] for i in sl])
    neg = generate_negative(tmp_kg, num_e, ns, np_random)
    pos_os, neg_os = oversample_data([tmp_kg, neg], py_random)
    return pos_os, neg_os


def pad(kg, bs, py_random=_pyrandom):
    """utils.py:149-153."""
    kg = list(kg)
    while len(kg) % bs != 0:
        kg.append(py_random.choice(kg))
    return np.asarray(kg)


def load_kg(path):
    """utils.py:106-112."""
    out = []
    with open(path, "r") as f:
        for l in f:
            out.append(l.strip().split())
    return out


# ----------------------------------------------------------------------------------------------
# filtered ranking (utils.py:165-264)
# ----------------------------------------------------------------------------------------------
def build_filters(filtering_triples):
    """utils.py:201-205."""
    filter_h = defaultdict(set)
    filter_t = defaultdict(set)
    for s, p, o in filtering_triples:
        filter_h[(p, o)].add(s)
        filter_t[(s, p)].add(o)
    return filter_h, filter_t


def candidate_rows(triple, num_entities, side, filt):
    """gen_tail_data / gen_head_data (utils.py:165-196): target first, then every entity that is neither the
    target nor filtered, ascending.  Raises ValueError like list.remove when a filtered id is absent."""
    s, p, o = triple
    target, key = (o, (s, p)) if side == "tail" else (s, (p, o))
    cand = list(range(num_entities))
    cand.remove(target)
    for x in filt[key]:
        cand.remove(x)
    ids = np.asarray([target] + cand, dtype=np.int64)
    out = np.empty((len(ids), 3), dtype=np.int64)
    out[:, 0] = s if side == "tail" else ids
    out[:, 1] = p
    out[:, 2] = ids if side == "tail" else o
    return out


def rank_literal(scores):
    """utils.py:211-215 with scipy's rankdata(.,'max') restated: target is row 0;
    r0 = #{j : res_j <= res_0};  real_rank = n - r0 + 1  ==  1 + #{j : res_j > res_0}."""
    res = np.asarray(scores).reshape(-1)
    r0 = int(np.sum(res <= res[0]))
    return len(res) - r0 + 1


def metrics_from_ranks(tail_ranks, head_ranks):
    """utils.py:216-262.  hits@k  <=>  real_rank <= k."""
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


def validate(predict, test_data, num_entities, filtering_triples):
    """utils.py:199-264 restated literally (Python loop per query; small inputs only)."""
    filter_h, filter_t = build_filters(filtering_triples)
    tails = [rank_literal(predict(candidate_rows(t, num_entities, "tail", filter_t))) for t in test_data]
    heads = [rank_literal(predict(candidate_rows(t, num_entities, "head", filter_h))) for t in test_data]
    return metrics_from_ranks(tails, heads), np.asarray(tails), np.asarray(heads)


def ranks_closed_form(model, E, Rel, test, filtering_triples, norm=1, gamma=12, dtype=np.float64, margin_out=None):
    """Vectorised 1-vs-all: rank = 1 + #{kept c : score_c > score_target}.  Returns
    (tail_ranks, head_ranks).  If `margin_out` is a list, appends per query the smallest |score_c -
    score_target| over kept candidates (to let a float32 checker skip knife-edge ties)."""
    E = np.asarray(E, dtype=dtype)
    Rel = np.asarray(Rel, dtype=dtype)
    N = E.shape[0]
    filter_h, filter_t = build_filters(filtering_triples)
    allc = np.arange(N, dtype=np.int64)
    out = {"tail": [], "head": []}
    for s, p, o in test:
        for side in ("tail", "head"):
            if side == "tail":
                sc = score_rows(model, E[s][None].repeat(N, 0), Rel[p][None].repeat(N, 0), E, norm, gamma)
                tgt, fl = o, filter_t.get((s, p), ())
            else:
                sc = score_rows(model, E, Rel[p][None].repeat(N, 0), E[o][None].repeat(N, 0), norm, gamma)
                tgt, fl = s, filter_h.get((p, o), ())
            keep = np.ones(N, dtype=bool)
            keep[tgt] = False
            if len(fl):
                keep[np.fromiter(fl, dtype=np.int64)] = False
            out[side].append(1 + int(np.sum(sc[keep] > sc[tgt])))
            if margin_out is not None:
                margin_out.append(float(np.min(np.abs(sc[keep] - sc[tgt]))) if keep.any() else np.inf)
    del allc
    return np.asarray(out["tail"]), np.asarray(out["head"])


def sorted_id_maps(*splits):
    """Deterministic replacement for tests.py:102-111 (whose ids depend on PYTHONHASHSEED): sorted strings."""
    ents = sorted({x for sp in splits for (s, _, o) in sp for x in (s, o)})
    rels = sorted({p for sp in splits for (_, p, _) in sp})
    return {e: i for i, e in enumerate(ents)}, {r: i for i, r in enumerate(rels)}


# ----------------------------------------------------------------------------------------------
# ConvE (models.py:239-275).  [TF] throughout: Conv2D 'valid' cross-correlation + bias, BatchNormalization
# (momentum 0.99, epsilon 1e-3, training mode = batch statistics; the 4-D one is Keras' fused kernel whose MOVING variance
# takes Bessel's correction, the 2-D one is not fused and does not), ReLU, Dropout, Flatten in (H, W, C) order, Dense.
# The two calls of the wrapper (tests.py:77-78) are separate invocations: batch statistics are per call.
# ----------------------------------------------------------------------------------------------
def conve_shape(dim, filters=8, kh=3, kw=3):
    """models.py:250-251: (w, h) = the largest factor pair i <= sqrt(dim); image (2w, h); Conv2D(filters, (conv_size_w,
    conv_size_h)) (models.py:254: kernel (kh, kw) over (rows, columns), 'valid', stride 1) -> (2w-kh+1, h-kw+1, filters)."""
    fac = [(int(i), int(dim / i)) for i in range(1, int(dim ** 0.5) + 1) if dim % i == 0]
    w, h = fac[-1]
    assert w > 1 or h > 1  # models.py:252
    Ho, Wo = 2 * w - kh + 1, h - kw + 1
    return w, h, Ho, Wo, Ho * Wo * filters


def conve_init(dim, rs, filters=8, dtype=np.float32, kh=3, kw=3):
    """Trainable parameters with the Keras default initialisers (Glorot uniform kernels, zero biases, BN gamma 1 / beta 0)
    and the non-trainable moving statistics (mean 0, variance 1)."""
    w, h, Ho, Wo, F = conve_shape(dim, filters, kh, kw)
    lim_k = np.sqrt(6.0 / (kh * kw * 1 + kh * kw * filters))  # fan_in = kh*kw*1, fan_out = kh*kw*filters
    lim_d = np.sqrt(6.0 / (F + dim))
    return dict(K=rs.uniform(-lim_k, lim_k, (kh, kw, filters)).astype(dtype), bc=np.zeros(filters, dtype), g1=np.ones(filters, dtype),
                b1=np.zeros(filters, dtype), Wd=rs.uniform(-lim_d, lim_d, (F, dim)).astype(dtype), bd=np.zeros(dim, dtype),
                g2=np.ones(dim, dtype), b2=np.zeros(dim, dtype),
                mm1=np.zeros(filters, dtype), mv1=np.ones(filters, dtype), mm2=np.zeros(dim, dtype), mv2=np.ones(dim, dtype))


CONVE_TRAINABLE = ("K", "bc", "g1", "b1", "Wd", "bd", "g2", "b2")
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def conve_forward(P, s, p, o, training=False, hidden_dp=0.0, seed=0, row_offset=0, cache=None, update_moving=True):
    """x = ConvE.func's hidden vector (m, dim) and score = sum(x * o, -1) (models.py:265-275)."""
    m, dim = s.shape
    kh, kw, C = P["K"].shape
    w, h, Ho, Wo, F = conve_shape(dim, C, kh, kw)
    X = np.concatenate([s.reshape(m, w, h), p.reshape(m, w, h)], axis=1)  # (m, 2w, h): tf.concat axis=1
    Y = np.zeros((m, Ho, Wo, C), dtype=s.dtype)
    for a in range(kh):
        for b in range(kw):
            Y += X[:, a:a + Ho, b:b + Wo, None] * P["K"][a, b][None, None, None, :]
    Y += P["bc"]
    if training:
        mu1 = Y.mean(axis=(0, 1, 2))
        var1 = Y.var(axis=(0, 1, 2))
        if update_moving:
            n1 = m * Ho * Wo
            P["mm1"][...] = P["mm1"] * BN_MOMENTUM + mu1 * (1 - BN_MOMENTUM)
            P["mv1"][...] = P["mv1"] * BN_MOMENTUM + var1 * (n1 / max(n1 - 1, 1)) * (1 - BN_MOMENTUM)  # fused BN: Bessel
    else:
        mu1, var1 = P["mm1"], P["mv1"]
    inv1 = 1.0 / np.sqrt(var1 + BN_EPS)
    xh1 = (Y - mu1) * inv1
    A = np.maximum(xh1 * P["g1"] + P["b1"], 0).reshape(m, F)
    d1 = np.ones_like(A)
    if training and hidden_dp:
        d1 = dropout_multipliers(seed, m, 3, F, hidden_dp, row_offset).astype(A.dtype)
        A = A * d1
    Z = A @ P["Wd"] + P["bd"]
    if training:
        mu2, var2 = Z.mean(0), Z.var(0)
        if update_moving:
            P["mm2"][...] = P["mm2"] * BN_MOMENTUM + mu2 * (1 - BN_MOMENTUM)
            P["mv2"][...] = P["mv2"] * BN_MOMENTUM + var2 * (1 - BN_MOMENTUM)
    else:
        mu2, var2 = P["mm2"], P["mv2"]
    inv2 = 1.0 / np.sqrt(var2 + BN_EPS)
    xh2 = (Z - mu2) * inv2
    x = np.maximum(xh2 * P["g2"] + P["b2"], 0)
    d2 = np.ones_like(x)
    if training and hidden_dp:
        d2 = dropout_multipliers(seed, m, 4, dim, hidden_dp, row_offset).astype(x.dtype)
        x = x * d2
    if cache is not None:
        cache.update(X=X, xh1=xh1, inv1=inv1, A=A, d1=d1, xh2=xh2, inv2=inv2, x=x, d2=d2, shape=(w, h, Ho, Wo, F, C))
    return x, np.sum(x * o, axis=-1)


def conve_backward(P, cache, o, g):
    """Gradients of sum_i g_i * score_i (training mode): returns (ds, dp, do, dict of parameter gradients)."""
    w, h, Ho, Wo, F, C = cache["shape"]
    m = o.shape[0]
    x, A = cache["x"], cache["A"]
    do = g[:, None] * x
    dx = g[:, None] * o
    relu2 = (cache["xh2"] * P["g2"] + P["b2"]) > 0
    dy2 = dx * cache["d2"] * relu2
    dg2 = np.sum(dy2 * cache["xh2"], 0)
    db2 = np.sum(dy2, 0)
    dxh2 = dy2 * P["g2"]
    dZ = cache["inv2"] * (dxh2 - dxh2.mean(0) - cache["xh2"] * (dxh2 * cache["xh2"]).mean(0))
    dWd = A.T @ dZ
    dbd = dZ.sum(0)
    dA = dZ @ P["Wd"].T
    relu1 = ((cache["xh1"] * P["g1"] + P["b1"]) > 0).reshape(m, F)
    dy1 = (dA * cache["d1"] * relu1).reshape(m, Ho, Wo, C)
    dg1 = np.sum(dy1 * cache["xh1"], axis=(0, 1, 2))
    db1 = np.sum(dy1, axis=(0, 1, 2))
    dxh1 = dy1 * P["g1"]
    dY = cache["inv1"] * (dxh1 - dxh1.mean(axis=(0, 1, 2)) - cache["xh1"] * (dxh1 * cache["xh1"]).mean(axis=(0, 1, 2)))
    dbc = dY.sum(axis=(0, 1, 2))
    X = cache["X"]
    dK = np.zeros_like(P["K"])
    dX = np.zeros_like(X)
    for a in range(P["K"].shape[0]):
        for b in range(P["K"].shape[1]):
            dK[a, b] = np.einsum("mij,mijc->c", X[:, a:a + Ho, b:b + Wo], dY)
            dX[:, a:a + Ho, b:b + Wo] += dY @ P["K"][a, b]
    ds = dX[:, :w, :].reshape(m, w * h)
    dp = dX[:, w:, :].reshape(m, w * h)
    return ds, dp, do, dict(K=dK, bc=dbc, g1=dg1, b1=db1, Wd=dWd, bd=dbd, g2=dg2, b2=db2)


def conve_train_step(P, E, Rel, opt, pos, neg, loss="pointwize_hinge", margin=1, negative_samples=1, dropout=0.0, hidden_dp=0.0, seed=0):
    """One Keras train step of the two-input wrapper with a ConvE embedding model: the head runs once on pos and once on
    neg (separate batch statistics), both update the moving statistics; every trainable variable gets an Adam/SGD update."""
    dt = E.dtype
    pos = np.asarray(pos).reshape(-1, 3).astype(np.int64)
    neg = np.asarray(neg).reshape(-1, 3).astype(np.int64)
    caches, rows, masks, scores = [], [], [], []
    for t, off in ((pos, 0), (neg, len(pos))):
        s_, p_, o_ = E[t[:, 0]], Rel[t[:, 1]], E[t[:, 2]]
        ms = [np.ones_like(s_)] * 3
        if dropout:
            ms = [dropout_multipliers(seed, len(t), r, (Rel if r == 1 else E).shape[1], dropout, off).astype(dt) for r in range(3)]
            s_, p_, o_ = s_ * ms[0], p_ * ms[1], o_ * ms[2]
        c = {}
        _, sc = conve_forward(P, s_, p_, o_, True, hidden_dp, seed, off, c)
        caches.append(c); rows.append((s_, p_, o_)); masks.append(ms); scores.append(sc.astype(dt))
    L, gt, gf = loss_value_and_grads(loss, scores[0], scores[1], margin, negative_samples)
    gE, gR = np.zeros_like(E), np.zeros_like(Rel)
    gP = {k: np.zeros_like(P[k]) for k in CONVE_TRAINABLE}
    for t, c, (s_, p_, o_), ms, g in ((pos, caches[0], rows[0], masks[0], gt), (neg, caches[1], rows[1], masks[1], gf)):
        ds, dp, do, gp = conve_backward(P, c, o_, g.astype(dt))
        _scatter_add(gE, t[:, 0], (ds * ms[0]).astype(dt))
        _scatter_add(gR, t[:, 1], (dp * ms[1]).astype(dt))
        _scatter_add(gE, t[:, 2], (do * ms[2]).astype(dt))
        for k in CONVE_TRAINABLE:
            gP[k] += gp[k]
    opt.begin_step()
    opt.apply("E", E, gE)
    opt.apply("R", Rel, gR)
    for k in CONVE_TRAINABLE:
        opt.apply("conve_" + k, P[k], gP[k])
    return L, scores[0], scores[1]


# ----------------------------------------------------------------------------------------------
# LiteralE-style input path (models.py:114-120,152-158; the configuration the reference's own script uses, tests.py:113,125-126):
#   l = dp(literal_layer(id))            frozen (len(literals), L) table
#   s = Dense(e_dim, activation)(concat([dp(entity_embedding(id)), l]))      for s and for o, the SAME Dense
# [TF] Dense: x @ W + b with W Glorot-uniform ((e_dim + L), e_dim), b zeros; Keras activation strings.
# Dropout multipliers: roles 0/1/2 = s/p/o rows as everywhere, 5/6 = the literal rows of s / o.
# ----------------------------------------------------------------------------------------------
ACTIVATIONS = ("linear", "relu", "sigmoid", "tanh", "softplus", "softsign", "elu", "selu", "exponential", "softmax")  # (softmax: row-wise, literal_transform)
_SELU_A, _SELU_S = 1.6732632423543772, 1.0507009873554805


def activation(name, z):
    """(act(z), act'(z)) for the Keras activation strings the reference's docstring lists (softmax excluded)."""
    if name in (None, "linear"):
        return z, np.ones_like(z)
    if name == "relu":
        return np.maximum(z, 0), (z > 0).astype(z.dtype)
    if name == "sigmoid":
        h = 1.0 / (1.0 + np.exp(-z))
        return h, h * (1 - h)
    if name == "tanh":
        h = np.tanh(z)
        return h, 1 - h * h
    if name == "softplus":
        return np.logaddexp(0, z), 1.0 / (1.0 + np.exp(-z))
    if name == "softsign":
        return z / (1 + np.abs(z)), 1.0 / (1 + np.abs(z)) ** 2
    if name == "elu":
        return np.where(z > 0, z, np.exp(np.minimum(z, 0)) - 1), np.where(z > 0, 1.0, np.exp(np.minimum(z, 0)))
    if name == "selu":
        e = np.exp(np.minimum(z, 0))
        return _SELU_S * np.where(z > 0, z, _SELU_A * (e - 1)), _SELU_S * np.where(z > 0, 1.0, _SELU_A * e)
    if name == "exponential":
        e = np.exp(z)
        return e, e
    raise KeyError(name)


def literal_init(e_dim, L, rs, dtype=np.float32):
    lim = np.sqrt(6.0 / ((e_dim + L) + e_dim))
    return dict(W=rs.uniform(-lim, lim, (e_dim + L, e_dim)).astype(dtype), b=np.zeros(e_dim, dtype))


def literal_transform(H, E, Lit, ids, act, me=None, ml=None):
    """act([E[ids]*me | Lit[ids]*ml] @ W + b); returns (rows, cache)."""
    a_e, a_l = E[ids], Lit[ids]
    if me is not None:
        a_e, a_l = a_e * me, a_l * ml
    A = np.concatenate([a_e, a_l], axis=1)
    Z = A @ H["W"] + H["b"]
    if act == "softmax":  # [TF] keras.activations.softmax over the layer's outputs (axis -1): the one choice that is not element-wise
        e = np.exp(Z - Z.max(axis=1, keepdims=True))
        h = e / e.sum(axis=1, keepdims=True)
        return h, dict(A=A, dh=None, h=h)
    h, dh = activation(act, Z)
    return h, dict(A=A, dh=dh)


def literal_act_backward(cache, g):
    """dL/dZ from dL/dh: g * act'(Z) element-wise, or the softmax Jacobian h (g - <g, h>) per row."""
    if cache["dh"] is None:
        h = cache["h"]
        return h * (g - np.sum(g * h, axis=1, keepdims=True))
    return g * cache["dh"]


def literal_train_step(model, H, act, E, Rel, Lit, opt, pos, neg, loss="pointwize_hinge", margin=1, negative_samples=1, norm=1, gamma=12,
                       dropout=0.0, seed=0):
    """One Keras train step with the literal path in front of `model`'s func.  E, Rel, H['W'], H['b'] are updated in place."""
    dt = E.dtype
    pos = np.asarray(pos).reshape(-1, 3).astype(np.int64)
    neg = np.asarray(neg).reshape(-1, 3).astype(np.int64)
    d, L = E.shape[1], Lit.shape[1]
    per = []
    for t, off in ((pos, 0), (neg, len(pos))):
        m = len(t)
        one_e, one_l = np.ones((m, d), dt), np.ones((m, L), dt)
        if dropout:
            mk = lambda role, w: dropout_multipliers(seed, m, role, w, dropout, off).astype(dt)  # noqa: E731
            ms, mp, mo, mls, mlo = mk(0, d), mk(1, Rel.shape[1]), mk(2, d), mk(5, L), mk(6, L)
        else:
            ms, mp, mo, mls, mlo = one_e, np.ones((m, Rel.shape[1]), dt), one_e, one_l, one_l
        s_, cs = literal_transform(H, E, Lit, t[:, 0], act, ms, mls)
        o_, co = literal_transform(H, E, Lit, t[:, 2], act, mo, mlo)
        p_ = Rel[t[:, 1]] * mp
        per.append(dict(t=t, s=s_, p=p_, o=o_, cs=cs, co=co, ms=ms, mp=mp, mo=mo, sc=score_rows(model, s_, p_, o_, norm, gamma).astype(dt)))
    Lv, gt, gf = loss_value_and_grads(loss, per[0]["sc"], per[1]["sc"], margin, negative_samples)
    gE, gR = np.zeros_like(E), np.zeros_like(Rel)
    gW, gb = np.zeros_like(H["W"]), np.zeros_like(H["b"])
    for c, g in ((per[0], gt), (per[1], gf)):
        ds, dp, do = score_row_grads(model, c["s"], c["p"], c["o"], g.astype(dt), norm, gamma)
        _scatter_add(gR, c["t"][:, 1], (dp * c["mp"]).astype(dt))
        for dh, cache, mask, col in ((ds, c["cs"], c["ms"], 0), (do, c["co"], c["mo"], 2)):
            dZ = literal_act_backward(cache, dh)
            gW += cache["A"].T @ dZ
            gb += dZ.sum(0)
            dA = dZ @ H["W"].T
            _scatter_add(gE, c["t"][:, col], (dA[:, :d] * mask).astype(dt))
    opt.begin_step()
    opt.apply("E", E, gE)
    opt.apply("R", Rel, gR)
    opt.apply("lit_W", H["W"], gW)
    opt.apply("lit_b", H["b"], gb)
    return Lv, per[0]["sc"], per[1]["sc"]


def literal_conve_train_step(P, H, act, E, Rel, Lit, opt, pos, neg, loss="pointwize_hinge", margin=1, negative_samples=1, dropout=0.0, hidden_dp=0.0,
                             seed=0):
    """One Keras train step with the literal path (models.py:152-158) in front of ConvE.func (:264-275): s and o are the Dense's outputs on
    [dp(E[id]) ; dp(Lit[id])], p = dp(Rel[p]); the head runs once on the positives and once on the negatives in training mode (batch
    statistics, hidden dropout: conve_forward / conve_backward).  E, Rel, the Dense (H) and the head's trainables (P) are updated in place."""
    dt = E.dtype
    pos = np.asarray(pos).reshape(-1, 3).astype(np.int64)
    neg = np.asarray(neg).reshape(-1, 3).astype(np.int64)
    d, L = E.shape[1], Lit.shape[1]
    per = []
    for t, off in ((pos, 0), (neg, len(pos))):
        m = len(t)
        one_e, one_l = np.ones((m, d), dt), np.ones((m, L), dt)
        if dropout:
            mk = lambda role, w: dropout_multipliers(seed, m, role, w, dropout, off).astype(dt)  # noqa: E731
            ms, mp, mo, mls, mlo = mk(0, d), mk(1, Rel.shape[1]), mk(2, d), mk(5, L), mk(6, L)
        else:
            ms, mp, mo, mls, mlo = one_e, np.ones((m, Rel.shape[1]), dt), one_e, one_l, one_l
        s_, cs = literal_transform(H, E, Lit, t[:, 0], act, ms, mls)
        o_, co = literal_transform(H, E, Lit, t[:, 2], act, mo, mlo)
        p_ = Rel[t[:, 1]] * mp
        c = {}
        _, sc = conve_forward(P, s_, p_, o_, True, hidden_dp, seed, off, c)
        per.append(dict(t=t, o=o_, cs=cs, co=co, ms=ms, mp=mp, mo=mo, cache=c, sc=sc.astype(dt)))
    Lv, gt, gf = loss_value_and_grads(loss, per[0]["sc"], per[1]["sc"], margin, negative_samples)
    gE, gR = np.zeros_like(E), np.zeros_like(Rel)
    gW, gb = np.zeros_like(H["W"]), np.zeros_like(H["b"])
    gP = {k: np.zeros_like(P[k]) for k in CONVE_TRAINABLE}
    for c, g in ((per[0], gt), (per[1], gf)):
        ds, dp, do, gp = conve_backward(P, c["cache"], c["o"], g.astype(dt))
        for k in CONVE_TRAINABLE:
            gP[k] += gp[k]
        _scatter_add(gR, c["t"][:, 1], (dp * c["mp"]).astype(dt))
        for dh, cache, mask, col in ((ds, c["cs"], c["ms"], 0), (do, c["co"], c["mo"], 2)):
            dZ = literal_act_backward(cache, dh)
            gW += cache["A"].T @ dZ
            gb += dZ.sum(0)
            dA = dZ @ H["W"].T
            _scatter_add(gE, c["t"][:, col], (dA[:, :d] * mask).astype(dt))
    opt.begin_step()
    opt.apply("E", E, gE)
    opt.apply("R", Rel, gR)
    for k in CONVE_TRAINABLE:
        opt.apply("conve_" + k, P[k], gP[k])
    opt.apply("lit_W", H["W"], gW)
    opt.apply("lit_b", H["b"], gb)
    return Lv, per[0]["sc"], per[1]["sc"]
