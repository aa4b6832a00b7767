#!/usr/bin/env python3
"""bench.py -- scored triples/sec of the KGE train step + 1-vs-all ranks/sec (BASELINE.json metric) on N MI355X.

Workload (BASELINE.json configs[1]): FB15k-237-shaped DistMult d=200, B=2048 positives, 100 negatives ->
m = 204 800 rows per input array, 409 600 scored triples per step; N=14 541 entities, R=237 relations;
synthetic uniform-id triples (the train split is a missing blob in the reference), Glorot-uniform tables from
RandomState(1234); pointwize_hinge (margin 1) + Keras-semantics Adam.  A step = score(pos) + score(neg) + loss +
gradients + optimizer update on one batch resident in HBM.  By default (--sampler device) every batch is ALSO drawn inside
the timed region by the bit-exact MT19937 HIP sampler (negative sampling + oversampling, the reference's exact draws) on a
side stream, one batch ahead of training; --sampler pre uses batches built on the host before timing.

--check (on by default): before the timed region the SAME code path (TrainPipeline, device sampler, counted plans, hipGraphs,
full batch size) runs 3 steps on a fresh model and is compared with the oracle -- drawn batches bit for bit against the host
generator, loss and both tables after every step against oracle/kge_oracle_c.c -- and the line carries "checked" / "max_abs_err".

N > 1: one process per GPU (torch.distributed; backend nccl = RCCL).  `python bench.py --gpus N` starts the N ranks itself when
it is not already running under torch.distributed.run.  Training: data-parallel replicas, the dense gradient tables are
all-reduced (mean) between the gradient half and the optimizer half of every step; value = triples scored by all ranks /
max-over-ranks wall time (weak scaling).  Evaluation: the YAGO3-10 shape (BASELINE configs[4]; N=123 182, d=200, 10 000
queries) with the entity table row-sharded, in both schedules of SURVEY 8(e): (i) all-gather the table, split the queries;
(ii) resident shards + kge_rank_counts over the rank's own rows + all-reduce of the int32 counts.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(N=14541, R=237, d=200, B=2048, ns=100, n_train=272115)
YAGO = dict(N=123182, R=37, d=200, n_test=5000)   # BASELINE configs[4]: YAGO3-10 DistMult d=200, 5000 test triples x head/tail
A_FWD = 4 * (2 * CFG["d"] + CFG["d"]) + 12 + 4   # two entity rows, one relation row, 3 ids, 1 score
A_BWD = 4 * (2 * CFG["d"] + CFG["d"])            # three gradient rows written once
A_TRAIN = A_FWD + A_BWD                          # 4816 B per scored triple at d=200
A_OPT = 32 * (CFG["N"] * CFG["d"] + CFG["R"] * CFG["d"])  # TF-Adam: read g,w,m,v; write w,m,v; re-zero g (94.6 MB per step)
HBM_PEAK = 8.0e12                                # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK = 157.3e12                         # MI355X_MICROARCH.md: dense fp32 matrix peak
METRIC = "scored triples/sec (train, 100 neg) + 1-vs-all ranks/sec, FB15k-237 d=200"  # BASELINE.json, verbatim


# ----------------------------------------------------------------------------------------------------------------- profiles
def pmc_traffic(kernel_label):
    """HBM bytes per launch of the named kernel family from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    produced by tools/pmc.sh; FETCH_SIZE x2 + WRITE_SIZE, see that file's _source).  None when no pass exists."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None, None
    fam = kernel_label.split()[0]
    best = None
    for name, v in d.get("kernels", {}).items():
        if v.get("family") == fam and (best is None or v["hbm_bytes_per_launch"] > best):
            best = v["hbm_bytes_per_launch"]
    return best, d.get("_tag")


def rocprof_avg_ms(kernel_label, pattern="*_serial_kernel_stats.csv"):
    """Average duration of the named kernel family in the newest committed `rocprofv3 --kernel-trace --stats` summary of
    `bench.py --serial` (profiles/*_serial_kernel_stats.csv): every kernel alone, dispatches serialised by the profiler.
    pattern "*_hbm_kernel_stats.csv": the summary of `bench.py --leg hbm` (the table past the Infinity Cache).
    None when no summary is committed."""
    import csv
    import glob
    fam = kernel_label.split()[0]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        for r in csv.DictReader(open(files[-1])):
            if ("kge::" + fam + "<") in r["Name"] or ("kge::" + fam + "(") in r["Name"]:
                return float(r["AverageNs"]) * 1e-6, os.path.basename(files[-1])
    except Exception:
        return None
    return None


# ----------------------------------------------------------------------------------------------------------------- workload
def workload(model_name="DistMult", rank=0, tables=True):
    """The synthetic FB15k-237-shaped workload of one rank, as plain host arrays (shared by the timed run, --check and the
    parity tests): Glorot tables, the rank's training triples, model kwargs.  tables=False (the HBM leg's 8 GB table): the tables are
    drawn on the device by the model's own Glorot initialiser (models.py:87-93,101-105) instead of on the host."""
    rs = np.random.RandomState(1234)

    def glorot(rows, cols):  # Keras GlorotUniform on an Embedding matrix (models.py:87-93): U(-l, l), l = sqrt(6 / (rows + cols))
        lim = np.sqrt(6.0 / (rows + cols))
        return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)
    E0, R0 = (glorot(CFG["N"], CFG["d"]), glorot(CFG["R"], CFG["d"])) if tables else (None, None)
    rs_t = np.random.RandomState(1234 + 7919 * rank)
    train = np.stack([rs_t.randint(0, CFG["N"], CFG["n_train"]), rs_t.randint(0, CFG["R"], CFG["n_train"]),
                      rs_t.randint(0, CFG["N"], CFG["n_train"])], 1).astype(np.int32)
    mkw = dict(norm=2, gamma=12) if model_name == "TransE" else {}
    return dict(E0=E0, R0=R0, train=train, mkw=mkw, model=model_name)


DP_BENCH = 0.0  # dropout rate of the contract line: 0 (duplicate rows fold); the reference's default 0.2 is the value_dp02 leg


def make_model(w, dp=DP_BENCH):
    from kge_keras_amd import models as KM
    kw = dict(w["mkw"])
    if w["E0"] is None:
        kw["seed"] = 1234  # (tables drawn on the device)
    model = getattr(KM, w["model"])(e_dim=CFG["d"], r_dim=CFG["d"], num_entities=CFG["N"], num_relations=CFG["R"], dp=dp,
                                    init_entities=w["E0"], init_relations=w["R0"], loss_function="pointwize_hinge", margin=1, **kw)
    model.compile("adam")
    return model


def make_pipeline(torch, model, w, rank, total_steps, graphs=True, between=None, between_halves=None, comm=None, world=1, row_shard=None):
    """The benched object: TrainPipeline over the device sampler seeded like the host generator of make_batches()."""
    from kge_keras_amd.pipeline import TrainPipeline
    from kge_keras_amd.sampler import NumpyStream, PythonStream, DeviceSampler
    nps = NumpyStream(rank)
    perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))  # epoch permutation (host, once per epoch)
    train_dev, perm_dev = torch.from_numpy(w["train"]).cuda(), torch.from_numpy(perm).cuda()
    dsamp = DeviceSampler(nps, PythonStream(rank))
    pipe = TrainPipeline(model, train_dev, CFG["n_train"], CFG["B"], CFG["ns"], CFG["N"], dsamp, graphs=graphs, between=between, between_halves=between_halves,
                         comm=comm, world=world, row_shard=row_shard)
    nb_epoch = CFG["n_train"] // CFG["B"]
    pipe.load_epoch(perm_dev, [i % nb_epoch for i in range(total_steps)])
    return pipe, dsamp


def host_batches(w, rank, n_batches, folded=False):
    """The same batches from the host generator (csrc/sampler.cpp: the reference's draws bit for bit).  folded=True also
    returns every batch before oversampling (B positives + B*ns generated negatives) and the pick counts of random.choices."""
    from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
    B, ns, N = CFG["B"], CFG["ns"], CFG["N"]
    m = B * ns
    nps, pys = NumpyStream(rank), PythonStream(rank)
    perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))
    nb_epoch = CFG["n_train"] // B
    out = []
    for i in range(n_batches):
        b = i % nb_epoch
        batch = w["train"][perm[b * B:(b + 1) * B]]
        if folded:  # twin streams replay the draws of this batch: utils.py:118-119 (fs, then fo), :136 (positives, then negatives)
            n2, p2 = NumpyStream(), PythonStream()
            n2._buf[:] = nps._buf
            p2._buf[:] = pys._buf
            fs, fo = n2.randint(0, N, m), n2.randint(0, N, m)
            cand = np.concatenate([batch, np.stack([fs, np.repeat(batch[:, 1], ns), fo], 1).astype(np.int32)])
            ip, ineg = p2.choices_idx(B, m), p2.choices_idx(m, m)
            counts = (np.bincount(ip, minlength=B), np.bincount(ineg, minlength=m))
        pos, neg = build_batch(nps, pys, batch, ns, N)
        if folded:
            assert np.array_equal(pos, cand[:B][ip]) and np.array_equal(neg, cand[B:][ineg])
            out.append((pos, neg, cand, counts))
        else:
            out.append((pos, neg))
    return out


def check_benched_path(model_name="DistMult", steps=3, rank=0, fp64=False, tol=2e-6, tol_loss_rel=1e-6):
    """Parity of the benched code path at the benched size: a fresh model driven by TrainPipeline (device sampler, counted
    plans, steps between twin tables) for `steps` steps on the bench workload.  (a) every batch the device drew == the host
    generator's, bit for bit (rows before oversampling and pick counts; tests.py:55-61, utils.py:114-147); (b) loss, entity
    table and relation table after every step against the oracle's C restatement (fp32; fp64=True: also the NumPy fp64 oracle):
    tables within `tol` absolute (2e-6 -- measured: DistMult 1e-7, TransE-L2 with its x / |x| 9.5e-7 --: a whole Adam update is 1e-3, so a wrong
    lr_t, eps placement or a dropped duplicate shows),
    the loss within `tol_loss_rel` relative.  The oracle is the checker only.  Returns a dict for the bench line; raises
    AssertionError on a mismatch."""
    import torch
    import oracle as O
    from oracle import c_port
    from kge_keras_amd import _lib
    w = workload(model_name, rank)
    model = make_model(w)
    pipe, dsamp = make_pipeline(torch, model, w, rank, steps + 2)
    hb = host_batches(w, rank, steps + 2, folded=True)
    B, L = CFG["B"], _lib.CAND_POS_LANES
    kw = w["mkw"]
    ct = c_port.CTrainer(model_name, w["E0"], w["R0"], "adam", "pointwize_hinge", 1.0, kw.get("norm", 1), kw.get("gamma", 12.0))
    if fp64:
        Eo, Ro, oo = w["E0"].astype(np.float64), w["R0"].astype(np.float64), O.AdamTF()
    seen = set()

    def check_batch(k):
        """slot k % 2 holds batch k in folded form"""
        rows, cnt = (x.cpu().numpy() for x in pipe.cand[k % 2])
        _pos, _neg, cand, (cp, cn) = hb[k]
        assert np.array_equal(rows, cand), f"batch {k}: rows before oversampling differ from the host generator"
        assert np.array_equal(cnt[:L * B].reshape(L, B).sum(0), cp) and np.array_equal(cnt[L * B:], cn), f"batch {k}: pick counts differ"
        seen.add(k)

    pipe.start()
    torch.cuda.synchronize()
    assert pipe._native_active and pipe.counted, "the benched path is the natively driven pipeline with counted plans"
    check_batch(0)
    check_batch(1)
    err = dict(loss=0.0, E=0.0, Rel=0.0)
    err64 = dict(loss=0.0, E=0.0, Rel=0.0)
    for t in range(steps):
        pipe.tick()
        torch.cuda.synchronize()
        if t + 2 < len(hb):
            check_batch(t + 2)
        pos, neg = hb[t][0], hb[t][1]
        Lg = float(model._loss[0].item())
        Eg, Rg = model.E.cpu().numpy(), model.Rel.cpu().numpy()
        Lc = ct.step(pos, neg)
        for k, a, b in (("loss", Lg, Lc), ("E", Eg, ct.E), ("Rel", Rg, ct.Rel)):
            e = float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
            err[k] = max(err[k], e)
            lim = tol if k != "loss" else tol_loss_rel * abs(float(b)) + 1e-7
            assert e <= lim, f"step {t}: {k} differs from the C oracle by {e:.3g} (> {lim:.3g})"
        if fp64:
            Lo = O.train_step(model_name, Eo, Ro, oo, pos, neg, "pointwize_hinge", 1.0, 1, kw.get("norm", 1), kw.get("gamma", 12))
            for k, a, b in (("loss", Lg, Lo), ("E", Eg, Eo), ("Rel", Rg, Ro)):
                e = float(np.max(np.abs(np.asarray(a, dtype=np.float64) - b)))
                err64[k] = max(err64[k], e)
                lim = tol if k != "loss" else tol_loss_rel * abs(float(b)) + 1e-7
                assert e <= lim, f"step {t}: {k} differs from the fp64 oracle by {e:.3g} (> {lim:.3g})"
    pipe.finish()
    pipe.close()  # (its streams go away: idle CU-masked queues left behind would oversubscribe the hardware queues of the timed run)
    torch.cuda.synchronize()
    model._raise_on_flag()
    assert int(dsamp.err.item()) == 0, "device sampler: rejection margin ran short"
    # the generator states the device left behind == the host streams after the same batches
    from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
    nps, pys = NumpyStream(rank), PythonStream(rank)
    perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))
    for i in range(steps + 2):
        build_batch(nps, pys, w["train"][perm[i * B:(i + 1) * B]], CFG["ns"], CFG["N"])
    dn, dp_ = dsamp.to_streams()
    assert np.array_equal(dn._buf[:625], nps._buf[:625]) and np.array_equal(dp_._buf[:625], pys._buf[:625]), "generator states diverged"
    res = dict(checked=True, check_steps=steps, check_batches_bit_exact=sorted(seen), max_abs_err=max(err.values()), max_abs_err_by=err,
               check_oracle="oracle/kge_oracle_c.c (fp32 C restatement)" + (" + oracle/kge_oracle.py (NumPy fp64)" if fp64 else ""), check_tol=tol,
               check_tol_loss_rel=tol_loss_rel)
    if fp64:
        res["max_abs_err_fp64"] = max(err64.values())
    return res


# ----------------------------------------------------------------------------------------------------------------- baselines
def cpu_baseline(batches, E0, R0, budget_s=15.0):
    """The oracle's C restatement of the same train step (oracle/kge_oracle_c.c: gather, DistMult score, pointwize_hinge,
    duplicate-summed gradients, Keras-formula Adam; fp32, OpenMP over the host cores) on the same batches, bounded to
    about budget_s seconds."""
    from oracle import c_port
    ct = c_port.CTrainer("DistMult", E0, R0, "adam", "pointwize_hinge", 1.0)
    ct.step(*batches[0])  # warm-up (thread pool, page faults)
    t0, n, steps = time.time(), 0, 0
    while time.time() - t0 < budget_s:
        pos, neg = batches[steps % len(batches)]
        ct.step(pos, neg)
        n += len(pos) + len(neg)
        steps += 1
    dt = time.time() - t0
    return dict(value=n / dt, unit="scored triples/s", cores=c_port.threads(), kind="port",
                sample=f"{steps} full-size steps ({n} scored triples) of the C oracle (oracle/kge_oracle_c.c, fp32, OpenMP x{c_port.threads()}) in {dt:.1f}s")


# ----------------------------------------------------------------------------------------------------------------- evaluation legs
def eval_leg(torch, model, reps=3):
    """1-vs-all ranks/sec on the FB15k-237-shaped tables of the train leg, one GPU: 40 932 queries (20 466 test triples x
    head/tail), filters = 3 random ids per query."""
    from kge_keras_amd import _lib
    rs = np.random.RandomState(99)
    nt = 20466
    test = np.stack([rs.randint(0, CFG["N"], nt), rs.randint(0, CFG["R"], nt), rs.randint(0, CFG["N"], nt)], 1).astype(np.int32)
    fq = np.repeat(np.arange(nt, dtype=np.int32), 3)
    fi = rs.randint(0, CFG["N"], 3 * nt).astype(np.int32)
    keep = (fi != np.repeat(test[:, 2], 3)) & (fi != np.repeat(test[:, 0], 3))
    fq, fi = fq[keep], fi[keep]
    # both sides of every test triple in ONE sweep of the table (kge_rank_counts, side 2: tail queries [0, nt), head queries [nt, 2 nt));
    # the filter index is resident, like the tables
    fq, fi = torch.from_numpy(np.concatenate([fq, fq + nt])).cuda(), torch.from_numpy(np.concatenate([fi, fi])).cuda()
    tq = torch.from_numpy(test).cuda()

    def sweep():
        c = model.rank_counts(tq, 2, fq, fi)
        return c[:nt], c[nt:]

    sweep()
    torch.cuda.synchronize()
    _lib.lib.kge_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        c0, c1 = sweep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.lib.kge_prof_enable(0)
    ms, n = _lib.prof_read("rank")
    sw_ms, sw_n = _lib.prof_read("rank_sweep")
    total_q = 2 * nt
    flops = 2.0 * CFG["d"] * CFG["N"]  # 2*d_e*N per rank
    return dict(ranks_per_sec=total_q * reps / dt, queries=total_q, ms_per_sweep=dt / reps * 1e3,
                mfma_tflops_algorithmic=total_q * reps * flops / dt / 1e12, mfma_peak_tflops=MFMA_F32_PEAK / 1e12,
                mfma_frac=total_q * reps * flops / dt / MFMA_F32_PEAK, rank_call_avg_ms=ms / max(1, n),
                sweep_kernel_avg_ms=sw_ms / max(1, sw_n), sweep_kernel_launches=sw_n, flop_per_launch=total_q * flops,
                sweep="both sides of the 20466 test triples in one launch (40932 queries x 14541 candidates)",
                mean_rank=float((1 + c0.float().mean() + 1 + c1.float().mean()).item() / 2))


def eval_sharded_leg(torch, dist, rank, world, reps=3):
    """The north star's multi-GPU half on the shape it is quoted for (BASELINE configs[4]: YAGO3-10 DistMult d=200): the entity
    table row-sharded across the ranks, 10 000 filtered queries, both schedules of SURVEY 8(e).  Ranks of the two schedules
    must be identical."""
    from kge_keras_amd import models as KM
    from kge_keras_amd import dist as D
    N, R, d, nt = YAGO["N"], YAGO["R"], YAGO["d"], YAGO["n_test"]
    rs = np.random.RandomState(4321)
    lim = np.sqrt(6.0 / (N + d))
    E0 = rs.uniform(-lim, lim, size=(N, d)).astype(np.float32)
    R0 = rs.uniform(-0.4, 0.4, size=(R, d)).astype(np.float32)
    test = np.stack([rs.randint(0, N, nt), rs.randint(0, R, nt), rs.randint(0, N, nt)], 1).astype(np.int32)
    fq_all = np.repeat(np.arange(nt, dtype=np.int32), 3)
    fi_all = rs.randint(0, N, 3 * nt).astype(np.int32)
    keep = (fi_all != np.repeat(test[:, 2], 3)) & (fi_all != np.repeat(test[:, 0], 3))
    fq_all, fi_all = fq_all[keep], fi_all[keep]
    lo, hi = D.shard_range(N, rank, world)
    model = KM.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0)
    shard = model.E[lo:hi].clone()          # what this rank holds at rest
    model.E.zero_()                         # schedule (i) rebuilds the full table from the shards

    # schedule (i): all-gather the table, split the queries, gather the ranks
    idx = D.split_queries(nt, rank, world).numpy()
    mine = torch.from_numpy(test[idx]).cuda()
    sel = np.isin(fq_all, idx)
    fq1 = np.searchsorted(idx, fq_all[sel]).astype(np.int32)
    # (both sides of a rank's test triples in one sweep: kge_rank_counts side 2, head queries numbered behind the tail queries)
    fq1 = torch.from_numpy(np.concatenate([fq1, fq1 + len(idx)])).cuda()
    fi1 = torch.from_numpy(np.concatenate([fi_all[sel], fi_all[sel]])).cuda()
    ag = [0.0]

    def sched1():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.E.copy_(D.allgather_rows(shard, N))
        e1.record()
        c = 1 + model.rank_counts(mine, 2, fq1, fi1).long()
        out = D.gather_ranks(c[:len(idx)], nt), D.gather_ranks(c[len(idx):], nt)
        torch.cuda.synchronize()
        ag[0] += e0.elapsed_time(e1)
        return out

    # schedule (ii): resident shards, referenced rows exchanged, counts all-reduced
    fq2, fi2 = torch.from_numpy(np.concatenate([fq_all, fq_all + nt])).cuda(), torch.from_numpy(np.concatenate([fi_all, fi_all])).cuda()

    # (what depends on the test set and the shard boundaries only is prepared once: dist.ResidentShardPlan)
    plan2 = D.ResidentShardPlan(test, shard.shape[0], lo, shard.device, n_entities=N)

    def sched2():
        return plan2.run(lambda ref, local, side: model.rank_counts_shard(ref, local, side, fq2, fi2, shard, lo), shard, both=True)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ag[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return out, dt / reps

    (t1, h1), dt1 = timed(sched1)
    ag_ms = ag[0] / reps
    (t2, h2), dt2 = timed(sched2)
    same = bool(torch.equal(t1, t2) and torch.equal(h1, h2))
    n_ref = int(len(D.referenced_rows(test)))
    per = (N + world - 1) // world
    return dict(workload=f"YAGO3-10 shape: DistMult d={d}, N={N}, R={R}, {2 * nt} filtered queries, entity table row-sharded over {world} rank(s)",
                ranks_identical_between_schedules=same, mean_rank=float((t1.double().mean() + h1.double().mean()).item() / 2),
                schedule_i=dict(what="all-gather E, split queries, all-gather ranks", ranks_per_sec=2 * nt / dt1, ms=dt1 * 1e3,
                                allgather_ms=ag_ms, allgather_bytes_per_rank=per * d * 4, allgather_bytes_total=per * world * d * 4),
                schedule_ii=dict(what="resident shards: exchange referenced rows, kge_rank_counts over own rows, all-reduce int32 counts",
                                 ranks_per_sec=2 * nt / dt2, ms=dt2 * 1e3, exchanged_row_bytes=n_ref * d * 4, allreduce_count_bytes=2 * nt * 4))



# ----------------------------------------------------------------------------------------------------------------- HBM leg
HBM_ENTITIES = 10_000_000  # 8 GB entity table (+ 2 x 8 GB Adam moments + its 8 GB twin): 31x the 256 MB Infinity Cache


def hbm_leg(torch, steps, n_entities=HBM_ENTITIES, serial_only=False):
    """roofline_hbm: the SAME benched pipeline (native driver, device sampler, counted plans, steps between twin tables, same batch
    shape B = 2048 x 100 negatives, DistMult d = 200, pointwize_hinge + Keras Adam) on an entity table that cannot sit in the 256 MB
    Infinity Cache, so that step1_kernel's bytes come from and go to DRAM.  Keras' Adam updates EVERY row of the table every step
    (dense moment decay + dense update: tests.py:86 'adam' [TF-knowledge]), so on a table far larger than the batch
    the launch is the optimizer's streams: read w, m, v + write w, m, v = 24 N d bytes, plus 16 d bytes per batch row for the gathers.
    Kernel time: library hipEvents around step1_kernel in a serial pass (every kernel alone on one stream), like the roofline leg."""
    from kge_keras_amd import _lib
    saved = CFG["N"]
    CFG["N"] = int(n_entities)
    try:
        w = workload("DistMult", 0, tables=False)
        model = make_model(w)
        k = max(4, min(int(steps), 40))
        pipe, dsamp = make_pipeline(torch, model, w, 0, 3 * k + 20)
        ms_step = None
        if not serial_only:
            pipe.start()
            pipe.run(4, join=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run(k, join=False)
            torch.cuda.synchronize()
            ms_step = (time.perf_counter() - t0) / k * 1e3
            pipe.finish()
        pipe.use_graphs, pipe.serial = False, True
        nb = CFG["n_train"] // CFG["B"]
        pipe.load_epoch(pipe.index, [i % nb for i in range(k + 2)])
        pipe.start()
        torch.cuda.synchronize()
        _lib.lib.kge_prof_enable(1)
        pipe.run(k)
        torch.cuda.synchronize()
        _lib.lib.kge_prof_enable(0)
        pipe.finish()
        k_ms, k_n = _lib.prof_read("train_grads")
        for lab in ("fwd", "opt", "index", "sampler", "dedup"):
            _lib.prof_read(lab)
        L = _lib.CAND_POS_LANES * CFG["B"]
        rows = float(np.mean([int((c[:L].view(_lib.CAND_POS_LANES, CFG["B"]).sum(0) > 0).sum() + (c[L:] > 0).sum()) for _r, c in pipe.cand]))
        model._raise_on_flag()
        assert int(dsamp.err.item()) == 0, "device sampler: rejection margin ran short"
        loss = float(model._loss.item())
        assert np.isfinite(loss), "HBM leg: loss is not finite"
        pipe.close()
        d, N = CFG["d"], CFG["N"]
        algo = rows * 16 * d + 24.0 * N * d
        k_live = k_ms / max(1, k_n) * 1e-3
        rp = rocprof_avg_ms("step1_kernel", "*_hbm_kernel_stats.csv")
        # Live hipEvents of THIS run price the leg; the committed rocprofv3 average of the same command rides along with its own fraction.
        # (The cache-resident roofline object takes the slower of the two, which agree within a few per cent there; on this 48 GB stream
        #  the boxes of the pool differ by up to 15 % -- 8.2 .. 9.5 ms -- so a committed figure from another box says little about this one.)
        k_avg = k_live
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_hbm.json")))
            if int(tj.get("n_entities", 0)) == N:
                traffic = tj["step1_kernel_bytes_per_launch"]
        except Exception:
            pass
        del pipe, model
        torch.cuda.empty_cache()
        return {"bound": "hbm", "kernel": "step1_kernel (relation-grouped forward + per-entity-row gather, own scores, gradient row, optimizer: one launch)",
                "workload": f"the benched pipeline and batch shape (DistMult d={d}, B={CFG['B']}, {CFG['ns']} negatives, pointwize_hinge + Adam) on {N} entities: "
                            f"a {N * d * 4 / 1e9:.1f} GB table + 2 moment tables + its twin, {N * d * 4 / 2.68e8:.0f}x the 256 MB Infinity Cache",
                "n_entities": N, "table_bytes": N * d * 4,
                "avg_kernel_ms": k_avg * 1e3, "avg_kernel_ms_live_hipevents": k_live * 1e3, "launches": k_n,
                "avg_kernel_ms_rocprofv3": rp[0] if rp else None, "rocprofv3_source": f"profiles/{rp[1]}" if rp else None,
                "frac_at_committed_rocprofv3_avg": (algo / (rp[0] * 1e-3) / HBM_PEAK) if rp else None,
                "algorithmic_bytes_per_launch": algo,
                "algorithmic_bytes_note": "24 N d (Keras Adam on every entity row: read w, m, v; write w, m, v) + 16 d per batch row (two entity rows gathered in the "
                                          "forward role, one per list entry = two per row in the entity role); relation rows (L2 resident) and index lists not counted",
                "rows_per_launch": rows,
                "achieved": algo / k_avg / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": algo / k_avg / HBM_PEAK,
                "frac_of_measured_copy_rate": algo / k_avg / 6.29e12,  # MI355X_MICROARCH.md: 6.29 TB/s float4 copy = 0.79 of the 8 TB/s spec
                "traffic": traffic,
                "traffic_source": "committed rocprofv3 --pmc passes of `bench.py --leg hbm` (profiles/pmc_traffic_hbm.json; FETCH_SIZE x 2 + WRITE_SIZE), not this run" if traffic else None,
                "ms_per_step": ms_step,
                "scored_triples_per_sec": (2 * CFG["B"] * CFG["ns"] / (ms_step * 1e-3)) if ms_step else None,
                "final_loss": loss}
    finally:
        CFG["N"] = saved


def dp02_leg(torch, w, steps):
    """value_dp02: the same workload with the reference's DEFAULT dropout rate (models.py:22 `dp=0.2`): every oversampled row has its
    own mask, so nothing folds -- the native pipeline runs in arrays mode (draw -> plan on the 409 600 rows as they are -> three-launch
    step with the mask seed of its number)."""
    model = make_model(w, dp=0.2)
    k = max(10, min(int(steps), 200))
    pipe, dsamp = make_pipeline(torch, model, w, 0, 2 * k + 30)
    pipe.start()
    pipe.run(10, join=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(k, join=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.finish()
    model._raise_on_flag()
    loss = float(model._loss.item())
    native = bool(pipe._native_active)
    pipe.close()
    return {"value_dp02": 2 * CFG["B"] * CFG["ns"] * k / dt, "ms_per_step_dp02": dt / k * 1e3, "steps_dp02": k, "dp": 0.2,
            "note_dp02": "reference default dropout 0.2: a mask per oversampled row, no duplicate folding, 409600 rows per step through the "
                         "three-launch step (native loop, arrays mode)" + ("" if native else " [Python loop]"), "final_loss_dp02": loss}

# ----------------------------------------------------------------------------------------------------------------- launcher
def launch_children(argv, n, backend):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as child processes (this parent makes no GPU
    call and never execs), forward their output, exit with the worst return code.  Rank 0 prints the JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the parity check of the benched path against the oracle (outside the timed region)")
    ap.add_argument("--no-sharded-eval", action="store_true", help="skip the YAGO3-10-shape sharded evaluation leg (profiling passes: keeps the sweep kernel's statistics on one shape)")
    ap.add_argument("--dp-overlap", choices=["auto", "on", "off"], default="auto", help="N > 1: start the all-reduce of the entity gradient table behind the entity half of the gradients, beside the relation half (auto: nccl backend only)")
    ap.add_argument("--dp-native", choices=["auto", "on", "off"], default="auto", help="N > 1: the data-parallel step inside the native pipeline driver, all-reduces on an RCCL communicator of the library's own (auto: nccl backend, falls back to the Python loop over torch.distributed when the communicator cannot be made)")
    ap.add_argument("--sampler", default="device", choices=["device", "pre"],
                    help="device: every step's batch is drawn by the bit-exact MT19937 HIP sampler on a side stream, one batch ahead "
                         "(inside the timed region); pre: batches pre-built on the host before timing")
    ap.add_argument("--plan", default="ahead", choices=["ahead", "step"],
                    help="ahead: batch plans (ids-only half of the step) built on their own stream ahead of the step; step: inside the step")
    ap.add_argument("--settle", type=int, default=0, help="0 (default): the K timed steps follow the W warm-up steps directly.  > 0: upper bound of extra "
                    "untimed steps run until the step time is stable, then K steps are timed again (the first K are kept as value_first_steps)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "stream"],
                    help="graph / auto: the TrainPipeline object (every stage one hipGraph launch per step on one GPU; kernel-by-kernel "
                         "launches of the same stages under torch.distributed); stream: the plain double-buffered loop of this file")
    ap.add_argument("--serial", action="store_true",
                    help="profiling aid: run the pipeline stages one after the other on one stream, kernel by kernel, in every phase "
                         "(what the roofline leg does), so that a rocprofv3 kernel trace of the run times every kernel alone")
    ap.add_argument("--model", default="DistMult", choices=["DistMult", "TransE"],
                    help="DistMult: the BASELINE configs[1] workload (the contract line).  TransE: configs[2] (L2, gamma 12) through the same "
                         "pipeline, for comparison only (no CPU baseline; config.workload names it)")
    ap.add_argument("--train", default="replicas", choices=["replicas", "rows"],
                    help="N > 1, the training half.  replicas (default): data-parallel replicas, every rank its own batch, dense gradient all-reduce "
                         "(weak scaling: value = N batches' rows / time).  rows: the tables row-sharded, OWNER COMPUTES (dist.RowShardTrainer): every rank "
                         "draws the SAME batch, updates its own entity / relation rows with its share of the Adam moments, then the updated rows are "
                         "all-gathered -- the reference's one-batch-per-step semantics at N GPUs (strong scaling: value = one batch's rows / time)")
    ap.add_argument("--leg", default="all", choices=["all", "hbm"], help="hbm: only the HBM leg (the benched pipeline on the table past the Infinity Cache), "
                    "kernel by kernel on one stream -- the command whose `rocprofv3 --kernel-trace --stats` summary is committed as profiles/*_hbm_kernel_stats.csv")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip roofline_hbm (the same pipeline on a table that does not fit the Infinity Cache)")
    ap.add_argument("--hbm-entities", type=int, default=HBM_ENTITIES, help="entities of the HBM leg's table (default 10 M: 8 GB + moments + twin = 32 GB)")
    ap.add_argument("--no-dp02", action="store_true", help="skip value_dp02 (the workload with the reference's default dropout 0.2)")
    ap.add_argument("--probe", action="store_true", help="timing experiments with KGE_PIPE_STAGES: do not insist on clean error flags")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed code paths even with WORLD_SIZE=1 (rehearsal on one GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(sys.argv[1:], args.gpus, args.backend))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # --force-dist: rehearse the torch.distributed code paths (RCCL process group, all-reduce between the halves of a step,
    # sharded eval) with a world of one process, on a box that has a single GPU
    dist_on = world > 1 or args.force_dist
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.force_dist and world == 1:
            for k, v in (("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from kge_keras_amd import _lib
    from kge_keras_amd import dist as D

    if args.leg == "hbm":  # (profiling aid: one shape, every kernel alone)
        print(json.dumps({"roofline_hbm": hbm_leg(torch, args.steps, args.hbm_entities, serial_only=args.serial)}), flush=True)
        return

    checked = None
    if not args.no_check and rank == 0:
        checked = check_benched_path(args.model, steps=3, rank=0)
    if dist_on:
        dist.barrier()

    rows_mode = dist_on and args.train == "rows"
    data_rank = 0 if rows_mode else rank  # (owner-computes: every rank holds the same triples and draws the same batches from the same seeds)
    w = workload(args.model, data_rank)
    E0, R0 = w["E0"], w["R0"]
    model = make_model(w)
    n_pool = 8
    hb = host_batches(w, data_rank, n_pool)
    dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda()) for p, n in hb]
    m = CFG["B"] * CFG["ns"]
    per_step = 2 * m

    # plain double-buffered loop (--launch stream / --sampler pre / --plan step): batch i+2 is drawn on a side stream while batch i trains
    from kge_keras_amd.sampler import NumpyStream, PythonStream, DeviceSampler
    main_stream = torch.cuda.current_stream()
    plan_ahead = args.plan == "ahead" and model.plan_supported()
    launch = args.launch if args.launch != "auto" else "graph"
    use_pipe = args.sampler == "device" and plan_ahead and launch == "graph"
    pipe = dsamp = None
    comm = comm_destroy = dp_native_note = None
    if use_pipe:
        # gradient tables and the loss word are one flat buffer.  The entity half of the gradients needs nothing from the forward
        # (kge_train_grads_planned_parts): the all-reduce of its table -- 11.6 of the 11.8 MB -- starts right behind it and runs while
        # the relation half (forward, loss, relation rows) still computes; the rest is reduced behind that.  Where the halves do not
        # come apart: ONE all-reduce of the whole buffer per step.
        # (--dp-overlap auto: with the nccl / RCCL backend only -- gloo runs an async collective on CUDA tensors through the host at
        # 90 ms per step in the two-rank rehearsal, against 3.3 ms for the single all-reduce)
        overlap = args.dp_overlap == "on" or (args.dp_overlap == "auto" and args.backend == "nccl")
        row_trainer = D.RowShardTrainer(model) if rows_mode else None
        pending = []

        def between_halves():
            pending.append(dist.all_reduce(model._gall[:model._gE.numel()], op=dist.ReduceOp.SUM, async_op=True))

        def between_fn():
            if pending:
                dist.all_reduce(model._gall[model._gE.numel():], op=dist.ReduceOp.SUM)
                pending.pop().wait()
            else:
                dist.all_reduce(model._gall, op=dist.ReduceOp.SUM)
            return 1.0 / world
        between = between_fn if dist_on and not rows_mode else None
        # the data-parallel step inside the native driver (kge_pipeline_desc.comm): an RCCL communicator of the library's own
        if dist_on and not rows_mode and (args.dp_native == "on" or (args.dp_native == "auto" and args.backend == "nccl")):
            try:
                comm, comm_destroy = D.make_rccl_comm()
            except Exception as e:  # (e.g. the gloo rehearsal with two ranks on one GPU: RCCL refuses duplicate devices)
                if args.dp_native == "on":
                    raise
                dp_native_note = f"native data-parallel step not available ({e!r}): Python loop over torch.distributed"
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks or none
            if int(ok.item()) == 0 and comm is not None:
                comm_destroy()
                comm = None
        if comm is not None:
            between = None
        total = args.warmup + args.settle + 3 * args.steps + 8
        pipe, dsamp = make_pipeline(torch, model, w, data_rank, total, graphs=True, between=between,
                                    between_halves=between_halves if dist_on and overlap and comm is None and not rows_mode else None,
                                    comm=comm, world=world, row_shard=row_trainer)
        if args.serial:
            pipe.use_graphs, pipe.serial = False, True
        pipe.start()
        step = lambda i: pipe.tick()  # noqa: E731
        steps_n = lambda n: pipe.run(n, join=False)  # noqa: E731  n steps in ONE library call (kge_pipeline_run): the tick loop is native; every timed region ends with a device synchronisation
    else:
        side = torch.cuda.Stream()
        nps = NumpyStream(rank)
        perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))
        train_dev, perm_dev = torch.from_numpy(w["train"]).cuda(), torch.from_numpy(perm).cuda()
        dsamp = DeviceSampler(nps, PythonStream(rank))
        bufs = [tuple(torch.empty((m, 3), dtype=torch.int32, device="cuda") for _ in range(2)) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]     # batch arrays drawn
        freed = [torch.cuda.Event() for _ in range(2)]     # batch arrays no longer needed
        planned = [torch.cuda.Event() for _ in range(2)]   # batch plan built
        pfreed = [torch.cuda.Event() for _ in range(2)]    # batch plan consumed
        nb_epoch = CFG["n_train"] // CFG["B"]
        pstream = torch.cuda.Stream() if plan_ahead else None
        plans = [None, None]

        def produce(i):
            j = i % 2
            if args.sampler == "device":
                with torch.cuda.stream(side):
                    side.wait_event(freed[j])
                    b = i % nb_epoch
                    batch = train_dev.index_select(0, perm_dev[b * CFG["B"]:(b + 1) * CFG["B"]])
                    dsamp.build_batch(batch, CFG["ns"], CFG["N"], *bufs[j])
                    ready[j].record(side)
            if plan_ahead:
                with torch.cuda.stream(pstream):
                    if args.sampler == "device":
                        pstream.wait_event(ready[j])
                    pstream.wait_event(pfreed[j])
                    plans[j] = model.plan_batch(list(bufs[j]) if args.sampler == "device" else list(dev[i % n_pool]), out=plans[j])
                    planned[j].record(pstream)
                    freed[j].record(pstream)

        def step(i):  # kernel-by-kernel launches
            j = i % 2
            if plan_ahead:
                main_stream.wait_event(planned[j])
                data = plans[j]
            elif args.sampler == "device":
                main_stream.wait_event(ready[j])
                data = list(bufs[j])
            else:
                data = list(dev[i % n_pool])
            if not dist_on:
                model.train_step(data)
            else:  # data-parallel: gradient half, ONE all-reduce of the flat gradient buffer over RCCL, optimizer half
                D.dp_train_step(lambda: model.train_grads(data), model.apply_grads, [model._gall], None)
            if plan_ahead:
                pfreed[j].record(main_stream)
            elif args.sampler == "device":
                freed[j].record(main_stream)
            if args.sampler == "device" or plan_ahead:
                produce(i + 2)

        if args.sampler == "device" or plan_ahead:
            side.wait_stream(main_stream)
            if pstream is not None:
                pstream.wait_stream(main_stream)
            produce(0)
            produce(1)

    if pipe is None:
        nxt = [0]

        def steps_n(n):
            for _ in range(n):
                step(nxt[0])
                nxt[0] += 1

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        fence()
        t0 = time.perf_counter()
        steps_n(n)
        t_issue = time.perf_counter() - t0  # host time to enqueue the steps (the GPU runs behind it)
        fence()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, t_issue

    steps_n(args.warmup)
    # the K steps right behind the W warm-up steps, exactly as asked for (on a fresh box the host is still paging the image in)
    if pipe is not None:
        pipe.host_stats(reset=True)
    elif args.settle == 0:
        _lib.lib.kge_prof_enable(1)  # hipEvents around the dominant kernels, recorded on the launch stream
    dt_first, t_issue_first = timed(args.steps)
    # optional settle phase (--settle N > 0; off by default: with the native step loop the first K steps run at the steady rate): keep
    # stepping in blocks of 20 until a block is within 15 % of the best one seen, for at most 1 s (one GPU) or a fixed count
    settle, best, t_settle = 0, None, time.perf_counter()
    while settle + 20 <= args.settle:
        el = time.perf_counter() - t_settle
        if dist_on and settle >= min(args.settle, 400):
            break
        if not dist_on and el > 1.0:
            break
        tb = time.perf_counter()
        steps_n(20)
        fence()
        tb = time.perf_counter() - tb
        settle += 20
        if not dist_on and best is not None and el >= 0.2 and tb <= 1.15 * best:
            break
        best = tb if best is None else min(best, tb)
    if pipe is None and settle > 0:
        _lib.lib.kge_prof_enable(1)
    if settle > 0:
        if pipe is not None:
            pipe.host_stats(reset=True)
        dt, t_issue = timed(args.steps)
    else:
        dt, t_issue = dt_first, t_issue_first
    host_stats = pipe.host_stats() if pipe is not None else None
    if pipe is not None:
        # kernel durations for the roofline object: K more steps launched kernel by kernel on ONE stream (kernels of concurrent stages
        # stretch each other: the roofline is about the kernel, so it is timed alone, like rocprofv3's kernel trace does), hipEvents
        # recorded by the library on that stream.  A fresh epoch of the same pipeline object started in serial mode -- the timed epoch
        # is closed first, nothing is handed over between the native driver and the Python loop.
        try:
            pipe.finish()
        except RuntimeError:
            if not args.probe:
                raise
        pipe.use_graphs = False
        pipe.serial = True
        nb_epoch = CFG["n_train"] // CFG["B"]
        pipe.load_epoch(pipe.index, [i % nb_epoch for i in range(args.steps + 2)])
        pipe.start()
        fence()
        _lib.lib.kge_prof_enable(1)
        steps_n(args.steps)
        fence()
        try:
            pipe.finish()
        except RuntimeError:
            if not args.probe:
                raise
    _lib.lib.kge_prof_enable(0)
    if not args.probe:
        model._raise_on_flag()
    if rows_mode:
        loss = float(row_trainer.loss.item())  # (partial losses summed by the exchange: the loss of the ONE batch)
    else:
        loss = float(model._loss.item()) / (world if dist_on else 1)  # summed over ranks by the flat all-reduce
    s_ms, s_n = _lib.prof_read("sampler")
    i_ms, i_n = _lib.prof_read("index")
    assert args.probe or int(dsamp.err.item()) == 0, "device sampler: rejection margin ran short"
    k_ms, k_n = _lib.prof_read("train_grads")
    f_ms, f_n = _lib.prof_read("fwd")
    o_ms, o_n = _lib.prof_read("opt")
    d_ms, d_n = _lib.prof_read("dedup")

    # rows a launch of the gather kernels really processes: with duplicate-row folding on (default; KGE_DEDUP=0 turns it off) the
    # distinct rows of a batch, counted here (outside the timed region) on the batches the run used
    def distinct(a):
        key = (a[:, 0].long() * CFG["R"] + a[:, 1].long()) * CFG["N"] + a[:, 2].long()
        return int(torch.unique(key).numel())
    folded = d_n > 0
    if pipe is not None and pipe.counted:  # rows with a pick count (the sampler's folded form of the last two batches)
        L = _lib.CAND_POS_LANES * CFG["B"]
        rows_launch = float(np.mean([int((c[:L].view(_lib.CAND_POS_LANES, CFG["B"]).sum(0) > 0).sum() + (c[L:] > 0).sum()) for _r, c in pipe.cand]))
        folded = True
    else:
        used = list(pipe.bufs) if pipe is not None else (list(bufs) if args.sampler == "device" else dev)
        rows_launch = float(np.mean([distinct(p) + distinct(n) for p, n in used])) if folded else float(per_step)

    ev = eval_leg(torch, model) if rank == 0 or not dist_on else None
    if dist_on:
        dist.barrier()
    # (also at N = 1 -- one shard, no exchange: the baseline the N-GPU lines' eval_sharded rates are compared with)
    ev_sh = eval_sharded_leg(torch, dist, rank, world) if not args.no_sharded_eval else None

    # one GPU only: the workload with the reference's default dropout, and the same pipeline on a table past the Infinity Cache
    dp02 = hbm = None
    if world == 1 and not dist_on and pipe is not None and args.model == "DistMult":
        pipe.close()  # (its CU-masked queues go away before the next pipeline makes its own)
        # (the large-table leg first, on an allocator that holds nothing else: r04_a measured its kernel 7 % slower behind the dropout leg
        #  than alone on a box of the same speed, r4l -- without the dropout leg in front -- the same as alone)
        if not args.no_hbm_leg:
            torch.cuda.empty_cache()
            hbm = hbm_leg(torch, args.steps, args.hbm_entities)
        if not args.no_dp02:
            dp02 = dp02_leg(torch, w, args.steps)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        work_ranks = 1 if rows_mode else world  # (rows: all ranks work on ONE batch per step)
        value = per_step * work_ranks * args.steps / dt
        # pull path: two gather kernels (fwd_rel_kernel: A_FWD + the relation-gradient row per triple; pull_kernel: the two entity
        # gradient rows per triple); push path: one fused kernel carrying A_TRAIN.
        kernels = {}
        variant = 1 if os.environ.get("KGE_PULL_VARIANT", "2") == "1" else 2
        twin = pipe is not None and (getattr(pipe, "_twins", None) is not None or rows_mode)  # the step between twin tables (kge_train_step_swap[_window])
        # Bytes a launch has to fetch from beyond the L2s, on the rows really processed (the 190 KB relation table is L2 resident and
        # is NOT counted; SURVEY 8(d)'s 4816 B per triple would also count three gradient rows, which this step keeps in registers):
        #   forward role   two entity rows per row                                  8 d   B per row
        #   entity role    one entity row per list entry, two entries per row       8 d   B per row
        #                  optimizer on every entity row: read w, m, v; write w, m, v 24 N d B per launch
        adam_e = 24 * CFG["N"] * CFG["d"] // work_ranks  # (rows mode: a rank streams the moments and rows of its window only)
        if twin and k_n and not f_n:
            kernels["step1_kernel (relation-grouped forward + per-entity-row gather, own scores, gradient row, optimizer: one launch)"] = (
                k_ms / k_n * 1e-3, rows_launch * 16 * CFG["d"] + adam_e)
        elif twin and k_n:
            kernels["fwd_rel_kernel (gather+score+loss+relation gradient)"] = (f_ms / f_n * 1e-3, rows_launch * (A_FWD + 4 * CFG["d"]))
            kernels["pull_self_kernel (per-entity-row gather, own scores, gradient row, optimizer)"] = (k_ms / k_n * 1e-3, rows_launch * 8 * CFG["d"] + adam_e)
        elif f_n and variant == 2:
            kernels["fwd_rel_kernel (gather+score+loss+relation gradient)"] = (f_ms / f_n * 1e-3, rows_launch * (A_FWD + 4 * CFG["d"]))
            kernels["pull_kernel (per-entity-row gather-reduce of gradients)"] = (k_ms / max(1, k_n) * 1e-3, rows_launch * 8 * CFG["d"])
        elif f_n:
            kernels["fwd_kernel (gather+score+loss)"] = (f_ms / f_n * 1e-3, per_step * A_FWD)
            kernels["pull_kernel (per-row gather-reduce of gradients)"] = (k_ms / max(1, k_n) * 1e-3, per_step * A_BWD)
        else:
            kernels["push_kernel (gather+score+loss+atomic grad)"] = (k_ms / max(1, k_n) * 1e-3, per_step * A_TRAIN)
        # the roofline object describes the kernel that moves the most bytes on the memory side per launch (committed PMC passes),
        # falling back to the longest one when no PMC pass exists
        dom = max(kernels, key=lambda k: (pmc_traffic(k)[0] or 0.0, kernels[k][0]))
        k_live, k_bytes = kernels[dom]
        rp = rocprof_avg_ms(dom)
        # The duration the roofline fraction is priced on: the SLOWER of (a) the live hipEvent average of this run (kernel by kernel on
        # one stream, right behind its producer kernels: part of its rows still in L2) and (b) the committed rocprofv3 --kernel-trace
        # average of the same command (profiles/: dispatches serialised by the profiler) -- so the headline fraction always follows
        # from profiles/ or is lower.
        k_avg = max(k_live, rp[0] * 1e-3) if rp is not None else k_live
        traffic, traffic_tag = pmc_traffic(dom)
        # Bytes the fraction is priced on: the algorithmic bytes of the launch (what it must fetch / store with perfect reuse inside the
        # launch) or, when a PMC pass of this kernel is committed and reads LOWER, the bytes that really left the L2s.  At this table
        # size (11.6 MB + moments: Infinity-Cache resident, a third of it per XCD L2) part of the algorithmic gathers are L2 hits, so
        # the algorithmic rate alone would overstate what the memory side delivered; both rates are in the line.
        priced_bytes = min(k_bytes, traffic) if traffic else k_bytes
        achieved = priced_bytes / k_avg if k_avg else 0.0
        res = {
            "metric": METRIC, "value": value, "unit": "scored triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if rows_mode else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FB15k-237 {args.model if args.model == 'DistMult' else 'TransE-L2 (gamma 12)'} d=200, B=2048, 100 negatives (409600 scored triples/step), pointwize_hinge + Adam(TF semantics), "
                                   f"dp={DP_BENCH:g} (the reference's default is 0.2: value_dp02); oversample_data draws with replacement, so {int(round(rows_launch))} of the 409600 rows of a step are distinct "
                                   "and the step folds the duplicates (same sums as TensorFlow's IndexedSlices accumulation): value counts all 409600, value_distinct the distinct ones",
                       "dp": DP_BENCH, "distinct_rows_per_step": rows_launch,
                       "N": CFG["N"], "R": CFG["R"], "d": CFG["d"], "B": CFG["B"], "ns": CFG["ns"],
                       "parallelism": (f"rows{world} (tables row-sharded, owner computes: same batch on every rank, updated rows all-gathered; "
                                       f"{row_trainer.tables.exchanged_bytes()} B received per rank and step)" if rows_mode else f"dp{world}") if dist_on else "single",
                       "sampler": ("bit-exact MT19937 HIP sampler inside the timed region (generator words ahead in rings, batches drawn two steps ahead on their own stream)" if args.sampler == "device"
                                   else "bit-exact MT19937 batches pre-built on host, resident in HBM"),
                       "final_loss": loss},
            # value counts every row of both oversampled arrays (what the reference's step scores); the step folds duplicate rows, so
            # the gather kernels touch rows_per_launch distinct rows: value_distinct is the rate on those
            "value_distinct": rows_launch * work_ranks * args.steps / dt,
            "value_ranks_per_sec": ev["ranks_per_sec"] if ev else None,
            "settle_steps": settle,  # 0: the K timed steps follow the W warm-up steps directly
            **({"value_first_steps": per_step * world * args.steps / dt_first, "ms_per_step_first_steps": dt_first / args.steps * 1e3} if settle > 0 else {}),
            # `roofline` is the contract's object for the dominant kernel AT THE BASELINE SHAPE -- where tables + moments (47 MB) live in L2 + the
            # Infinity Cache, so its bytes are fabric-side bytes, not DRAM bytes: bound says so.  The DRAM-bound figure is `roofline_hbm` below.
            "roofline": {"bound": "infinity-cache/fabric", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_source": (f"committed rocprofv3 --pmc passes ({traffic_tag or 'profiles/pmc_traffic.json'}; tools/pmc.sh), not this run" if traffic else None),
                         "priced_on": ("measured traffic beyond the L2s (committed PMC pass; lower than the algorithmic bytes: part of the gathers hit in L2)"
                                       if traffic and traffic < k_bytes else "algorithmic bytes of the launch"),
                         "algorithmic_bytes_per_launch": k_bytes, "algorithmic_GBps": k_bytes / k_avg / 1e9 if k_avg else None,
                         "algorithmic_frac": k_bytes / k_avg / HBM_PEAK if k_avg else None,
                         "residency_note": "tables + Adam moments (47 MB) fit the 256 MB Infinity Cache: `traffic` counts fabric-side requests of the L2s (Infinity-Cache hits included), not DRAM reads; "
                                           "`peak` is the 8 TB/s HBM figure the contract asks for -- the Infinity Cache's gather ceiling for rows of this size is ~8.6 TB/s (MI355X_MICROARCH.md, Indexed rows)",
                         # what DRAM itself must move per launch at this shape even with perfect caching between launches: every entity row and both of
                         # its moments read and written once (Keras Adam is dense) + both relation-table copies + the plan's list entries read once
                         "compulsory_dram_bytes_per_launch": 24.0 * CFG["N"] * CFG["d"] + 24.0 * CFG["R"] * CFG["d"] + rows_launch * 48.0,
                         "compulsory_dram_frac": ((24.0 * CFG["N"] * CFG["d"] + 24.0 * CFG["R"] * CFG["d"] + rows_launch * 48.0) / k_avg / HBM_PEAK) if k_avg else None,
                         # SURVEY 8(d)'s own formula on the distinct rows: 4816 B per row + the 94.6 MB optimizer sweep, over the dominant kernel's time.
                         # It reads ABOVE 1: that byte model prices three gradient rows written per triple and a separate dense optimizer sweep, neither
                         # of which this step performs (gradient rows stay in registers, duplicates fold, the optimizer runs in the gather wave) --
                         # the model no longer measures this kernel, which is why `frac` is priced on measured traffic instead.
                         "survey_8d_frac": ((rows_launch * A_TRAIN + A_OPT) / k_avg / HBM_PEAK) if k_avg else None,
                         "survey_8d_frac_nominal_rows": ((per_step * A_TRAIN + A_OPT) / k_avg / HBM_PEAK) if k_avg else None,
                         "avg_kernel_ms": k_avg * 1e3, "launches": k_n,
                         "algorithmic_bytes_note": ("entity rows fetched (two per row in the forward role, one per list entry = two per row in the entity role) "
                                                    "+ the optimizer's streams over the entity table; relation rows (190 KB table, L2 resident) not counted; "
                                                    "SURVEY 8(d)'s 4816 B per triple on the same rows would be %.0f" % (rows_launch * A_TRAIN)) if twin else None,
                         "avg_kernel_ms_live_hipevents": k_live * 1e3,
                         "avg_kernel_ms_rocprofv3": rp[0] if rp else None, "rocprofv3_source": f"profiles/{rp[1]}" if rp else None,
                         "frac_live_hipevents": k_bytes / k_live / HBM_PEAK if k_live else None,
                         "all_kernels": {k: {"avg_ms": v[0] * 1e3, "algorithmic_GBps": v[1] / v[0] / 1e9} for k, v in kernels.items()},
                         # whole step on the rows really processed: gathers + gradient rows + the dense optimizer sweep
                         "whole_step_algorithmic_GBps": (rows_launch * A_TRAIN + A_OPT) / (dt / args.steps) / 1e9,  # SURVEY 8(d) figures: 4816 B per row + the 32 B per parameter of a dense TF-Adam sweep
                         "rows_per_launch": rows_launch, "batch_rows": per_step,
                         "duplicate_row_folding": ("on: the gather kernels run on the distinct rows of the batch; "
                                                   "bytes per launch are priced on those rows" if folded else "off"),
                         "fold_avg_ms": d_ms / max(1, d_n),
                         "opt_kernel_avg_ms": o_ms / max(1, o_n), "index_build_avg_ms": i_ms / max(1, i_n),
                         "sampler_avg_ms": s_ms / max(1, s_n), "host_enqueue_ms_per_step": t_issue / args.steps * 1e3,
                         # of which: waiting for a stage graph's previous launch to finish (back-pressure from the GPU) vs inside hipGraphLaunch
                         "host_wait_ms_per_step": host_stats["wait_prev_launch_ms"] / args.steps if host_stats else None,
                         "host_graph_launch_ms_per_step": host_stats["graph_launch_ms"] / args.steps if host_stats else None,
                         "host_wait_by_stage_ms_per_step": {k: v / args.steps for k, v in host_stats["wait_by_stage_ms"].items()} if host_stats else None,
                         "launch": (("the library's native loop (kge_pipeline_run: one C call for the K timed steps) launches every stage's kernels on four streams (generator rounds | draw | plan | step); the step is ONE launch between twin tables (chained steps: step1_kernel with its closer workgroups; KGE_PIPE_CHAIN=0: step1_kernel + rel_tail_kernel)" if not dist_on else
                                     ("the library's native loop issues every stage on its streams; the step is the data-parallel one (entity half of the gradients, all-reduce of its table on an RCCL communicator of the library's own beside the relation half, all-reduce of the rest, optimizer half)" if comm is not None else
                                      "draw and plan stages replayed from hipGraphs on their streams; the step (gradient half, all-reduce, optimizer half) launched kernel by kernel from Python over torch.distributed" + (f" [{dp_native_note}]" if dp_native_note else "")))
                                    + "; kernel durations above come from a second pass of the same steps launched kernel by kernel on one stream (no overlap; the same one-launch form of the step: kge_train_step_swap_chain from the Python loop)"
                                    if pipe is not None
                                    else "kernel-by-kernel stream launches"),
                         "plan": ("ids-only half of the step (fold + gradient lists) built 1-2 batches ahead on its own stream, inside the timed region"
                                  if plan_ahead else "built inside the step")},
        }
        if ev:
            sw = ev["sweep_kernel_avg_ms"] * 1e-3
            res["roofline_mfma"] = {"bound": "mfma", "kernel": "rank_sweep5_kernel (fp32 MFMA 1-vs-all sweep, 256x128 tiles filled by LDS-DMA, fused count epilogue)",
                                    "achieved": (ev["flop_per_launch"] / sw / 1e12) if sw else None, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                                    "frac": (ev["flop_per_launch"] / sw / MFMA_F32_PEAK) if sw else None, "avg_kernel_ms": ev["sweep_kernel_avg_ms"],
                                    "launches": ev["sweep_kernel_launches"], "algorithmic_flop_per_launch": ev["flop_per_launch"],
                                    "end_to_end_frac": ev["mfma_frac"], "traffic": pmc_traffic("rank_sweep5_kernel")[0]}
            res["eval_1vsall"] = ev
        if ev_sh:
            res["eval_sharded"] = ev_sh
        if dp02:
            res.update(dp02)
        if hbm:
            res["roofline_hbm"] = hbm
        if checked is not None:
            res.update(checked)
        else:
            res["checked"] = False
        if not args.no_cpu_baseline and world == 1 and args.model == "DistMult":
            res["cpu_baseline"] = cpu_baseline(hb, E0, R0)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        if pipe is not None:
            pipe.close()
        if comm_destroy is not None:
            comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
