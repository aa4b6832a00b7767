#!/usr/bin/env python3
"""bench.py -- scored triples/sec of the KGE train step (BASELINE.json metric) on N MI355X.

Workload (BASELINE.json configs[1]): FB15k-237-shaped DistMult d=200, B=2048 positives, 100 negatives ->
m = 204 800 rows per input array, 409 600 scored triples per step; N=14 541 entities, R=237 relations;
synthetic uniform-id triples (the train split is a missing blob in the reference), Glorot-uniform tables from
RandomState(1234); pointwize_hinge (margin 1) + Keras-semantics Adam.  A step = score(pos) + score(neg) + loss +
gradients + optimizer update on one batch resident in HBM.  By default (--sampler device) every batch is ALSO drawn inside
the timed region by the bit-exact MT19937 HIP sampler (negative sampling + oversampling, the reference's exact draws) on a
side stream, one batch ahead of training; --sampler pre uses batches built on the host before timing.

N > 1: data-parallel replicas, one process per GPU (torch.distributed, backend nccl = RCCL): every rank runs the
gradient half on its own batch, the dense gradient tables are all-reduced (mean) over xGMI, every rank applies
the same update.  value = triples scored by all ranks / max-over-ranks wall time (weak scaling).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(N=14541, R=237, d=200, B=2048, ns=100, n_train=272115)
A_FWD = 4 * (2 * CFG["d"] + CFG["d"]) + 12 + 4   # two entity rows, one relation row, 3 ids, 1 score
A_BWD = 4 * (2 * CFG["d"] + CFG["d"])            # three gradient rows written once
A_TRAIN = A_FWD + A_BWD                          # 4816 B per scored triple at d=200
HBM_PEAK = 8.0e12                                # MI355X_MICROARCH.md: 8 TB/s spec


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the named kernel family from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    produced by tools/pmc.sh; FETCH_SIZE x2 + WRITE_SIZE, see that file's _source).  None when no pass exists."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None
    fam = kernel_label.split()[0]
    best = None
    for name, v in d.get("kernels", {}).items():
        if v.get("family") == fam and (best is None or v["hbm_bytes_per_launch"] > best):
            best = v["hbm_bytes_per_launch"]
    return best


def rocprof_avg_ms(kernel_label):
    """Average duration of the named kernel family in the newest committed `rocprofv3 --kernel-trace --stats` summary of
    `bench.py --serial` (profiles/*_serial_kernel_stats.csv): every kernel alone, dispatches serialised by the profiler with cold
    caches in between -- the cold-cache counterpart of the live hipEvent duration.  None when no summary is committed."""
    import csv
    import glob
    fam = kernel_label.split()[0]
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_serial_kernel_stats.csv")))
    if not files:
        return None
    try:
        for r in csv.DictReader(open(files[-1])):
            if ("kge::" + fam + "<") in r["Name"] or ("kge::" + fam + "(") in r["Name"]:
                return float(r["AverageNs"]) * 1e-6, os.path.basename(files[-1])
    except Exception:
        return None
    return None


def make_batches(n_batches, rank):
    from kge_keras_amd.sampler import NumpyStream, PythonStream, build_batch
    rs = np.random.RandomState(1234 + 7919 * rank)
    train = np.stack([rs.randint(0, CFG["N"], CFG["n_train"]), rs.randint(0, CFG["R"], CFG["n_train"]),
                      rs.randint(0, CFG["N"], CFG["n_train"])], 1).astype(np.int32)
    nps, pys = NumpyStream(rank), PythonStream(rank)
    perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))
    out = []
    for b in range(n_batches):
        batch = train[perm[b * CFG["B"]:(b + 1) * CFG["B"]]]
        out.append(build_batch(nps, pys, batch, CFG["ns"], CFG["N"]))
    return out


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


def eval_leg(torch, dist, model, rank, world, reps=3, dist_on=False):
    """1-vs-all ranks/sec on the same FB15k-237-shaped tables: 40 932 queries (20 466 test triples x head/tail),
    filters = 3 random ids per query.  N > 1: every rank holds a row shard of E, the shards are all-gathered over
    RCCL into a full table, and the queries are split across ranks (SURVEY 8(e) schedule i)."""
    import numpy as np
    from kge_keras_amd import _lib
    rs = np.random.RandomState(99)
    nt = 20466
    test = np.stack([rs.randint(0, CFG["N"], nt), rs.randint(0, CFG["R"], nt), rs.randint(0, CFG["N"], nt)], 1).astype(np.int32)
    mine = test[rank::world]
    nq = len(mine)
    fq = np.repeat(np.arange(nq, dtype=np.int32), 3)
    fi = rs.randint(0, CFG["N"], 3 * nq).astype(np.int32)
    keep = (fi != np.repeat(mine[:, 2], 3)) & (fi != np.repeat(mine[:, 0], 3))
    fq, fi = torch.from_numpy(fq[keep]).cuda(), torch.from_numpy(fi[keep]).cuda()  # the filter index is resident, like the tables
    tq = torch.from_numpy(mine).cuda()
    from kge_keras_amd import dist as D
    lo, hi = D.shard_range(CFG["N"], rank, world)
    shard_at_rest = model.E[lo:hi].clone()

    def sweep():
        if dist_on:  # all-gather of the row shards into the full table (kge_keras_amd.dist.allgather_rows)
            model.E.copy_(D.allgather_rows(shard_at_rest, CFG["N"]))
        c0 = model.rank_counts(tq, 0, fq, fi)
        c1 = model.rank_counts(tq, 1, fq, fi)
        return c0, c1

    sweep()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    _lib.lib.kge_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        c0, c1 = sweep()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    _lib.lib.kge_prof_enable(0)
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms, n = _lib.prof_read("rank")
    total_q = 2 * nt
    flops = 2.0 * CFG["d"] * CFG["N"]  # 2*d_e*N per rank
    return dict(ranks_per_sec=total_q * reps / dt, queries=total_q, ms_per_sweep=dt / reps * 1e3,
                mfma_tflops_algorithmic=total_q * reps * flops / dt / 1e12, mfma_peak_tflops=157.3,
                mfma_frac=total_q * reps * flops / dt / 157.3e12, rank_call_avg_ms=ms / max(1, n),
                mean_rank=float((1 + c0.float().mean() + 1 + c1.float().mean()).item() / 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sampler", default="device", choices=["device", "pre"],
                    help="device: every step's batch is drawn by the bit-exact MT19937 HIP sampler on a side stream, one batch ahead "
                         "(inside the timed region); pre: batches pre-built on the host before timing")
    ap.add_argument("--plan", default="ahead", choices=["ahead", "step"],
                    help="ahead: batch plans (ids-only half of the step) built on their own stream ahead of the step; step: inside the step")
    ap.add_argument("--settle", type=int, default=20000, help="upper bound of the extra untimed steps run until the step time is stable")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "stream"],
                    help="graph / auto: the TrainPipeline object (every stage one hipGraph launch per step on one GPU; kernel-by-kernel "
                         "launches of the same stages under torch.distributed); stream: the plain double-buffered loop of this file")
    ap.add_argument("--serial", action="store_true",
                    help="profiling aid: run the pipeline stages one after the other on one stream, kernel by kernel, in every phase "
                         "(what the roofline leg does), so that a rocprofv3 kernel trace of the run times every kernel alone")
    ap.add_argument("--model", default="DistMult", choices=["DistMult", "TransE"],
                    help="DistMult: the BASELINE configs[1] workload (the contract line).  TransE: configs[2] (L2, gamma 12) through the same "
                         "pipeline, for comparison only (no CPU baseline; the metric string names it)")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed code paths even with WORLD_SIZE=1 (rehearsal on one GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # --force-dist: rehearse the torch.distributed code paths (RCCL process group, all-reduce between the halves of a step,
    # sharded eval) with a world of one process, on a box that has a single GPU
    dist_on = world > 1 or args.force_dist
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")
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
    from kge_keras_amd import models as KM

    rs = np.random.RandomState(1234)
    def glorot(rows, cols):  # Keras GlorotUniform on an Embedding matrix (models.py:87-93): U(-l, l), l = sqrt(6 / (rows + cols))
        lim = np.sqrt(6.0 / (rows + cols))
        return rs.uniform(-lim, lim, size=(rows, cols)).astype(np.float32)
    E0, R0 = glorot(CFG["N"], CFG["d"]), glorot(CFG["R"], CFG["d"])
    mkw = dict(norm=2, gamma=12) if args.model == "TransE" else {}
    model = getattr(KM, args.model)(e_dim=CFG["d"], r_dim=CFG["d"], num_entities=CFG["N"], num_relations=CFG["R"], dp=0.0,
                                    init_entities=E0, init_relations=R0, loss_function="pointwize_hinge", margin=1, **mkw)
    model.compile("adam")
    n_pool = 8
    host_batches = make_batches(n_pool, rank)
    dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(n).cuda()) for p, n in host_batches]
    m = CFG["B"] * CFG["ns"]
    per_step = 2 * m

    # device sampler pipeline: batch i+2 is drawn on a side stream while batch i trains (double-buffered in HBM)
    from kge_keras_amd.sampler import NumpyStream, PythonStream, DeviceSampler
    main_stream, side = torch.cuda.current_stream(), torch.cuda.Stream()
    rs_t = np.random.RandomState(1234 + 7919 * rank)
    train = np.stack([rs_t.randint(0, CFG["N"], CFG["n_train"]), rs_t.randint(0, CFG["R"], CFG["n_train"]),
                      rs_t.randint(0, CFG["N"], CFG["n_train"])], 1).astype(np.int32)
    nps = NumpyStream(rank)
    perm = nps.shuffle(np.arange(CFG["n_train"], dtype=np.int64))  # epoch permutation (host, once per epoch)
    train_dev, perm_dev = torch.from_numpy(train).cuda(), torch.from_numpy(perm).cuda()
    dsamp = DeviceSampler(nps, PythonStream(rank))
    bufs = [tuple(torch.empty((m, 3), dtype=torch.int32, device="cuda") for _ in range(2)) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]     # batch arrays drawn
    freed = [torch.cuda.Event() for _ in range(2)]     # batch arrays no longer needed
    planned = [torch.cuda.Event() for _ in range(2)]   # batch plan built
    pfreed = [torch.cuda.Event() for _ in range(2)]    # batch plan consumed
    nb_epoch = CFG["n_train"] // CFG["B"]
    # --plan ahead (default): the ids-only half of the step (duplicate-row fold, gradient lists) is built on a third stream
    # right behind the sampler, one to two batches ahead of the step that consumes it; --plan step builds it inside the step
    plan_ahead = args.plan == "ahead" and model.plan_supported()
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

    def step(i):  # kernel-by-kernel launches (--launch stream, --sampler pre, --plan step)
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
        else:  # data-parallel: gradient half, all-reduce of the dense gradient tables over RCCL, optimizer half
            # gradient tables and the loss word are one flat buffer: ONE all-reduce per step
            D.dp_train_step(lambda: model.train_grads(data), model.apply_grads, [model._gall], None)
        if plan_ahead:
            pfreed[j].record(main_stream)
        elif args.sampler == "device":
            freed[j].record(main_stream)
        if args.sampler == "device" or plan_ahead:
            produce(i + 2)

    # default: the three-stage pipeline (draw batch t+2 | plan batch t+1 | train step t) replayed from hipGraphs, one launch per step
    pipe = None
    # auto: the pipeline object in both cases; under torch.distributed it launches its stages kernel by kernel (the gradient
    # all-reduce sits between the two halves of the step and stays outside any graph)
    launch = args.launch if args.launch != "auto" else "graph"
    if args.sampler == "device" and plan_ahead and launch == "graph":
        from kge_keras_amd.pipeline import TrainPipeline
        # gradient tables and the loss word are one flat buffer: ONE all-reduce per step (= dist.allreduce_sum_([model._gall]))
        between = (lambda: (dist.all_reduce(model._gall, op=dist.ReduceOp.SUM), 1.0 / world)[1]) if dist_on else None
        pipe = TrainPipeline(model, train_dev, CFG["n_train"], CFG["B"], CFG["ns"], CFG["N"], dsamp, graphs=True, between=between)
        total = args.warmup + args.settle + 2 * args.steps + 8
        if args.serial:
            pipe.use_graphs, pipe.serial = False, True
        pipe.load_epoch(perm_dev, [i % nb_epoch for i in range(total)])
        pipe.start()
        step = lambda i: pipe.tick()  # noqa: E731
    elif args.sampler == "device" or plan_ahead:
        side.wait_stream(main_stream)
        if pstream is not None:
            pstream.wait_stream(main_stream)
        produce(0)
        produce(1)

    def fence():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # settle (untimed, bounded): the first process on a fresh box can run several times slower for its first tenths of a second
    # (the image is still paging in: a cold host, not a cold GPU).  Keep stepping in blocks of 20 for at least 0.6 s and until a
    # block is within 15 % of the best one seen, at most 3 s.  Under torch.distributed every rank runs the same fixed count.
    settle, best, t_settle = 0, None, time.perf_counter()
    while settle + 20 <= args.settle:
        el = time.perf_counter() - t_settle
        if dist_on and settle >= min(args.settle, 2000):
            break
        if not dist_on and el > 3.0:
            break
        tb = time.perf_counter()
        for i in range(20):
            step(args.warmup + settle + i)
        fence()
        tb = time.perf_counter() - tb
        settle += 20
        if not dist_on and best is not None and el >= 0.6 and tb <= 1.15 * best:
            break
        best = tb if best is None else min(best, tb)
    base = args.warmup + settle
    if pipe is None:
        _lib.lib.kge_prof_enable(1)  # hipEvents around the dominant kernels, recorded on the launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(base + i)
    t_issue = time.perf_counter() - t0  # host time to enqueue the steps (the GPU runs behind it)
    fence()
    dt = time.perf_counter() - t0
    if pipe is not None:
        # kernel durations for the roofline object: the same K steps again, launched kernel by kernel on ONE stream (events cannot
        # be read back from inside a replayed graph, and kernels of concurrent stages stretch each other: the roofline is about
        # the kernel, so it is timed alone, like rocprofv3's kernel trace does), hipEvents recorded by the library on that stream
        fence()
        pipe.use_graphs = False
        pipe.serial = True
        _lib.lib.kge_prof_enable(1)
        for i in range(args.steps):
            step(base + args.steps + i)
        fence()
        pipe.finish()
    _lib.lib.kge_prof_enable(0)
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    model._raise_on_flag()
    loss = float(model._loss.item()) / world  # summed over ranks by the flat all-reduce
    s_ms, s_n = _lib.prof_read("sampler")
    i_ms, i_n = _lib.prof_read("index")
    assert int(dsamp.err.item()) == 0, "device sampler: rejection margin ran short"
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
    else:
        used = list(pipe.bufs) if pipe is not None else (list(bufs) if args.sampler == "device" else dev)
        rows_launch = float(np.mean([distinct(p) + distinct(n) for p, n in used])) if folded else float(per_step)

    ev = eval_leg(torch, dist, model, rank, world, dist_on=dist_on)

    if rank == 0:
        value = per_step * world * args.steps / dt
        # pull path: two gather kernels (fwd_kernel: A_FWD per triple; pull_kernel: A_BWD per triple); push path: one
        # fused kernel carrying A_TRAIN.  The roofline object is for the dominant (longest) one.
        kernels = {}
        variant = 1 if os.environ.get("KGE_PULL_VARIANT", "2") == "1" else 2
        if f_n and variant == 2:
            # relation-grouped forward carries the relation-gradient row too; the pull is left with the two entity rows
            kernels["fwd_rel_kernel (gather+score+loss+relation gradient)"] = (f_ms / f_n * 1e-3, rows_launch * (A_FWD + 4 * CFG["d"]))
            kernels["pull_kernel (per-entity-row gather-reduce of gradients)"] = (k_ms / max(1, k_n) * 1e-3, rows_launch * 8 * CFG["d"])
        elif f_n:
            kernels["fwd_kernel (gather+score+loss)"] = (f_ms / f_n * 1e-3, per_step * A_FWD)
            kernels["pull_kernel (per-row gather-reduce of gradients)"] = (k_ms / max(1, k_n) * 1e-3, per_step * A_BWD)
        else:
            kernels["push_kernel (gather+score+loss+atomic grad)"] = (k_ms / max(1, k_n) * 1e-3, per_step * A_TRAIN)
        # the roofline object describes the kernel that moves the most bytes on the memory side per launch (committed PMC passes:
        # pull_kernel 196 MB, fwd_rel_kernel 169 MB), falling back to the longest one when no PMC pass exists
        dom = max(kernels, key=lambda k: (pmc_traffic(k) or 0.0, kernels[k][0]))
        k_avg, k_bytes = kernels[dom]
        achieved = k_bytes / k_avg if k_avg else 0.0
        res = {
            "metric": "scored triples/sec (train, 100 neg), FB15k-237 d=200" + ("" if args.model == "DistMult" else f" [{args.model}-L2]"), "value": value, "unit": "scored triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FB15k-237 {args.model if args.model == 'DistMult' else 'TransE-L2 (gamma 12)'} d=200, B=2048, 100 negatives (409600 scored triples/step), pointwize_hinge + Adam(TF semantics)",
                       "N": CFG["N"], "R": CFG["R"], "d": CFG["d"], "B": CFG["B"], "ns": CFG["ns"],
                       "parallelism": f"dp{world}" if dist_on else "single",
                       "sampler": ("bit-exact MT19937 HIP sampler inside the timed region (side stream, one batch ahead)" if args.sampler == "device"
                                   else "bit-exact MT19937 batches pre-built on host, resident in HBM"),
                       "final_loss": loss},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": pmc_traffic(dom),
                         "algorithmic_bytes_per_launch": k_bytes, "avg_kernel_ms": k_avg * 1e3, "launches": k_n,
                         "cold_cache": (lambda rp: None if rp is None else {
                             "avg_kernel_ms": rp[0], "achieved": k_bytes / (rp[0] * 1e-3) / 1e9, "frac": k_bytes / (rp[0] * 1e-3) / HBM_PEAK,
                             "source": f"profiles/{rp[1]} (rocprofv3 --kernel-trace --stats of bench.py --serial: the profiler serialises "
                                       "dispatches and every kernel starts with cold caches; the live hipEvent duration above is the same "
                                       "launch right behind its producer kernels, most of its rows still in L2)"})(rocprof_avg_ms(dom)),
                         "all_kernels": {k: {"avg_ms": v[0] * 1e3, "algorithmic_GBps": v[1] / v[0] / 1e9} for k, v in kernels.items()},
                         "whole_step_algorithmic_GBps": per_step * A_TRAIN / (dt / args.steps) / 1e9,
                         "rows_per_launch": rows_launch, "batch_rows": per_step,
                         "duplicate_row_folding": ("on: the gather kernels run on the distinct rows of the batch (hash fold inside the step); "
                                                   "bytes per launch are priced on those rows" if folded else "off"),
                         "fold_avg_ms": d_ms / max(1, d_n),
                         "opt_kernel_avg_ms": o_ms / max(1, o_n), "index_build_avg_ms": i_ms / max(1, i_n),
                         "sampler_avg_ms": s_ms / max(1, s_n), "host_enqueue_ms_per_step": t_issue / args.steps * 1e3,
                         "launch": (("three hipGraph launches per step (draw | plan | step stages, one graph each, on three streams)" if not dist_on else
                                     "draw and plan stages replayed from hipGraphs on their streams; the step (gradient half, all-reduce, optimizer half) launched kernel by kernel")
                                    + "; kernel durations above come from a second pass of the same steps launched kernel by kernel on one stream (no overlap)"
                                    if pipe is not None
                                    else "kernel-by-kernel stream launches"),
                         "plan": ("ids-only half of the step (fold + gradient lists) built 1-2 batches ahead on its own stream, inside the timed region"
                                  if plan_ahead else "built inside the step")},
        }
        res["eval_1vsall"] = ev
        if rank == 0 and not args.no_cpu_baseline and world == 1 and args.model == "DistMult":
            res["cpu_baseline"] = cpu_baseline(host_batches, E0, R0)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
