"""Stage timeline of the native training pipeline on the bench workload (GPU box): python tools/pipe_trace.py [n_launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = bench.workload("DistMult", 0)
model = bench.make_model(w)
pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 1000)
pipe.start()
pipe.run(300)
torch.cuda.synchronize()
pipe.trace(n)
pipe.run(n)
rows = pipe.trace()
for name, batch, a, b in sorted(rows, key=lambda r: r[2]):
    print(f"{a:9.1f} {b - a:7.1f}  {name:6s} {batch}")
steps_tr = sorted((a, b) for name, _k, a, b in rows if name == "step")
pipe.run(100)
pipe.finish()
# un-instrumented pace of the same object (what bench.py times)
import time  # noqa: E402
pipe.load_epoch(pipe.index, [i % (bench.CFG["n_train"] // bench.CFG["B"]) for i in range(1300)])
pipe.start()
pipe.run(200, join=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.run(1000, join=False)
torch.cuda.synchronize()
pace = (time.perf_counter() - t0) / 1000 * 1e6
pipe.finish()
# per-kernel durations INSIDE the concurrent pipeline (the library's hipEvent hooks around its dominant kernels)
from kge_keras_amd import _lib  # noqa: E402
w2 = bench.workload("DistMult", 0)
model2 = bench.make_model(w2)
pipe2, _ = bench.make_pipeline(torch, model2, w2, 0, 1000)
pipe2.start()
pipe2.run(200)
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(1)
pipe2.run(200)
torch.cuda.synchronize()
_lib.lib.kge_prof_enable(0)
conc = {}
for lab in ("fwd", "train_grads", "opt", "index", "sampler", "dedup"):
    ms, k = _lib.prof_read(lab)
    conc[lab] = ms / max(1, k) * 1e3
    print(f"concurrent {lab:12s} avg {ms / max(1, k) * 1e3:8.1f} us over {k} scopes")
pipe2.finish()
# ---- where a step's time goes: the step stage is the chain step1_kernel -> rel_tail_kernel on the step stream
import numpy as np  # noqa: E402
if len(steps_tr) > 8:
    dur = np.median([b - a for a, b in steps_tr])
    s2s = np.median(np.diff([a for a, _b in steps_tr]))
    k1, k2 = conc["train_grads"], conc["opt"]
    chained = os.environ.get("KGE_PIPE_CHAIN", "1") != "0"  # (chained steps: ONE launch per step, the closing launch only ends a run)
    print(f"\naccounting of one training step (medians / averages, us; {'chained steps: one launch per step' if chained else 'two launches per step'}):")
    print(f"  pace of the un-instrumented pipeline (1000 steps, wall clock)                 {pace:7.1f}")
    print(f"  step1_kernel beside the producer stages (library hipEvents)                  {k1:7.1f}" + ("   (write-through partial rows, closers and state workgroup included)" if chained else ""))
    if chained:
        print(f"  step stage begin -> end with the stage trace armed                           {dur:7.1f}")
        print(f"  step stage start-to-start with the stage trace armed                         {s2s:7.1f}   => between two step stages ~ {s2s - dur:5.1f} (record of 'plan consumed' behind")
        print(f"      every second step, the satisfied wait for the next plan, launch of step1_kernel; the trace's own two event records per stage are inside this number)")
        print(f"  un-instrumented: pace - step1 = {pace - k1:5.1f} us between two launches of step1_kernel (one kernel boundary on a stream that three other streams")
        print(f"      run beside + the wait for the plan + every second step one event record)")
    else:
        print(f"  rel_tail_kernel (library hipEvents)                                          {k2:7.1f}")
        print(f"  step stage begin -> end with the stage trace armed (both kernels + the gap)  {dur:7.1f}   => stream-order gap step1 -> rel_tail ~ {dur - k1 - k2:5.1f}")
        print(f"  step stage start-to-start with the stage trace armed                         {s2s:7.1f}   => between two step stages ~ {s2s - dur:5.1f} (record of 'plan consumed', the")
        print(f"      satisfied wait for the next plan, launch of step1_kernel; the trace's own two event records per stage are inside this number)")
        print(f"  un-instrumented: pace - step1 - rel_tail = {pace - k1 - k2:5.1f} us of launch gaps and event operations per step (two kernel boundaries of ~3 us on a")
        print(f"      stream that three other streams run beside + one event record + one event wait)")
