"""Where does the native pipeline lose time when it is driven from a stream of its own?  python tools/stream_probe.py MODE
MODE ctx: everything inside `with torch.cuda.stream(st)`;  own: TrainPipeline's own-stream driving (joins with the default stream)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

mode = sys.argv[1]
w = bench.workload("DistMult", 0)
model = bench.make_model(w)


def go():
    pipe, dsamp = bench.make_pipeline(torch, model, w, 0, 4000)
    pipe.start()
    pipe.run(300)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        pipe.run(1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mode} mask {os.environ.get('KGE_PIPE_CUMASK', 'default')} nojoin {os.environ.get('KGE_DEBUG_NOJOIN', '0')}: {dt / 1000 * 1e3:.4f} ms per step", flush=True)
    pipe.finish()
    torch.cuda.synchronize()


if mode == "ctx":
    with torch.cuda.stream(torch.cuda.Stream()):
        go()
else:
    go()
