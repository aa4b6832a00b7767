"""Wall time of utils.validate (utils.py:199-264) end to end at FB15k-237 sizes: 20 466 test triples, filters = 310 k triples."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd import models as M  # noqa: E402
from kge_keras_amd.utils import validate  # noqa: E402

N, R, d = 14541, 237, 200
rs = np.random.RandomState(0)
mdl = M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, seed=1)
tri = lambda n: [tuple(int(v) for v in r) for r in np.stack([rs.randint(0, N, n), rs.randint(0, R, n), rs.randint(0, N, n)], 1)]  # noqa: E731
test = list(dict.fromkeys(tri(20466)))
tset = set(test)
filt = [t for t in tri(310000) if t not in tset]
validate(mdl, test[:100], N, None, filtering_triples=filt[:1000])
t0 = time.perf_counter()
res = validate(mdl, test, N, None, filtering_triples=filt)
dt = time.perf_counter() - t0
print(f"validate: {len(test)} test triples x 2 sides, {len(filt)} filter triples: {dt:.3f} s  ({2*len(test)/dt/1e3:.1f} k ranks/s end to end)  mrr={res['mrr']:.5f}")
