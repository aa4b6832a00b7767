"""oracle/ -- TEST INFRASTRUCTURE ONLY (not product code).

CPU restatement of the reference algorithm for the KGE hot path (NIVA-Knowledge-Graph/KGE-Keras),
used solely as the *checker* by `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py`.  Nothing in `kge-keras_amd/` imports it; the product path fails loudly when the HIP
library is missing instead of falling back to this code.

Pinning status (see DESIGN.md "Oracle"):
  * sampler / batch builder / evaluator  -> pinned by golden vectors produced by the reference's own
    NumPy/SciPy code in the build container (tests/golden/make_golden.py).
  * scoring / loss / autodiff / optimizer arithmetic -> the reference delegates these to TensorFlow,
    which is not installed and whose source is not under /root/reference: **parity unpinned** at that
    boundary.  The restatement follows KGEkeras/models.py and KGEkeras/utils.py line by line (cited
    per function) plus the published Keras semantics; it is cross-checked by known-answer tests and
    by torch-CPU autograd.
"""
from .kge_oracle import *  # noqa: F401,F403
from . import c_port  # noqa: F401  (C restatement, built on first use by oracle/Makefile)
