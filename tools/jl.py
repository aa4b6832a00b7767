"""Prints the headline fields of bench.py JSON lines: python tools/jl.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, m = d["roofline"], d.get("roofline_mfma") or {}
    print(f"{f}: {d['value'] / 1e9:.3f} G triples/s  {d['ms_per_step']:.5f} ms/step  n_gpus {d['n_gpus']}  roofline frac {r['frac']:.3f} "
          f"(algorithmic {r.get('algorithmic_frac') or 0:.3f}, kernel {r['avg_kernel_ms'] * 1e3:.1f} us)  ranks/s {(d.get('value_ranks_per_sec') or 0) / 1e6:.2f} M  "
          f"mfma frac {m.get('frac') or 0:.3f} ({m.get('avg_kernel_ms')} ms)  checked {d.get('checked')} err {d.get('max_abs_err')}")
    es = d.get("eval_sharded")
    if es:
        print(f"    sharded eval: (i) {es['schedule_i']['ranks_per_sec'] / 1e6:.3f} M ranks/s, (ii) {es['schedule_ii']['ranks_per_sec'] / 1e6:.3f} M ranks/s, identical {es['ranks_identical_between_schedules']}")
