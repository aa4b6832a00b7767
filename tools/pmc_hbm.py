"""gpurun_out/pmc_<tag>_hbm_{fetch,write}/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --leg hbm --serial`, tools/profile_round.sh)
-> profiles/pmc_traffic_hbm.json: HBM-side bytes per launch of step1_kernel on the table past the Infinity Cache, which bench.py reports as
roofline_hbm.traffic.  Correction per MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE x 2 on gfx950 (wide reads are tallied at half their
bytes), WRITE_SIZE exact for 16-byte stores; both counters are in KiB.  usage: python tools/pmc_hbm.py <tag>"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
acc = {}
for kind in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}_hbm_{kind}/*/*counter_collection.csv")
    v = collections.defaultdict(list)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            v[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    acc[kind] = v
out = None
for name, vals in acc["fetch"].items():
    if "step1_kernel" not in name:
        continue
    f = sum(vals) / len(vals)
    w = acc["write"].get(name, [0.0])
    w = sum(w) / len(w)
    out = {"_source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) of `python3 bench.py --leg hbm --serial --steps 6`, build {tag}; "
                      "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE tallies wide reads at half their bytes)",
           "n_entities": 10_000_000, "kernel": name[:120], "launches": len(vals), "fetch_size_kb": f, "write_size_kb": w,
           "step1_kernel_bytes_per_launch": (2.0 * f + w) * 1024.0}
    break
if out is None:
    print("no step1_kernel rows found", {k: len(v) for k, v in acc.items()})
    sys.exit(1)
json.dump(out, open("profiles/pmc_traffic_hbm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
