"""gpurun_out/pmc_<tag>.json (tools/pmc.sh) -> profiles/pmc_traffic.json, the per-kernel HBM-side bytes bench.py reports as
roofline.traffic.  usage: python tools/pmc_to_traffic.py <tag> <build label>"""
import json
import re
import sys

tag, label = sys.argv[1], sys.argv[2]
src = json.load(open(f"gpurun_out/pmc_{tag}.json"))
out = {"_tag": f"profiles/pmc_traffic.json from gpurun_out/pmc_{tag}.json, build {label}", "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, tools/pmc.sh) of "
                  f"`python3 bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-check`, build {label} (folded batches: the gather "
                  "kernels run on ~131.5 k distinct rows per batch); corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                  "MI355X_MICROARCH.md (gfx950 FETCH_SIZE reads 1/2 of wide coalesced reads; Infinity-Cache hits are counted).  "
                  "Note: the profiler serialises dispatches and the caches start cold for every kernel, so these are upper bounds "
                  "of the traffic of the un-profiled run",
       "kernels": {}}
for name, v in src.items():
    fam = re.sub(r"^void ", "", name)
    fam = re.sub(r"^kge::", "", fam)
    fam = re.split(r"[<(]", fam)[0]
    out["kernels"][name[:90]] = {"family": fam, "fetch_size_kb": v.get("fetch_kb_per_launch", 0.0), "write_size_kb": v.get("write_kb_per_launch", 0.0),
                                 "hbm_bytes_per_launch": v["hbm_bytes_per_launch_corrected"], "launches": v.get("launches", 0)}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(len(out["kernels"]), "kernels")
