#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 counter passes (csv output):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -- <cmd>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -- <cmd>
usage: pmc_summary.py A B out.json
FETCH_SIZE / WRITE_SIZE are in KB per dispatch; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mmg::", "")
            name = name.split("<")[0]
            acc[name].append(float(row["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), per-dispatch averages in KB; "
               "'corrected' doubles FETCH_SIZE as MI355X_MICROARCH.md (HBM section) prescribes for gfx950", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f = sum(fetch[k]) / max(1, len(fetch[k])); w = sum(write[k]) / max(1, len(write[k]))
    out["kernels"][k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "dispatches": len(fetch[k]),
                         "traffic_bytes_raw": int((f + w) * 1024), "traffic_bytes_corrected": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
