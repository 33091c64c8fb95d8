#!/usr/bin/env python
"""Per-kernel averages of every counter found in one or more rocprofv3 --pmc passes (csv output).
usage: pmc_counters.py out.json dirA [dirB ...]"""
import csv, glob, json, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("mmg::", "").split("<")[0]
            if name.startswith("k_"):
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
