#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per dispatch (in order) kernel name + counters."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60])
    by.setdefault(k, {})[r["Counter_Name"]] = by.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (d, name), c in by.items():
    if flt and flt not in name: continue
    print(d, name, " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
