#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc run: python tools/pmc_mean.py <dir> [kernel substring]
Prints, per kernel and counter, the mean over dispatches and the maximum (predicated launches that exit at once would
drag a mean down; for those the maximum is the dispatch that did the work)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "exb::"
files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[-1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if sub in name:
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(acc.items()):
    print(k)
    for c, v in sorted(dd.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.6g} max={max(v):.6g}")
