"""Sum rocprofv3 --pmc counter CSVs per kernel: python benchmarks_dev/pmc_summary.py <dir> [name filter]."""
import csv
import glob
import sys
from collections import defaultdict

root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if flt and flt not in k:
                continue
            k = k[:90]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
for k, cs in acc.items():
    print(f"{k}  dispatches={len(calls[k])}")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {v:18.0f}  per dispatch {v / max(len(calls[k]), 1):16.0f}")
