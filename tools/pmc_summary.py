"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")[:60]
            acc[k][row.get("Counter_Name", "")].append(float(row.get("Counter_Value", 0)))
for k, cs in sorted(acc.items()):
    for c, v in sorted(cs.items()):
        print(f"{k:60s} {c:24s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
