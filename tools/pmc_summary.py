"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: sum of each counter."""
import csv, sys, collections, glob, os
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    if sum(cs.values()) == 0: continue
    print(k, {c: (v, calls[(k, c)]) for c, v in cs.items()})
