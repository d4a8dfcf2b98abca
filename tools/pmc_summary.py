"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: python tools/pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k, c in acc.items():
        n = len(cnt[k])
        print(k[:100], "dispatches", n)
        for name, v in sorted(c.items()):
            print(f"   {name:32s} {v / n:16.1f} per dispatch")
