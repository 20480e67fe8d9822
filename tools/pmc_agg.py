"""Aggregate a rocprofv3 counter_collection.csv by kernel: mean counter value per dispatch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg:
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()): print("   %-28s %14.1f per dispatch" % (c, v / n))
