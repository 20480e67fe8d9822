"""rocprofv3 --stats summary without MIOpen's find-mode search: drop the `naive_conv_*` rows (the reference kernels MIOpen runs once per new
convolution shape on a fresh box, seconds of GPU time that are not part of any timed region) and renormalise the Percentage column.
usage: stats_without_find.py in.csv out.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if not r["Name"].startswith("naive_conv")]
tot = sum(float(r["TotalDurationNs"]) for r in keep) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    for r in keep:
        r["Percentage"] = "%.4f" % (100.0 * float(r["TotalDurationNs"]) / tot)
        w.writerow(r)
print("dropped %d find-mode rows (%.1f ms)" % (len(rows) - len(keep), sum(float(r["TotalDurationNs"]) for r in rows if r not in keep) / 1e6))
