"""per-kernel totals of a rocprofv3 `--kernel-trace` CSV, counting only the dispatches BEHIND the last dispatch of a marker kernel (default: torch's
`spin_kernel`, which tools/transform_profile.py launches after its warm-up pass): statistics of the steady state, without MIOpen's find-mode search.
usage: trace_after_marker.py kernel_trace.csv [marker substring] > stats.csv"""
import collections
import csv
import sys

path, marker = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "spin_kernel")
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
last = max((i for i, r in enumerate(rows) if marker in r["Kernel_Name"]), default=-1)
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in rows[last + 1:]:
    tot[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[r["Kernel_Name"]] += 1
allns = sum(tot.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
for k in sorted(tot, key=tot.get, reverse=True):
    w.writerow([k, cnt[k], int(tot[k]), "%.1f" % (tot[k] / cnt[k]), "%.2f" % (100 * tot[k] / allns)])
