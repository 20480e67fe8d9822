"""kernel durations and the idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (single stream probes)"""
import csv, sys, collections
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))))
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for i, (s, e, k) in enumerate(rows):
    dur[k] += e - s
    cnt[k] += 1
    if i + 1 < len(rows):
        gap[k] += max(0, rows[i + 1][0] - e)
print("%-60s %7s %10s %10s" % ("kernel", "calls", "dur us", "gap after us"))
for k in sorted(dur, key=lambda k: -dur[k]):
    print("%-60s %7d %10.2f %10.2f" % (k[:60], cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
print("span %.1f ms, busy %.1f ms" % ((rows[-1][1] - rows[0][0]) / 1e6, sum(dur.values()) / 1e6))
