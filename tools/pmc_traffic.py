"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 counter_collection.csv files (one --pmc FETCH_SIZE pass and one
--pmc WRITE_SIZE pass of `PB=32 python tools/dc_probe.py`, MI355X_MICROARCH.md HBM section: separate passes; both counters
are in KB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads, so it is doubled)."""
import csv, json, sys, collections

fetch_csv, write_csv, out_json, images = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
res = collections.defaultdict(dict)
for path, ctr in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in tot:
        res[k][ctr] = {"launches": len(disp[k]), "sum_KB": tot[k], "per_launch_KB": tot[k] / len(disp[k])}
dom = [k for k in res if k.startswith("k_cconv4v6<4, false")][0]          # decode order, hidden layers
rd = 2.0 * res[dom]["FETCH_SIZE"]["per_launch_KB"] * 1024
wr = res[dom]["WRITE_SIZE"]["per_launch_KB"] * 1024
json.dump({"command": "PB=%d rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python tools/dc_probe.py  (one encode + one decode of %d images, "
                      "single stream; separate passes per counter)" % (images, images),
           "note": "FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of 16-B-per-lane streaming reads "
                   "(MI355X_MICROARCH.md, HBM section) -> doubled for hbm_read_bytes",
           "kernels": res,
           "dominant_kernel": {"name": dom, "images_per_launch": images, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                               "traffic_bytes_per_launch": rd + wr}}, open(out_json, "w"), indent=1)
print(dom, "read %.1f MB + write %.1f MB per launch" % (rd / 1e6, wr / 1e6))
