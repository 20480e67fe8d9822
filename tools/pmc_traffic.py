"""Build profiles/rNN_pmc_traffic.json from rocprofv3 counter_collection.csv files: for each probe (tools/dc_probe.py, tools/ec_probe.py,
PB images, single stream) one --pmc FETCH_SIZE pass and one --pmc WRITE_SIZE pass (MI355X_MICROARCH.md HBM section: separate passes;
both counters are in KB and count the L2's memory-side requests, Infinity-Cache hits included; on gfx950 FETCH_SIZE reports half of
the bytes of 16-B-per-lane streaming reads, so it is doubled).  Keys are bench.py's kernel classes.

usage: pmc_traffic.py out.json images  dc_fetch.csv dc_write.csv  ec_fetch.csv ec_write.csv [imp_fetch.csv imp_write.csv]"""
import collections
import csv
import json
import sys

# (since round 5 the decode-order kernel has two names per layer type: k_cconv4v6t = the launches that tape-pack their samples, k_cconv4v6 = the others;
#  a class sums over its kernels)
# round 6: with the dead-cone lists (batches of >= 16 images) every cin = 4 decode-order launch is k_cconv4v6l<4>
CLASSES = [("k_cconv4v6<4, false", "dc_hidden"), ("k_cconv4v6t<4>", "dc_hidden"), ("k_cconv4v6l<4>", "dc_hidden"), ("k_cconv4v6<1, false", "dc_first"), ("k_cconv4v6t<1>", "dc_first"),
           ("k_cconv16s", "ec_hidden"), ("k_cconv16<4, false>", "ec_hidden"),
           ("k_cconv16<4, true>", "ec_last"), ("k_cconv16<1, false>", "ec_first"), ("k_cconv144<1, true", "imp_dc"), ("k_cconv144<1, false", "imp_ec"), ("k_imp_dc_map", "imp_dc_fused")]
out_json, images = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(dict)
for path in sys.argv[3:]:
    tot, disp, ctr = collections.defaultdict(float), collections.defaultdict(set), None
    per_disp = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        ctr = r["Counter_Name"]
        k = r["Kernel_Name"].replace("void ", "")
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        per_disp[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
    acc = {}
    for k in sorted(tot):
        for pat, cls in CLASSES:
            if k.startswith(pat):
                t = acc.setdefault(cls, {"kernel": [], "launches": 0, "KB": 0.0})
                t["kernel"].append(k.split("(")[0]); t["launches"] += len(disp[k]); t["KB"] += tot[k]
    # dc_last: the same kernels as dc_hidden; a plane's cin = 4 launches are 10 hidden layers and then the last one, so in dispatch order every
    # 11th of them is the last layer (planes on which a layer has nothing to launch do not occur for 64 x 128 latents of 48 groups)
    seq = sorted((int(d), k) for k in per_disp for pat, cls in CLASSES if cls == "dc_hidden" and k.startswith(pat) for d in per_disp[k])
    if seq and len(seq) % 11 == 0:
        last = [(d, k) for i, (d, k) in enumerate(seq) if i % 11 == 10]
        acc["dc_last"] = {"kernel": sorted({k.split("(")[0] for _, k in last}), "launches": len(last), "KB": sum(per_disp[k][str(d)] for d, k in last)}
        hid = [(d, k) for i, (d, k) in enumerate(seq) if i % 11 != 10]
        acc["dc_hidden"] = {"kernel": sorted({k.split("(")[0] for _, k in hid}), "launches": len(hid), "KB": sum(per_disp[k][str(d)] for d, k in hid)}
    for cls, t in acc.items():
        res[cls][ctr] = {"kernel": " + ".join(t["kernel"]), "launches": t["launches"], "per_launch_KB": t["KB"] / t["launches"]}
doc = {"command": "PB=%d rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/{dc,ec,imp}_probe.py (one encode + decodes of %d images, "
                  "single stream; separate passes per counter)" % (images, images),
       "note": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); Infinity-Cache hits are included in both counters; "
               "dc_hidden / dc_last: the same kernels, split by dispatch order (every 11th cin = 4 launch of the decode is a plane's last layer)",
       "masks": __import__("os").environ.get("MASKS", "smooth")}
for cls, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd, wr = 2.0 * d["FETCH_SIZE"]["per_launch_KB"] * 1024, d["WRITE_SIZE"]["per_launch_KB"] * 1024
        doc[cls] = {"kernel": d["FETCH_SIZE"]["kernel"], "images_per_launch": images, "launches_profiled": d["FETCH_SIZE"]["launches"],
                    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr}
json.dump(doc, open(out_json, "w"), indent=1)
for cls in doc:
    if isinstance(doc[cls], dict):
        print(cls, "read %.1f MB + write %.1f MB per launch" % (doc[cls]["hbm_read_bytes_per_launch"] / 1e6, doc[cls]["hbm_write_bytes_per_launch"] / 1e6))
