#!/usr/bin/env python3
"""gpurun_out/profiles_<tag>/ (tools/collect_resize_profile.sh) -> profiles/<tag>_resize_kernel_stats.csv,
profiles/<tag>_resize_pmc.csv and the resize entries of profiles/pmc_traffic.json (the other workloads'
entries stay).  Unit corrections as tools/import_profiles_r2.py: FETCH_SIZE is KiB and counts half the
bytes of a wide coalesced stream on gfx950 (x2), WRITE_SIZE is KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5b"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
LABELS = (("resize_stream_careful", "resize_stream_careful"), ("resize_stream", "resize_stream"),
          ("resize_mfma", "resize_mfma"), ("resize_vertical", "resize_vertical"), ("resize_horizontal", "resize_horizontal"))


def label(name):
    for needle, lab in LABELS:
        if needle in name:
            return lab
    return None


stats = glob.glob(os.path.join(src, "resize_stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    with open(os.path.join(dst, tag + "_resize_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ns,average_ns,percentage\n")
        for r in csv.DictReader(open(stats[0])):
            f.write('"%s",%s,%s,%s,%s\n' % (r["Name"].replace('"', "'"), r["Calls"], r["TotalDurationNs"],
                                            r["AverageNs"], r["Percentage"]))
            if label(r["Name"]):
                print("%-24s calls %s average %.3f ms" % (label(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6))
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(src, "pmc_resize_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        lab = label(r["Kernel_Name"])
        if lab:
            acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic_path = os.path.join(dst, "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {"_detail": {}}
with open(os.path.join(dst, tag + "_resize_pmc.csv"), "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for lab in sorted(acc):
        for c in sorted(acc[lab]):
            v = acc[lab][c]
            f.write("%s,%s,%.6g,%d\n" % (lab, c, sum(v) / len(v), len(v)))
        if "FETCH_SIZE" in acc[lab] and "WRITE_SIZE" in acc[lab]:
            fetch = sum(acc[lab]["FETCH_SIZE"]) / len(acc[lab]["FETCH_SIZE"]) * 1024.0 * 2.0
            write = sum(acc[lab]["WRITE_SIZE"]) / len(acc[lab]["WRITE_SIZE"]) * 1024.0
            traffic[lab] = round(fetch + write)
            traffic.setdefault("_detail", {})[lab] = {"kernels": [lab], "fetch_bytes": round(fetch), "write_bytes": round(write),
                                                       "source": "profiles/%s_resize_pmc.csv" % tag}
            print("%-24s HBM bytes per launch: fetch %.3f GB write %.3f GB" % (lab, fetch / 1e9, write / 1e9))
json.dump(traffic, open(traffic_path, "w"), indent=1, sort_keys=True)
