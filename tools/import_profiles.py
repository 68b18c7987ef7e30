#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ (made by tools/collect_profiles.sh on the GPU box) into
the committed files under profiles/:
   <tag>_kernel_stats.csv   per-kernel calls / total / average duration (rocprofv3 --stats)
   <tag>_pmc.csv            per-kernel mean FETCH_SIZE, WRITE_SIZE, TCC hit/miss per launch
   pmc_traffic.json         HBM bytes per launch of the blur kernels, corrected as
                            MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and reads
                            half the bytes of a wide coalesced stream on gfx950: x2; WRITE_SIZE KiB)
   <tag>_bench.json         the bench line of the same run
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void mh::", "")
    if name.startswith("conv_mfma_kernel<"):         # <VERTICAL, NQ>: one kernel template, two passes
        return "conv_column_mfma" if name.startswith("conv_mfma_kernel<true") else "conv_row_mfma"
    return name.split("<")[0].split("(")[0]


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for sub, suffix in (("stats", "_kernel_stats.csv"), ("stats_full", "_kernel_stats_all_configs.csv")):
        stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(dst, tag + suffix), "w") as f:
            f.write("kernel,calls,total_ns,average_ns,percentage\n")
            for r in rows:
                n = r["Name"]
                n = n if len(n) < 140 else n[:137] + "..."
                f.write('"%s",%s,%s,%s,%s\n' % (n.replace('"', "'"), r["Calls"], r["TotalDurationNs"],
                                                r["AverageNs"], r["Percentage"]))
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic = {}
    with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch,launches\n")
        for k in sorted(acc):
            if not k.startswith(("conv_", "resize_", "morph", "hist", "lut", "color", "unsharp")):
                continue
            for c in sorted(acc[k]):
                v = acc[k][c]
                f.write("%s,%s,%.6g,%d\n" % (k, c, sum(v) / len(v), len(v)))
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                fetch = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
                write = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
                traffic[k] = {"fetch_bytes": fetch * 1024.0 * 2.0, "write_bytes": write * 1024.0,
                              "bytes": fetch * 1024.0 * 2.0 + write * 1024.0}
    # bench.py's kernel_profile() names the passes conv_row / conv_column
    out = {}
    for k, v in traffic.items():
        if k.startswith("conv_row"):
            out["conv_row"] = round(v["bytes"])
        elif k.startswith("conv_column"):
            out["conv_column"] = round(v["bytes"])
    out["_detail"] = {k: {kk: round(vv) for kk, vv in v.items()} for k, v in traffic.items()}
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        lines = [l for l in open(bench).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, tag + "_bench.json"), "w").write(lines[-1] + "\n")
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
