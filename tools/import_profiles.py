#!/usr/bin/env python3
"""gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh, run on the GPU box) -> the committed files under profiles/:
   <tag>_kernel_stats.csv              rocprofv3 --kernel-trace --stats of the headline command (per kernel)
   <tag>_kernel_stats_all_configs.csv  the same for the whole default bench command
   <tag>_pmc.csv                       per workload and kernel: mean FETCH_SIZE / WRITE_SIZE per launch
   pmc_traffic.json                    HBM bytes per launch, keyed the way bench.py looks them up (FETCH_SIZE is in
                                       KiB and counts half the bytes of a wide coalesced stream on gfx950: x2;
                                       WRITE_SIZE is KiB: /opt/skills/guides/MI355X_MICROARCH.md, HBM section)
   <tag>_sq_counters.json              SQ issue / wait / LDS counters of the two one-launch blur kernels
   <tag>_bench.json                    the bench line of the same box
    python tools/import_profiles.py <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# HIP kernel name -> the label the library's launch records (and bench.py) use
LABELS = [
    (r"blur_fused_hybrid_kernel", "blur_fused_hybrid"),
    (r"resize_stream_careful", "resize_stream_careful"), (r"resize_stream", "resize_stream"), (r"resize_mfma", "resize_mfma"),
    (r"blur_fused_exact_kernel<\d+, \d+, true, true>", "unsharp_fused_exact"),
    (r"blur_fused_exact_kernel<\d+, \d+, false, true>", "blur_fused_exact"),
    (r"conv2d_exact_kernel", "conv2d_exact"), (r"conv2d_tie", "conv2d_tie"),
    (r"stretch_apply", "apply_lut"), (r"stretch_", "build_lut"), (r"morph_strips", "morph_rects"),
    (r"conv_mfma_kernel<true", "conv_column"), (r"conv_mfma_kernel<false", "conv_row"),
    (r"conv_column_", "conv_column"), (r"conv_row_alpha_audit", "conv_row_alpha_audit"), (r"conv_row_", "conv_row"),
    (r"resize_vertical", "resize_vertical"), (r"resize_horizontal", "resize_horizontal"),
    (r"lab_histogram_fast", "colorspace_histogram"), (r"colorspace_", "colorspace"),
    (r"histogram_packed_reduce", "colorspace_histogram"), (r"histogram_", "histogram"), (r"apply_lut", "apply_lut"),
    (r"lut_", "build_lut"), (r"gray_", "gray_check"),
    (r"conv2d_mfma", "conv2d_mfma"), (r"morph_rects", "morph_rects"), (r"morph_convex", "morph_convex"),
    (r"morph2d", "morph2d"), (r"unsharp_kernel", "unsharp_epilogue"),
]
PREFIX = {"fast": "", "exact": "", "hdri": "hdri:", "resize": "", "c4": "c4:", "c5": "c5:"}


def label(name):
    name = name.replace("void ", "").replace("mh::", "")
    for pattern, lab in LABELS:
        if re.match(pattern, name):
            return lab
    return None


def kernel_stats(src, dst, tag):
    for sub, suffix in (("stats", "_kernel_stats.csv"), ("stats_full", "_kernel_stats_all_configs.csv")):
        stats = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        with open(os.path.join(dst, tag + suffix), "w") as f:
            f.write("kernel,calls,total_ns,average_ns,percentage\n")
            for r in csv.DictReader(open(stats[0])):
                n = r["Name"]
                n = n if len(n) < 140 else n[:137] + "..."
                f.write('"%s",%s,%s,%s,%s\n' % (n.replace('"', "'"), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))


def pmc(src, dst, tag):
    traffic, detail = {}, {}
    with open(os.path.join(dst, tag + "_pmc.csv"), "w") as f:
        f.write("workload,kernel,label,counter,mean_per_launch,launches\n")
        for w, prefix in PREFIX.items():
            # label -> kernel name -> counter -> values (a label can cover several kernels of one operator call)
            acc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
            for path in glob.glob(os.path.join(src, "pmc_%s_*" % w, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    lab = label(r["Kernel_Name"])
                    if lab is None:
                        continue
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:90]
                    acc[lab][name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for lab in sorted(acc):
                fetch = write = 0.0
                complete = True
                for name in sorted(acc[lab]):
                    for c in sorted(acc[lab][name]):
                        v = acc[lab][name][c]
                        f.write('%s,"%s",%s,%s,%.6g,%d\n' % (w, name, lab, c, sum(v) / len(v), len(v)))
                    k = acc[lab][name]
                    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
                        fetch += sum(k["FETCH_SIZE"]) / len(k["FETCH_SIZE"]) * 1024.0 * 2.0
                        write += sum(k["WRITE_SIZE"]) / len(k["WRITE_SIZE"]) * 1024.0
                    else:
                        complete = False
                if complete:
                    traffic[prefix + lab] = round(fetch + write)
                    detail[prefix + lab] = {"kernels": sorted(acc[lab]), "fetch_bytes": round(fetch), "write_bytes": round(write)}
    traffic["_detail"] = detail
    traffic["_source"] = "profiles/%s_pmc.csv (tools/collect_profiles.sh + tools/import_profiles.py)" % tag
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    return traffic


def sq_counters(src, dst, tag):
    out = {}
    for w, needle in (("fast", "blur_fused_hybrid_kernel"), ("exact", "blur_fused_exact_kernel")):
        acc = defaultdict(list)
        for path in glob.glob(os.path.join(src, "sq_%s_*" % w, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if needle in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        c = out[w] = {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    c[k + "/SQ_WAVE_CYCLES"] = round(c[k] / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_INSTS_MFMA"):
            c["SQ_INSTS_VALU/SQ_INSTS_MFMA"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    if out:
        json.dump(out, open(os.path.join(dst, tag + "_sq_counters.json"), "w"), indent=1, sort_keys=True)
    return out


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    kernel_stats(src, dst, tag)
    traffic = pmc(src, dst, tag)
    counters = sq_counters(src, dst, tag)
    bench = os.path.join(src, "bench.json")
    if os.path.exists(bench):
        lines = [l for l in open(bench).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(dst, tag + "_bench.json"), "w").write(lines[-1] + "\n")
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1))
    print(json.dumps(counters, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r6a")
