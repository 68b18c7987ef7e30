#!/usr/bin/env python3
"""Round 3: gpurun_out/profiles_<tag>/ (tools/collect_profiles_r3.sh) -> profiles/<tag>_*.csv,
profiles/pmc_traffic.json, profiles/<tag>_sq_counters.json.  Same files and unit corrections as
tools/import_profiles_r2.py (which this reuses), plus the kernels and workloads of the round: the
exact-integer fused blur in its two forms, round 2's f16 kernel as `legacy`, the float-Quantum blur."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import import_profiles_r2 as base

base.LABELS = [
    (r"blur_fused_exact_kernel<\d+, \d+, true, true>", "unsharp_fused_exact"),
    (r"blur_fused_exact_kernel<\d+, \d+, false, true>", "blur_fused_exact"),
    (r"blur_fused_exact_kernel<\d+, \d+, true, false>", "unsharp_fused_exact_row"),
    (r"blur_fused_exact_kernel<\d+, \d+, false, false>", "blur_fused_exact_row"),
    (r"conv2d_exact_kernel", "conv2d_exact"),
    (r"stretch_apply", "apply_lut"), (r"stretch_", "build_lut"), (r"morph_strips", "morph_rects"),
] + base.LABELS
base.PREFIX = {"fast": "", "exact": "", "legacy": "", "hdri": "hdri:", "resize": "", "c4": "c4:", "c5": "c5:"}


def sq_counters(tag):
    src = os.path.join(base.ROOT, "gpurun_out", "profiles_" + tag)
    out = {}
    for w in ("fast", "exact"):
        acc = defaultdict(list)
        for path in glob.glob(os.path.join(src, "sq_%s_*" % w, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "blur_fused_exact_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if acc:
            out[w] = {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}
            c = out[w]
            if "SQ_WAVE_CYCLES" in c:
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if k in c:
                        c[k + "/SQ_WAVE_CYCLES"] = round(c[k] / c["SQ_WAVE_CYCLES"], 3)
    if out:
        json.dump(out, open(os.path.join(base.ROOT, "profiles", tag + "_sq_counters.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r3a"
    base.main(tag)
    sq_counters(tag)
