#!/usr/bin/env python3
"""Rounds 4-5: gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) -> profiles/<tag>_*.csv,
profiles/pmc_traffic.json, profiles/<tag>_sq_counters.json.  Same files and unit corrections as
tools/import_profiles_r3.py, plus the round's kernel: blur_fused_hybrid (FAST BlurImage: f16 colour
sums + exact alpha sums) and round 5's one-launch resize kernels (resize_stream, resize_mfma)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import import_profiles_r2 as base
import import_profiles_r3 as r3  # noqa: F401  (registers round 3's labels on base)

base.LABELS = [(r"blur_fused_hybrid_kernel", "blur_fused_hybrid"),
               (r"resize_stream_careful", "resize_stream_careful"), (r"resize_stream", "resize_stream"),
               (r"resize_mfma", "resize_mfma")] + base.LABELS
base.PREFIX = {"fast": "", "exact": "", "hdri": "hdri:", "resize": "", "c4": "c4:", "c5": "c5:"}


def sq_counters(tag):
    src = os.path.join(base.ROOT, "gpurun_out", "profiles_" + tag)
    out = {}
    for w, needle in (("fast", "blur_fused_hybrid_kernel"), ("exact", "blur_fused_exact_kernel")):
        acc = defaultdict(list)
        for path in glob.glob(os.path.join(src, "sq_%s_*" % w, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if needle in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if acc:
            out[w] = {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}
            c = out[w]
            if "SQ_WAVE_CYCLES" in c:
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if k in c:
                        c[k + "/SQ_WAVE_CYCLES"] = round(c[k] / c["SQ_WAVE_CYCLES"], 3)
            if c.get("SQ_INSTS_MFMA"):
                c["SQ_INSTS_VALU/SQ_INSTS_MFMA"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"], 2)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                c["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    if out:
        json.dump(out, open(os.path.join(base.ROOT, "profiles", tag + "_sq_counters.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r4a"
    base.main(tag)
    sq_counters(tag)
