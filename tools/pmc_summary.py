#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass, made by tools/pmc.sh):
per kernel, the mean of every counter over its dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void mh::", "")[:70]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        if "conv" not in k and "blur" not in k and "resize" not in k and "morph" not in k and "hist" not in k and "lut" not in k \
           and "color" not in k:
            continue
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-24s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main(sys.argv[1])
