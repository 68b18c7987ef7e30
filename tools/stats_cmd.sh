#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command: per-kernel calls / average duration
#   tools/stats_cmd.sh <tag> <command ...>      -> gpurun_out/stats_<tag>/, table on stdout
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %5s  avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
