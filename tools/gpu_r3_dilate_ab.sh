#!/bin/bash
# timing only: library variants of the strip walk ("-" = the default library)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-r3_dilate_ab}
mkdir -p $OUT
L=$PWD/imagemagick_amd/lib
for v in $2; do
  [ "$v" = "-" ] && s="" || s="_$v"
  MAGICKHIP_LIBRARY=$L/libmagickhip$s.so timeout 200 python tools/time_dilate.py 16384 "${3:-Disk:15,Square:8}" 2>&1 | grep "strips\|tiles" | sed "s/^/lib$s: /"
done | tee $OUT/ab.txt
