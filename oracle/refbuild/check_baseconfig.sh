#!/bin/bash
# TEST INFRASTRUCTURE.  Runs the reference's own `configure` (pre-generated; 2-3 minutes, build
# container only) out of tree and compares the #define set of the magick-baseconfig.h it writes
# with the hand-written oracle/refbuild/baseconfig.h the oracle and shim builds use.
#   bash oracle/refbuild/check_baseconfig.sh [/root/reference]  > oracle/refbuild/baseconfig.check.txt
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
WORK=$(mktemp -d /tmp/refcfg.XXXXXX)
( cd "$WORK" && "$REF/configure" --disable-hdri --with-quantum-depth=16 --enable-openmp --disable-opencl \
    --without-x --without-perl --without-magick-plus-plus --disable-shared --enable-static --disable-docs \
    --without-modules > configure.log 2>&1 ) || { echo "configure failed, see $WORK/configure.log"; exit 1; }
echo "# configure: $(grep -m1 'ImageMagick' "$WORK/configure.log" | head -1)"
echo "# '<' = the reference's configure, '>' = oracle/refbuild/baseconfig.h; #define lines only"
diff <(grep -E '^#define' "$WORK/MagickCore/magick-baseconfig.h" | sort) \
     <(grep -E '^#define' "$HERE/baseconfig.h" | sort)
echo "# CFLAGS the reference's configure chose: $(grep -m1 '^CFLAGS' "$WORK/Makefile" | cut -c1-200)"
rm -rf "$WORK"
