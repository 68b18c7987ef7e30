#!/bin/bash
# A/B of resize_stream.hip builds:  tools/build_stream_variant.sh <name> <extra hipcc flags...>
# -> imagemagick_amd/lib/libmagickhip_<name>.so (the other objects are those of the regular build)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/imagemagick_amd/csrc
obj=$root/imagemagick_amd/build_$name
rm -rf "$obj"; mkdir -p "$obj"; cp $root/imagemagick_amd/build/*.o "$obj"/
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -DMAGICKHIP_BUILD \
  -I$root/include -I$src -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c $src/resize_stream.hip -o $obj/resize_stream.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/imagemagick_amd/lib/libmagickhip_$name.so $obj/*.o \
  -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
rm -rf "$obj"
echo built libmagickhip_$name.so
