#!/bin/bash
# Like build_variant.sh, but recompiles only sgpr_embed.hip with the extra flags and links it with the main library's other
# objects (sg_pr_amd/lib/obj): tools/build_embed_variant.sh <name> [-D...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/variants; mkdir -p $out/obj_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -I$root/include -I$root/sg_pr_amd/csrc -c $root/sg_pr_amd/csrc/sgpr_embed.hip -o $out/obj_$name/sgpr_embed.o
others=$(ls $root/sg_pr_amd/lib/obj/*.o | grep -v sgpr_embed.o)
hipcc --offload-arch=gfx950 -shared -fPIC $out/obj_$name/sgpr_embed.o $others -o $out/libsgpr_$name.so
echo built $out/libsgpr_$name.so
