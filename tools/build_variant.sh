#!/bin/bash
# Build an alternative libsgpr_hip.so of the same C-ABI for same-box A/B runs (SGPR_HIP_LIB=variants/libsgpr_<name>.so):
#   tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSGPR_ROTATE_ROLES=0] [--src <dir with csrc + include>]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root
flags=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--src" ]; then src=$2; shift 2; else flags+=("$1"); shift; fi
done
out=$root/variants; mkdir -p $out/obj_$name
pids=()
for f in $src/sg_pr_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "${flags[@]}" -I$src/include -I$src/sg_pr_amd/csrc -c $f -o $out/obj_$name/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $out/obj_$name/*.o -o $out/libsgpr_$name.so
echo built $out/libsgpr_$name.so
