#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh <name> <file.hip> "<-D flags>" [replaces.hip]  -> strajnet_amd/variants/lib_<name>.so
# The variant object REPLACES build/<replaces>.o (default: the object of the same source name) in the link; every other object comes from
# strajnet_amd/build/ as it is -- so build the baseline library first, and pass `replaces` when the variant source has another file name
# (without it a differently named source was linked NEXT to nothing and the variant library was the baseline: round 4).
set -e
name=$1; src=$2; flags=$3; repl=${4:-$2}
cd "$(dirname "$0")/.."
mkdir -p strajnet_amd/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=fast"
/opt/rocm/bin/hipcc $F $flags -c strajnet_amd/csrc/$src -o strajnet_amd/variants/${name}.o
objs=""
for o in strajnet_amd/build/*.o; do
  [ "$(basename $o)" = "${repl%.hip}.o" ] && { objs="$objs strajnet_amd/variants/${name}.o"; found=1; } || objs="$objs $o"
done
[ -n "$found" ] || { echo "no object build/${repl%.hip}.o to replace" >&2; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o strajnet_amd/variants/lib_${name}.so $objs
echo built strajnet_amd/variants/lib_${name}.so
