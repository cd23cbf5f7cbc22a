#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh <name> <file.hip> "<-D flags>"  -> strajnet_amd/variants/lib_<name>.so
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/.."
mkdir -p strajnet_amd/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -ffp-contract=fast"
/opt/rocm/bin/hipcc $F $flags -c strajnet_amd/csrc/$src -o strajnet_amd/variants/${name}.o
objs=""
for o in strajnet_amd/build/*.o; do
  [ "$(basename $o)" = "${src%.hip}.o" ] && objs="$objs strajnet_amd/variants/${name}.o" || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o strajnet_amd/variants/lib_${name}.so $objs
echo built strajnet_amd/variants/lib_${name}.so
