#!/bin/bash
# Tool-only library variants: rebuild ONE source with extra -D flags and link it with the product objects.
#   bash tools/build_variant.sh <name> <source.hip> "<extra flags>"   -> edgestyle_amd/lib/ablate/libes_<name>.so
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p ../lib/ablate
base=$(basename $src .hip)
objs=""
for o in gemm_conv gemm_conv8p linear_xs attention norm fusion elementwise plan builder; do
  if [ "$o" != "$base" ]; then objs="$objs ../lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $flags -c $src -o ../lib/ablate/${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_${name}.so $objs ../lib/ablate/${name}.o
echo "built edgestyle_amd/lib/ablate/libes_${name}.so"
