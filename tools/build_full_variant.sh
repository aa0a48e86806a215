#!/bin/bash
# Tool-only: the WHOLE library rebuilt with extra compiler flags -> edgestyle_amd/lib/ablate/libes_<name>.so
#   bash tools/build_full_variant.sh <name> "<extra flags>" [sources to apply the flags to (default: all)]
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
name=$1; flags=$2; shift; shift
only="$@"
dir=../lib/ablate/full_$name
mkdir -p $dir
pids=""
for s in gemm_conv gemm_conv8p linear_xs attention norm fusion elementwise plan builder; do
  f="$flags"
  if [ -n "$only" ] && ! echo " $only " | grep -q " $s "; then f=""; fi
  if [ -z "$f" ] && [ -f ../lib/obj/$s.o ]; then cp ../lib/obj/$s.o $dir/$s.o; continue; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $f -c $s.hip -o $dir/$s.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_${name}.so $dir/*.o
echo "built edgestyle_amd/lib/ablate/libes_${name}.so"
