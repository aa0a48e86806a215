#!/bin/bash
# Builds ablated variants of the library (linear_xs.hip with XS_ABLATE=n) next to the product one and times them.
# Run on the GPU box:  bash tools/xs_ablate.sh
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
mkdir -p ../lib/ablate
OBJS="../lib/obj/gemm_conv.o ../lib/obj/gemm_conv8p.o ../lib/obj/attention.o ../lib/obj/norm.o ../lib/obj/fusion.o ../lib/obj/elementwise.o ../lib/obj/plan.o ../lib/obj/builder.o"
for a in ${XS_VARIANTS:-0 1 2 4 8 16 32 3 7 35 39 63}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -DXS_ABLATE=$a -c linear_xs.hip -o ../lib/ablate/xs_$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_xs_$a.so $OBJS ../lib/ablate/xs_$a.o
done
