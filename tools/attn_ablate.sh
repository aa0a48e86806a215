#!/bin/bash
# Tool-only builds of attention.hip (static priority forms and ablations) -> edgestyle_amd/lib/ablate/libes_attn_<tag>.so
#   bash tools/attn_ablate.sh           (in the container; then on the GPU box: python tools/attn_ablate_run.py)
cd "$(dirname "$0")/../edgestyle_amd/csrc"
mkdir -p ../lib/ablate
OBJS="../lib/obj/gemm_conv.o ../lib/obj/linear_xs.o ../lib/obj/norm.o ../lib/obj/fusion.o ../lib/obj/elementwise.o ../lib/obj/plan.o"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast"
for v in "p0:-DATTN_PRIO=0" "p3:-DATTN_PRIO=3"; do
  tag=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c attention.hip -o ../lib/ablate/attn_$tag.o &
done
wait
for tag in p0 p3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_attn_$tag.so $OBJS ../lib/ablate/attn_$tag.o
done
ls -la ../lib/ablate/libes_attn_*.so
