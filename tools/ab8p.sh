#!/bin/bash
# Tool-only builds of the 256 x 320 phase-interleaved tile (csrc/gemm_conv8p.hip): one library per "-D..." argument set,
#   bash tools/ab8p.sh v1 "-DES8P_SCHED=1"   -> edgestyle_amd/lib/ablate/libes_8p_v1.so
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
NAME=$1; shift
mkdir -p ../lib/ablate/obj_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=fast "$@" -c gemm_conv8p.hip -o ../lib/ablate/obj_$NAME/gemm_conv8p.o
OBJS=""
for f in gemm_conv linear_xs attention norm fusion elementwise plan builder; do OBJS="$OBJS ../lib/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_8p_$NAME.so $OBJS ../lib/ablate/obj_$NAME/gemm_conv8p.o
echo built ../lib/ablate/libes_8p_$NAME.so
