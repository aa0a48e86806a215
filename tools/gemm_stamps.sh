#!/bin/bash
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
mkdir -p ../lib/ablate
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -DES_STAMPS=1 -c gemm_conv.hip -o ../lib/ablate/gemm_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_gemm_stamps.so ../lib/ablate/gemm_stamps.o ../lib/obj/linear_xs.o ../lib/obj/attention.o ../lib/obj/norm.o ../lib/obj/fusion.o ../lib/obj/elementwise.o ../lib/obj/plan.o ../lib/obj/builder.o ../lib/obj/gemm_conv8p.o
