#!/bin/bash
# Build ablation variants of the GEMM kernel (see ES_ABLATE in gemm_conv.hip) into edgestyle_amd/lib/ablate/.
# Usage (here): tools/gemm_ablate.sh 1 2 6 ...   then on the GPU box: ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_ablN.so python tools/gemm_bench.py ...
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
mkdir -p ../lib/ablate
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-uninitialized -ffp-contract=fast"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DES_ABLATE=$n -c gemm_conv.hip -o ../lib/ablate/gemm_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ablate/libes_abl$n.so ../lib/ablate/gemm_abl$n.o \
      ../lib/obj/linear_xs.o ../lib/obj/attention.o ../lib/obj/norm.o ../lib/obj/fusion.o ../lib/obj/elementwise.o ../lib/obj/plan.o
  echo built abl$n
done
