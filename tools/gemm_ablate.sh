#!/bin/bash
# Build ablation variants of the GEMM kernel (see ES_ABLATE in gemm_conv.hip) into edgestyle_amd/lib/ablate/.
# Usage (here): tools/gemm_ablate.sh 1 2 6 ...   then on the GPU box: ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_ablN.so python tools/gemm_bench.py ...
#   (tools/skinny_bench.py for the weight-streaming launches of the deep levels: profiles/r05_skinny_ablate.txt)
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  bash tools/build_variant.sh abl$n gemm_conv.hip "-DES_ABLATE=$n -Wno-uninitialized"
done
