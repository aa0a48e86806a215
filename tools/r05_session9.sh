#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== A/B product (new) vs all_noslp (old)"; bash tools/ab_bench.sh edgestyle_amd/lib/ablate/libes_all_noslp.so 2>&1 | tee gpurun_out/r05_ab_all_noslp.txt
echo "== A/B product (new) vs xs-only noslp (old)"; bash tools/ab_bench.sh edgestyle_amd/lib/ablate/libes_xs_noslp2.so 2>&1 | tee gpurun_out/r05_ab_xs_noslp.txt
