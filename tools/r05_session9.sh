#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== A/B product (new) vs gemm_conv noslp (old)"; bash tools/ab_bench.sh edgestyle_amd/lib/ablate/libes_gc_noslp.so 2>&1 | tee gpurun_out/r05_ab_gc_noslp.txt
echo "== A/B product (new) vs attention noslp (old)"; bash tools/ab_bench.sh edgestyle_amd/lib/ablate/libes_at_noslp.so 2>&1 | tee gpurun_out/r05_ab_at_noslp.txt
