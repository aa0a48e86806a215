#!/bin/bash
# skinny (weight-streaming) 3x3 launches of a batch-1 step: the planner's plan, product library against ablation builds of gemm_conv.hip
cd "$(dirname "$0")/.."
out=gpurun_out/skinny_ablate.txt
: > $out
echo "== product" >> $out; python3 tools/skinny_bench.py >> $out 2>&1
for n in 1 16 32 48 64; do
  echo "== ES_ABLATE=$n (1 no MFMA, 16 no activation DMA, 32 no weight DMA, 48 neither, 64 every DMA out of range)" >> $out
  ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_abl$n.so python3 tools/skinny_bench.py >> $out 2>&1
done
cat $out
