#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "(57344,320,2560,g) (14336,640,5120,g) (458752,320,2560,g) (114688,640,5120,g) (57344,320,960) (8192,320,2560,g): us [checksum]" > gpurun_out/r05_xs_geglu_ab.txt
for rep in 1 2; do
for v in "" xs_v2 xs_v2noslp xs_noslp; do
  if [ -z "$v" ]; then unset ES_HIP_LIB; else export ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_$v.so; fi
  timeout -k 10 200 python3 tools/xs_geglu_ab.py 2>/dev/null >> gpurun_out/r05_xs_geglu_ab.txt
done
done
cat gpurun_out/r05_xs_geglu_ab.txt
