#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "GEGLU M,K,N: (28672,1280,10240) (3584,1280,10240) (512,1280,10240) (57344,320,2560) (458752,320,2560) (14336,640,5120) (2048,640,5120): us [checksum]" > gpurun_out/r05_geglu_ab.txt
for rep in 1 2; do
for v in "" xs_noslp2; do
  if [ -z "$v" ]; then unset ES_HIP_LIB; else export ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_$v.so; fi
  timeout -k 10 200 python3 tools/geglu_ab.py 2>/dev/null >> gpurun_out/r05_geglu_ab.txt
done
done
cat gpurun_out/r05_geglu_ab.txt
