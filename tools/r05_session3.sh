#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in xs_mix_s; do
  echo "== stamps $v"; ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_$v.so timeout -k 10 120 python3 tools/xs_stamps.py > gpurun_out/r05_xs_stamps_$v.txt 2>&1; echo "rc $?"
  sed -n '/N=960/,/wg1 early/p' gpurun_out/r05_xs_stamps_$v.txt | head -22
done
for v in xs_mix xs_mix1; do
  echo "== pp bench $v"; ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_$v.so timeout -k 10 200 python3 tools/xs_pp_bench.py > gpurun_out/r05_xs_pp_bench_$v.txt 2>&1; echo "rc $?"
  head -8 gpurun_out/r05_xs_pp_bench_$v.txt
done
