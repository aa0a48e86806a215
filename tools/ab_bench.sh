#!/bin/bash
# Same-box A/B of two library builds on the whole pipeline: bash tools/ab_bench.sh <variant .so> [tag]  (product library = "new")
# Alternates new / old twice at batch 1 (10 images) and once at batch 8 (3 calls); prints ms per image / per call.
cd "$(dirname "$0")/.."
OLD=$PWD/${1:-edgestyle_amd/lib/ablate/libes_r4base.so}
F="--no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline --no-native-abi"
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export ES_HIP_LIB=$OLD; else unset ES_HIP_LIB; fi
    echo "batch 1 $v: $(timeout -k 10 200 python bench.py --steps 10 --warmup 3 $F 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
  done
done
for v in new old; do
  if [ $v = old ]; then export ES_HIP_LIB=$OLD; else unset ES_HIP_LIB; fi
  echo "batch 8 $v: $(timeout -k 10 300 python bench.py --batch 8 --steps 3 --warmup 2 $F 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done
