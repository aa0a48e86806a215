#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== gemm/ops tests"; timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py tests/test_load_weights_gpu.py tests/test_native_gpu.py -x -q -m gpu > gpurun_out/r05_bn256_tests2.txt 2>&1; echo "rc $?"; tail -6 gpurun_out/r05_bn256_tests2.txt
F="--no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline --no-native-abi"
for i in 1 2; do for v in 1 0; do
  echo "batch 1 big256=$v: $(ES_BIG_TILE_256=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 $F 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done; done
for i in 1 2; do for v in 1 0; do
  echo "batch 8 big256=$v: $(ES_BIG_TILE_256=$v timeout -k 10 300 python bench.py --batch 8 --steps 3 --warmup 2 $F 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done; done
