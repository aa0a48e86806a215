#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== bn=256 tests"; timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "big_tile or grouped or geglu or ln_fold or layer_norm" > gpurun_out/r05_bn256_tests.txt 2>&1; echo "rc $?"; tail -15 gpurun_out/r05_bn256_tests.txt
echo "== ln256 bench"; timeout -k 10 400 python3 tools/ln256_bench.py > gpurun_out/r05_ln256_bench.txt 2>&1; echo "rc $?"; cat gpurun_out/r05_ln256_bench.txt
