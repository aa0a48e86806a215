#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== xs pp bench"; timeout -k 10 300 python3 tools/xs_pp_bench.py > gpurun_out/r05_xs_pp_bench.txt 2>&1; echo "rc $?"
echo "== xs pp bench big"; timeout -k 10 300 python3 tools/xs_pp_bench.py --big >> gpurun_out/r05_xs_pp_bench.txt 2>&1; echo "rc $?"
echo "== xs tests"; timeout -k 10 400 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "xs or linear or grouped" > gpurun_out/r05_xs_tests.txt 2>&1; echo "rc $?"
cat gpurun_out/r05_xs_pp_bench.txt; tail -5 gpurun_out/r05_xs_tests.txt
