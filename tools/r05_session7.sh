#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== full gpu suite"; timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.txt 2>&1; echo "rc $?"; tail -8 gpurun_out/r05_gpu_suite.txt
echo "== default bench"; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default_try.json 2> gpurun_out/r05_bench_default_try.err; echo "rc $?"; tail -c 1500 gpurun_out/r05_bench_default_try.err
