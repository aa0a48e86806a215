#!/bin/bash
# Round-5 GPU session 1: where linear_xs spends its cycles (stamps + ablations), the new bench fields, kernel stats at 768x768.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== xs stamps"; ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_xs_stamps.so timeout -k 10 120 python3 tools/xs_stamps.py > gpurun_out/r05_xs_stamps.txt 2>&1; echo "rc $?"
echo "== xs ablate"; timeout -k 10 400 python3 tools/xs_ablate_run.py > gpurun_out/r05_xs_ablate.txt 2>&1; echo "rc $?"
echo "== bench"; timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_try1.json 2> gpurun_out/r05_bench_try1.err; echo "rc $?"
echo "== prof 768"; ES_PROF_TIMEOUT=280 bash tools/prof_bench.sh r05_bench_768_b4 --resolution 768 --dtype bf16 --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline --no-native-abi; echo "rc $?"
tail -3 gpurun_out/r05_xs_stamps.txt gpurun_out/r05_xs_ablate.txt; tail -c 600 gpurun_out/r05_bench_try1.err
