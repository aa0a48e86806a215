#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== xs vs tiled (forced)"; ES_XS_MIN_M=0 timeout -k 10 300 python3 tools/xs_bench.py > gpurun_out/r05_xs_bench.txt 2>&1; echo "rc $?"; cat gpurun_out/r05_xs_bench.txt
