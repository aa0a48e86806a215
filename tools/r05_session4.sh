#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== kvres tests"; timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r05_attn_tests.txt 2>&1; echo "rc $?"; tail -5 gpurun_out/r05_attn_tests.txt
echo "== xattn bench"; timeout -k 10 300 python3 tools/xattn_bench.py > gpurun_out/r05_xattn_bench.txt 2>&1; echo "rc $?"; cat gpurun_out/r05_xattn_bench.txt
