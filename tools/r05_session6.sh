#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== vae decode pmc"; bash tools/collect_vae_decode_pmc.sh > gpurun_out/r05_vae_pmc.log 2>&1; echo "rc $?"; tail -70 gpurun_out/r05_vae_pmc.log
echo "== config4 tests"; timeout -k 10 900 python3 -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "config4" > gpurun_out/r05_config4_tests.txt 2>&1; echo "rc $?"; tail -15 gpurun_out/r05_config4_tests.txt
