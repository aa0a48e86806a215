#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_attn_stamps.so timeout -k 10 200 python3 tools/attn_stamps.py --pp > gpurun_out/r05_attn40pp_stamps.txt 2>&1; echo "rc $?"; cat gpurun_out/r05_attn40pp_stamps.txt
