#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + stats of a short bench run, condensed to gpurun_out/<name>_kernel_stats.csv
#   bash tools/prof_bench.sh <name> [bench args...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o run -- python3 bench.py "$@" > gpurun_out/prof_$name.log 2>&1
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" 60 > gpurun_out/${name}_kernel_stats.csv
grep '^{' gpurun_out/prof_$name.log > gpurun_out/${name}_under_rocprof.json || true
rm -rf gpurun_out/prof_$name
