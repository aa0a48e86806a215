#!/bin/bash
# On the GPU box: rocprofv3 kernel trace + stats of a short bench run, condensed to gpurun_out/<name>_kernel_stats.csv
#   bash tools/prof_bench.sh <name> [bench args...]
# The run's own log is kept as gpurun_out/prof_<name>.log in every case; when the profiled command fails (round 3: a SIGSEGV of
# the profiled process that a later run's log overwrote) it is ALSO copied to gpurun_out/prof_<name>.FAILED.<unix time>.log, which
# nothing overwrites, and the script exits with the command's code.  -X faulthandler: a fatal signal prints the Python stack too.
cd "$(dirname "$0")/.."
name=$1; shift
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf gpurun_out/prof_$name
timeout -k 10 ${ES_PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -o run -- python3 -X faulthandler bench.py "$@" > gpurun_out/prof_$name.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then
  keep=gpurun_out/prof_$name.FAILED.$(date +%s).log
  cp gpurun_out/prof_$name.log $keep
  echo "prof_bench.sh: the profiled command exited with $rc; its log is kept as $keep" >&2
  tail -60 $keep >&2
  exit $rc
fi
f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py "$f" 60 > gpurun_out/${name}_kernel_stats.csv
grep '^{' gpurun_out/prof_$name.log > gpurun_out/${name}_under_rocprof.json || true
rm -rf gpurun_out/prof_$name
