#!/bin/bash
# On the GPU box: where the waves of the step's kernels spend their cycles (parked / issue-stalled / issuing), real pipeline,
# eager launches under rocprofv3 --pmc.   bash tools/collect_wave_states.sh [batch]   -> gpurun_out/waves_b<batch>/wave_states.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
B=${1:-1}
out=gpurun_out/waves_b$B
rm -rf $out; mkdir -p $out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout -k 10 600 rocprofv3 --pmc $C --output-format csv -d $out/pmc -o run -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-throughput-mode --no-stress-mode --no-native-abi > $out/pmc.log 2>&1
echo "pmc pass exit code $?" | tee $out/status.txt
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 tools/pmc_kernel_counters.py "$f" $out/wave_states.json $C > $out/summary.txt 2>&1
cat $out/summary.txt
rm -rf $out/pmc
