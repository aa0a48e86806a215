#!/bin/bash
# On the GPU box: matrix-core utilisation of the step's kernels on the real pipeline (eager launches) under rocprofv3 --pmc.
#   bash tools/collect_mfma_util.sh [batch]      -> gpurun_out/mfma_b<batch>/mfma_util.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
B=${1:-1}
out=gpurun_out/mfma_b$B
rm -rf $out; mkdir -p $out
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -o run -- python3 bench.py --batch $B --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-throughput-mode --no-stress-mode --no-native-abi > $out/pmc.log 2>&1
echo "pmc pass exit code $?" | tee $out/status.txt
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
head -3 "$f" > $out/csv_head.txt
python3 tools/pmc_mfma_util.py "$f" $out/mfma_util.json > $out/summary.txt 2>&1
tail -5 $out/summary.txt
rm -rf $out/pmc
