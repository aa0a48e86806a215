#!/bin/bash
# On the GPU box: HBM-side traffic of the step's GEMM launches, measured on the real pipeline (eager launches) under
# rocprofv3 --pmc, one counter per pass.   bash tools/collect_traffic_pipeline.sh [batch]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# bash tools/collect_traffic_pipeline.sh [batch] [tag] [extra bench.py arguments, e.g. --resolution 768 --dtype bf16]
B=${1:-1}
TAG=${2:-b$B}
shift; shift
EXTRA="$@"
export ES_TRAFFIC_WORKLOAD="batch $B${EXTRA:+ ($EXTRA)}"
out=gpurun_out/traffic_$TAG
rm -rf $out; mkdir -p $out
ES_DUMP_GEMM=1 ES_DUMP_GEMM_NAME=traffic_$TAG/launches.json python3 bench.py --batch $B $EXTRA --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-stress-mode --no-native-abi > $out/dump.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d $out/$c -o run -- python3 bench.py --batch $B $EXTRA --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-throughput-mode --no-stress-mode --no-native-abi > $out/$c.log 2>&1
  echo "$c pass exit code $?" | tee -a $out/status.txt
done
head -4 $(find $out/FETCH_SIZE -name "*counter_collection.csv") > $out/csv_head.txt; grep -c incr_kernel $(find $out/FETCH_SIZE -name "*counter_collection.csv") >> $out/csv_head.txt
python3 tools/pmc_step_traffic.py $(find $out/FETCH_SIZE -name "*counter_collection.csv") $(find $out/WRITE_SIZE -name "*counter_collection.csv") $out/launches.json $out/traffic.json
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
