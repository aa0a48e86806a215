#!/bin/bash
# On the GPU box: dump the GEMM launches of one batch-1 step, replay them stand-alone under rocprofv3 --pmc
# (FETCH_SIZE and WRITE_SIZE in separate passes) and aggregate -> gpurun_out/traffic/*.json
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
mkdir -p gpurun_out/traffic
ES_DUMP_GEMM=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/traffic/dump.log 2>&1
cp gpurun_out/gemm_step_launches.json gpurun_out/traffic/launches.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic/$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/traffic/$c -- python3 tools/gemm_step_traffic.py gpurun_out/traffic/launches.json > gpurun_out/traffic/$c.log 2>&1
  python3 tools/pmc_traffic.py $(find gpurun_out/traffic/$c -name "*counter_collection.csv") > gpurun_out/traffic/$c.json
  rm -rf gpurun_out/traffic/$c
done
python3 - <<'PY'
import json
L = json.load(open("gpurun_out/traffic/launches.json"))
alg = sum(x["geom"]["algorithmic_bytes"] for x in L) / len(L)
print("launches", len(L), "algorithmic bytes/launch", alg, "splitk launches", sum(1 for x in L if x["geom"]["splitk"] > 1),
      "grouped", sum(1 for x in L if x["geom"].get("group_n")))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = json.load(open(f"gpurun_out/traffic/{c}.json"))
    for k, v in d.items():
        print(c, k, {kk: round(vv["per_launch"], 1) for kk, vv in v.items()}, {kk: vv["launches"] for kk, vv in v.items()})
PY
