#!/bin/bash
# Same-box A/B of one environment switch on the pipeline (batch 1 value + batch 8 throughput_mode), alternated twice.
#   bash tools/ab_env_bench.sh ES_GN_FOLD 0 1
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3
out=gpurun_out/ab_${VAR}.txt
: > $out
for rep in 1 2; do
  for v in $A $B; do
    env $VAR=$v python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-stress-mode --no-native-abi --throughput-sweep "" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 - "$VAR=$v rep $rep" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], "batch 1:", d["value"], "images/s", d["ms_per_step"], "ms;  batch 8:", d["throughput_mode"]["value"], "images/s", d["throughput_mode"]["ms_per_step"], "ms", flush=True)
PY
  done
done
cat $out
