#!/bin/bash
# On the GPU box: HBM-side traffic of the 768x768 bf16 VAE decode (BASELINE configs[4]) op by op -> gpurun_out/vae_decode_pmc/traffic.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
out=gpurun_out/vae_decode_pmc
rm -rf $out; mkdir -p $out
python3 tools/vae_decode_pmc.py run $out/ops.json > $out/run.log 2>&1 || { tail -20 $out/run.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/$c -o run -- python3 tools/vae_decode_pmc.py run $out/ops_$c.json > $out/$c.log 2>&1
  echo "$c pass exit code $?" | tee -a $out/status.txt
done
python3 tools/vae_decode_pmc.py sum $(find $out/FETCH_SIZE -name "*counter_collection.csv") $(find $out/WRITE_SIZE -name "*counter_collection.csv") $out/ops.json $out/traffic.json > $out/summary.txt 2>&1
tail -60 $out/summary.txt
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
