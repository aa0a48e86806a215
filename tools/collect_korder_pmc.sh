#!/bin/bash
# On the GPU box: cache / traffic counters of one 3x3 convolution in both K orders.  bash tools/collect_korder_pmc.sh [bn] [N H Cin Cout]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# rocprofiler-sdk's queue interception faults on the HIP runtime's pre-built graph AQL packets: replay graphs packet by packet
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
out=gpurun_out/korder_pmc
rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/counters_available.txt 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $out/p$i -o run -- python3 tools/korder_pmc.py "$@" > $out/p$i.log 2>&1
  echo "pass $i ($C) exit $?" >> $out/status.txt
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "== $C" >> $out/summary.txt; python3 tools/korder_pmc_sum.py "$f" >> $out/summary.txt 2>&1; fi
  rm -rf $out/p$i
done
cat $out/status.txt $out/summary.txt
