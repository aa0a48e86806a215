#!/bin/bash
# On the GPU box: everything profiles/ holds for a round, into gpurun_out/final/ (copy what is to be judged to profiles/).
#   bash tools/collect_profiles.sh r05 [a|b|c]      (three parts, so that each fits one gpurun call; no argument: all of them)
cd "$(dirname "$0")/.."
R=${1:-r05}
PART=${2:-all}
out=gpurun_out/final
mkdir -p $out
export TMPDIR=/tmp
if [ "$PART" = a ] || [ "$PART" = all ]; then
# 1. the default bench line (driver's flags)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${R}_bench_default_run.json 2> $out/${R}_bench_default_run.err
# 2. rocprofv3 kernel stats at batch 1, batch 8 and BASELINE configs[4] (768 x 768, bf16, batch 4)
bash tools/prof_bench.sh ${R}_bench_b1 --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline
bash tools/prof_bench.sh ${R}_bench_b8 --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline
ES_PROF_TIMEOUT=400 bash tools/prof_bench.sh ${R}_bench_768_b4 --resolution 768 --dtype bf16 --batch 4 --steps 1 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-stress-mode --no-roofline --no-native-abi
cp gpurun_out/${R}_bench_b1_kernel_stats.csv gpurun_out/${R}_bench_b8_kernel_stats.csv gpurun_out/${R}_bench_768_b4_kernel_stats.csv \
   gpurun_out/${R}_bench_b1_under_rocprof.json gpurun_out/${R}_bench_b8_under_rocprof.json gpurun_out/${R}_bench_768_b4_under_rocprof.json $out/ 2>/dev/null
# 4. clock / power while the loop runs
python3 tools/clock_probe.py > $out/${R}_clock_power_probe.txt 2>&1 || true
python3 tools/clock_probe.py --attn > $out/${R}_clock_power_probe_attention.txt 2>&1 || true
python3 tools/clock_probe.py --pipeline > $out/${R}_clock_power_probe_pipeline_b1.txt 2>&1 || true
python3 tools/clock_probe.py --pipeline --batch 8 > $out/${R}_clock_power_probe_pipeline_b8.txt 2>&1 || true
fi
if [ "$PART" = b ] || [ "$PART" = all ]; then
# 3. HBM-side traffic of the step's GEMM launches (PMC, real pipeline): batch 1, batch 8, configs[4]
bash tools/collect_traffic_pipeline.sh 1 > $out/traffic_b1.log 2>&1
bash tools/collect_traffic_pipeline.sh 8 > $out/traffic_b8.log 2>&1
bash tools/collect_traffic_pipeline.sh 4 768_b4 --resolution 768 --dtype bf16 > $out/traffic_768_b4.log 2>&1
for t in b1 b8 768_b4; do
  cp gpurun_out/traffic_$t/traffic.json $out/${R}_gemm_pmc_traffic_$t.json
  cp gpurun_out/traffic_$t/launches.json $out/${R}_gemm_step_launches_$t.json
done
# 3c. the VAE decode of configs[4], op by op
bash tools/collect_vae_decode_pmc.sh > $out/vae_pmc.log 2>&1; cp gpurun_out/vae_decode_pmc/traffic.json $out/${R}_vae_decode_pmc_768.json
fi
if [ "$PART" = c ] || [ "$PART" = all ]; then
# 3b. matrix-core utilisation and wave states of the step's kernels (PMC, real pipeline)
bash tools/collect_mfma_util.sh 1 > $out/mfma_b1.log 2>&1; cp gpurun_out/mfma_b1/mfma_util.json $out/${R}_mfma_util_b1.json
bash tools/collect_mfma_util.sh 8 > $out/mfma_b8.log 2>&1; cp gpurun_out/mfma_b8/mfma_util.json $out/${R}_mfma_util_b8.json
bash tools/collect_wave_states.sh 1 > $out/waves_b1.log 2>&1; cp gpurun_out/waves_b1/wave_states.json $out/${R}_wave_states_b1.json
bash tools/collect_wave_states.sh 8 > $out/waves_b8.log 2>&1; cp gpurun_out/waves_b8/wave_states.json $out/${R}_wave_states_b8.json
# 5. micro-benchmarks quoted in DESIGN.md
python3 tools/attn_bench.py > $out/${R}_attn_bench.txt 2>&1 || true
python3 tools/norm_bench.py > $out/${R}_norm_bench.txt 2>&1 || true
python3 tools/fusion_bench.py > $out/${R}_fusion_bench.txt 2>&1 || true
fi
ls -la $out
