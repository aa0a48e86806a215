#!/bin/bash
# A/B of a tool-only library build against the product one, per GEMM launch of a batch-1 step (in-kernel stamps):
#   bash tools/ab_lib.sh edgestyle_amd/lib/ablate/libes_wnt.so   -> gpurun_out/gemm_launches_{base,variant}.json
cd "$(dirname "$0")/.."
VAR=${1:-edgestyle_amd/lib/ablate/libes_wnt.so}
for v in base variant; do
  if [ $v = variant ]; then export ES_HIP_LIB=$PWD/$VAR; else unset ES_HIP_LIB; fi
  ES_DUMP_GEMM=1 ES_DUMP_GEMM_NAME=gemm_launches_$v.json python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-throughput-mode --no-stress-mode > gpurun_out/bench_$v.json 2>/dev/null
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/gemm_launches_base.json')); b=json.load(open('gpurun_out/gemm_launches_variant.json'))
assert len(a)==len(b)
rows=[]
for i,(x,y) in enumerate(zip(a,b)):
    g=x['geom']; rows.append((y['seconds']/x['seconds'], x['seconds']*1e6, y['seconds']*1e6, i, g['N'],g['Hout'],g['C1']+g['C2'],g['cout'],g['k'],g['splitk'],g['bn']))
print('total us base %.0f variant %.0f'%(sum(r[1] for r in rows), sum(r[2] for r in rows)))
print('ratio  base_us  var_us  idx  N  H  Cin cout k sk bn   (launches where the variant is >= 3 % faster)')
for r in sorted(rows):
    if r[0] < 0.97: print('%.2f %7.1f %7.1f %4d %3d %3d %5d %5d %d %2d %3d'%r)
PY
