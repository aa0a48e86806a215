#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the library (CPU container only: GPU ASan is not available on the pool):
# every .hip file's HOST code is instrumented (-fno-gpu-sanitize keeps the device code as it is), so the dry builds
# (es_load_weights device = -1 / -2), the plan recorder, relocation and the image writer / loader run under the sanitizers.
#   bash tools/asan_build.sh   -> edgestyle_amd/lib/ablate/libes_asan.so
#   LD_PRELOAD=$(tools/asan_build.sh --rt) ASAN_OPTIONS=detect_leaks=0 ES_HIP_LIB=$PWD/edgestyle_amd/lib/ablate/libes_asan.so python -m pytest tests/test_load_weights_cpu.py
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
if [ "$1" = "--rt" ]; then echo $RT; exit 0; fi
set -e
cd "$(dirname "$0")/../edgestyle_amd/csrc"
mkdir -p ../lib/ablate/obj_asan
OBJS=""
for f in gemm_conv gemm_conv8p linear_xs attention norm fusion elementwise plan builder; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=fast \
     -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -c $f.hip -o ../lib/ablate/obj_asan/$f.o &
  OBJS="$OBJS ../lib/ablate/obj_asan/$f.o"
  if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o ../lib/ablate/libes_asan.so $OBJS
echo built ../lib/ablate/libes_asan.so
