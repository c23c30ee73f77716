#!/bin/bash
# Run ON THE GPU BOX: A/B of the non-temporal hint on the GEMM kernels' C / aux stores.  V0: ordinary stores; V1: staggered kernel only;
# V2: staggered + ping-pong kernel (the build's default).  Block GEMM bench and the whole ViT-L/16 step, same box, interleaved.
set -u
OUT=gpurun_out/stmod
mkdir -p $OUT
PKG=ucf-vit_amd
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast"
OTHER=$(ls $PKG/build/*.o | grep -v -e gemm_stagger.o -e gemm2.o)
$CXX '-DS5_ST_MOD=""' -c $PKG/csrc/gemm_stagger.hip -o /tmp/gs_t.o || exit 1
$CXX -DUCFVIT_GEMM_STORE_TEMPORAL -c $PKG/csrc/gemm2.hip -o /tmp/g2_t.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_0.so /tmp/gs_t.o /tmp/g2_t.o $OTHER || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_1.so $PKG/build/gemm_stagger.o /tmp/g2_t.o $OTHER || exit 1
cp $PKG/lib/libucfvit_hip.so /tmp/lib_2.so
for rep in 1 2; do
  for i in 0 1 2; do
    echo "== V$i" >> $OUT/ab.txt
    UCFVIT_HIP_LIB=/tmp/lib_$i.so SKIP_CHECK=1 timeout -k 10 200 python tools/block_gemm_bench.py 665 1024 5 4 2>&1 | grep -E "fwd|dgrad|total" >> $OUT/ab.txt || exit 1
    echo "== V$i" >> $OUT/step.txt
    UCFVIT_HIP_LIB=/tmp/lib_$i.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null >> $OUT/step.txt || exit 1
  done
done
for i in 0 2; do
  echo "== V$i, UCFVIT_GEMM_STAGGER=0" >> $OUT/ab.txt
  UCFVIT_GEMM_STAGGER=0 UCFVIT_HIP_LIB=/tmp/lib_$i.so SKIP_CHECK=1 timeout -k 10 200 python tools/block_gemm_bench.py 665 1024 5 4 2>&1 | grep -E "fwd|dgrad|total" >> $OUT/ab.txt || exit 1
done
grep -E "==|total" $OUT/ab.txt
python - <<'PY'
import json
for l in open("gpurun_out/stmod/step.txt"):
    if l.startswith("=="): print(l.strip(), end=" ")
    elif l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
