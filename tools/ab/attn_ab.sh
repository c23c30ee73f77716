#!/bin/bash
# GPU box: old two-phase fused attention backward (objects saved under tools/ab) against the build's
set -u
OUT=gpurun_out/attn_ab
mkdir -p $OUT
PKG=ucf-vit_amd
OTHER=$(ls $PKG/build/*.o | grep -v -e attention_short.o -e attention.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_old.so tools/ab/attention_short_old.o tools/ab/attention_old.o $OTHER || exit 1
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -k "attention" > $OUT/test.log 2>&1; tail -5 $OUT/test.log
for rep in 1 2; do
for cfg in "665 197 16 64" "1330 197 12 64" "1002 50 16 64" "665 196 16 64" "64 256 16 64"; do
  echo "== old $cfg" >> $OUT/ab.txt; UCFVIT_HIP_LIB=/tmp/lib_old.so timeout -k 10 120 python tools/attn_bench.py $cfg 2>&1 | grep "bwd" >> $OUT/ab.txt
  echo "== new $cfg" >> $OUT/ab.txt; timeout -k 10 120 python tools/attn_bench.py $cfg 2>&1 | grep "bwd" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
