#!/bin/bash
# GPU box: the committed bench lines of the round (the PMC traffic JSONs of the same source hash are in profiles/ when this runs, so
# roofline.traffic is a measured number).  Output gpurun_out/final/bench_<workload>.json
set -u
OUT=gpurun_out/final
mkdir -p $OUT
for WL in vit_l16_224 vit_b16_224 vit_l16_adaptive196 mae_vit_l16_224 unetr_enc_512x512x128 unetr_512x512x128; do
  timeout -k 10 500 python bench.py --workload $WL > $OUT/bench_$WL.json 2>> $OUT/err.txt || echo "FAILED $WL" >> $OUT/err.txt
done
ls -la $OUT
