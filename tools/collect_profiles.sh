#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 evidence for the three BASELINE workloads of bench.py.  Output under gpurun_out/prof/<tag>/...
#   tools/collect_profiles.sh <tag> ["workload ..."]           then locally: python tools/prof_summarize.py gpurun_out/prof/<tag> profiles/<round>_<tag>
# Counter passes are separate from each other (TCC: FETCH_SIZE 3 of 4 slots, WRITE_SIZE 2; SQ: 8 slots) and carry only --kernel-trace;
# the program follows `--` directly (no env / bash -c hop under the profiler).
set -u
TAG=${1:-x}
WLS=${2:-"vit_l16_224 mae_vit_l16_224 unetr_enc_512x512x128 unetr_512x512x128"}     # second argument: subset of workloads (a call is limited to 20 minutes)
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
B=$PWD/bench.py
run() { echo "== $*" >> $OUT/log.txt; timeout -k 10 400 "$@" >> $OUT/log.txt 2>&1 || echo "FAILED rc=$? : $*" >> $OUT/log.txt; }
for WL in $WLS; do
  run rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o run -- python3 $B --workload $WL --steps 4 --warmup 2 --no-cpu-baseline
done
for WL in $WLS; do
  for C in FETCH_SIZE WRITE_SIZE; do
    run rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$WL/$C -o run -- python3 $B --workload $WL --steps 2 --warmup 1 --no-cpu-baseline
  done
  run rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq_$WL/wave -o run -- python3 $B --workload $WL --steps 2 --warmup 1 --no-cpu-baseline
  run rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_$WL/mfma -o run -- python3 $B --workload $WL --steps 2 --warmup 1 --no-cpu-baseline
done
# the un-profiled bench lines of the same build, same box
for WL in $WLS; do
  timeout -k 10 400 python3 $B --workload $WL > $OUT/bench_$WL.json 2>> $OUT/log.txt
done
ls -R $OUT | head -60
