#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round-3 GEMM evidence that is not a rocprofv3 profile.  Output under gpurun_out/r3f/, copied into profiles/r03_c_*.
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
{
  echo "tools/block_gemm_bench.py 665 1024 5 4 on one MI355X, same gpurun call: UCFVIT_GEMM_STAGGER=0 (every launch on gemm3_kernel), then the build's default"
  echo "(staggered kernel for the plain / GELU launches and the K = 4096 residual launch; fc2 dgrad with column sums and the K = 1024 residual launch stay on gemm3_kernel), twice each."
  for v in 0 1 0 1; do
    if [ $v = 0 ]; then export UCFVIT_GEMM_STAGGER=0; else unset UCFVIT_GEMM_STAGGER; fi
    echo; echo "== UCFVIT_GEMM_STAGGER=${UCFVIT_GEMM_STAGGER:-unset (default)}"
    SKIP_CHECK=1 timeout -k 10 200 python tools/block_gemm_bench.py 665 1024 5 4 2>&1 | grep -v amdgpu.ids
  done
} > $OUT/block_gemm_b665.txt
unset UCFVIT_GEMM_STAGGER
{
  echo "epilogue steps E of the staggered kernel (UCFVIT_GEMM_STAGGER=E forces it for every launch it can run, incl. the K = 1024 residual one); tools/block_gemm_bench.py 665 1024 3 4"
  for e in 1 2 4 8; do
    echo; echo "== E=$e"
    UCFVIT_GEMM_STAGGER=$e SKIP_CHECK=1 timeout -k 10 200 python tools/block_gemm_bench.py 665 1024 3 4 2>&1 | grep -E "fwd|dgrad|total"
  done
} > $OUT/stagger_epilogue_steps.txt
for w in qkv fc1; do
  for e in 1 2; do
    UCFVIT_GEMM_STAGGER=$e timeout -k 10 200 python tools/stagger_stamps.py $w 2>&1 | grep -v amdgpu.ids > $OUT/stagger_stamps_${w}_e$e.txt
  done
done
{
  echo "tools/dma_rate.hip: LDS-DMA delivery rate of every CU at once (256 workgroups, nothing else running), shader clocks per step; window = bytes each workgroup re-reads"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/dma_rate.hip -o /tmp/dma_rate 2>/dev/null
  for r in 32 128 512; do timeout -k 10 120 /tmp/dma_rate $r; echo; done
} > $OUT/dma_rate.txt
timeout -k 10 300 python bench.py --batch 166 --no-cpu-baseline > $OUT/bench_vitl16_b166.json 2>/dev/null
timeout -k 10 300 python bench.py --workload unetr_enc_512x512x128 > $OUT/bench_unetr_enc_b2.json 2>/dev/null
timeout -k 10 300 python bench.py --workload vit_l16_adaptive196 --no-cpu-baseline > $OUT/bench_vitl16_adaptive196_b665.json 2>/dev/null
ls -la $OUT
