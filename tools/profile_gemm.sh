#!/bin/bash
# GEMM variants at the DiT shapes (interleaved rounds) + the qkv-prepare / LN kernels through the bench line.
set -u
OUT=gpurun_out/${1:-prof_gemm}
export TMPDIR=/tmp
mkdir -p $OUT
python tools/gpu_microbench.py --only gemm --out $OUT/gemm_variants.json > $OUT/gemm_variants.log 2>&1
grep -h "kernel" $OUT/gemm_variants.log | cut -c1-220
