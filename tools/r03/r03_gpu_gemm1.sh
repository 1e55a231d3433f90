#!/bin/bash
OUT=gpurun_out/r03b2
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -5
# DiT shapes (bias+GELU epilogue for all: the epilogue class matters little here), one-tile kernel (5) vs persistent (32773)
for shape in "15076 9216 3072" "15076 3072 3072" "15076 12288 3072" "15076 3072 12288" "30152 9216 3072"; do
  set -- $shape
  for fl in 5 32773 5 32773 5 32773; do
    timeout 120 python tools/gpu_gemm_probe.py --flags $fl --iters 20 --M $1 --N $2 --K $3 | tee -a $OUT/gemm_probe.jsonl
  done
done
