#!/bin/bash
OUT=gpurun_out/r03b1
mkdir -p $OUT
for fl in 513 4609 2561 6657 2081 513 4609; do
  python tools/gpu_attn_probe.py --flags $fl --iters 30 | tee -a $OUT/attn_probe2.jsonl
done
python -m pytest tests/test_kernels_gpu.py -x -q -k "flash and (guarded or variants_agree)" 2>&1 | tail -2
