#!/bin/bash
OUT=gpurun_out/r03b1
mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -x -q -k "flash" 2>&1 | tail -8
for fl in 4609 8193 10241 8193 10241; do
  python tools/gpu_attn_probe.py --flags $fl --iters 30 | tee -a $OUT/attn_probe3.jsonl
done
python tools/gpu_attn_probe.py --flags 10241 --iters 30 --gain 10 | tee -a $OUT/attn_probe3.jsonl
