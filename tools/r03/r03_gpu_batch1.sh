#!/bin/bash
# Round-3 GPU batch 1: attention variants (correctness + TF/s at the BASELINE shape) and the full-size parity fixtures.
set -u
OUT=gpurun_out/r03b1
mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -x -q -k flash > $OUT/flash_tests.log 2>&1
echo "flash tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/flash_tests.log | tee -a $OUT/summary.txt
# flags: 1 wide store | 256 interleave | 512 pair pipeline | 2048 dot2 sums | 32 conservative path only
for fl in 513 2561 1 2049 257 2305 33 2081; do
  python tools/gpu_attn_probe.py --flags $fl --iters 20 | tee -a $OUT/attn_probe.jsonl
done
for gain in 3 10; do
  python tools/gpu_attn_probe.py --flags 2561 --iters 20 --gain $gain | tee -a $OUT/attn_probe.jsonl
  python tools/gpu_attn_probe.py --flags 2081 --iters 20 --gain $gain | tee -a $OUT/attn_probe.jsonl
done
[ "${1:-all}" = "attn" ] && exit 0
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -s > $OUT/fullsize_parity.log 2>&1
echo "fullsize parity rc=$?" | tee -a $OUT/summary.txt
grep "\[fullsize\]" $OUT/fullsize_parity.log | tee -a $OUT/summary.txt
tail -5 $OUT/fullsize_parity.log | tee -a $OUT/summary.txt
