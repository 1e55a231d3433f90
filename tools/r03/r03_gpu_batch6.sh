#!/bin/bash
OUT=gpurun_out/r03b6
mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -x -q -s -k "qkv_prep or fused_qkv or qk_norm or gemm" 2>&1 | tail -12
for fl in 0 65536 0 65536; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-clip --no-extra-legs --dit-flags-or $fl 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('flags_or=$fl', round(d['value'],4), round(d['ms_per_step'],2), d['kernel_ms_per_step'])
" | tee -a $OUT/fuse_ab.txt
done
