#!/bin/bash
OUT=gpurun_out/r03b3
mkdir -p $OUT
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "dpm" 2>&1 | tail -4
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03b3/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','mfma_frac_whole_step','kernel_ms_per_step','kernel_tflops','gpu_kernel_ms_per_step_total','attention_paths','cfg_step','clip','windows','vae'):
    print(k, json.dumps(d.get(k)))
P
