#!/bin/bash
OUT=gpurun_out/r03b8
mkdir -p $OUT
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_demo_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py -x -q -s -k "not cfg_parallel" > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep "fullsize\]" $OUT/tests.log | sort -u | tail -3; tail -3 $OUT/tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r03b8/bench_full.json'):
    if l.startswith('{'):
        d=json.loads(l)
        for k in ('value','ms_per_step','mfma_frac_whole_step','kernel_tflops','cfg_step','clip','windows','vae'):
            print(k, json.dumps(d.get(k)))
P
