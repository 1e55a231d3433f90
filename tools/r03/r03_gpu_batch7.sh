#!/bin/bash
OUT=gpurun_out/r03b7
mkdir -p $OUT
timeout 300 python -m pytest tests/test_vae_gpu.py -x -q -k "decode_pair" 2>&1 | tail -3
timeout 600 python tools/gpu_decode_pair.py 2>/dev/null | tee $OUT/decode_pair.json
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "bench rc=$?"
python - <<'P'
import json
for l in open('gpurun_out/r03b7/bench_full.json'):
    if l.startswith('{'):
        d=json.loads(l)
        for k in ('value','ms_per_step','mfma_frac_whole_step','kernel_ms_per_step','kernel_tflops','attention_paths','cfg_step','clip','windows','vae','roofline','cpu_baseline'):
            print(k, json.dumps(d.get(k)))
P
