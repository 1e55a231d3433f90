#!/bin/bash
# round 6: LDS-staged epilogue — correctness (kernel + VAE + DiT suites), then timings (conv probe, VAE, DiT step)
mkdir -p gpurun_out/r06c
python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py tests/test_dit_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06c/tests.log
python tools/gpu_conv_probe.py > gpurun_out/r06c/conv_probe.jsonl 2>&1
for lanes in 2 1; do python tools/gpu_vae_bench.py --reps 5 --lanes $lanes --out gpurun_out/r06c/vae_l$lanes.json > gpurun_out/r06c/vae_l$lanes.log 2>&1; done
python tools/gpu_microbench.py --only gemm --out gpurun_out/r06c/microbench_gemm.json > gpurun_out/r06c/microbench_gemm.log 2>&1
python bench.py --steps 10 --warmup 3 --no-clip --no-cpu-baseline --no-extra-legs > gpurun_out/r06c/bench_step.json 2> gpurun_out/r06c/bench_step.err
tail -4 gpurun_out/r06c/tests.log; cut -c1-220 gpurun_out/r06c/conv_probe.jsonl; grep -h seconds gpurun_out/r06c/vae_l*.log | cut -c1-150
grep -h "kernel" gpurun_out/r06c/microbench_gemm.log | cut -c1-200
python - <<'P'
import json
l=json.loads([x for x in open('gpurun_out/r06c/bench_step.json') if x.startswith('{')][-1])
print(l['value'], l['ms_per_step'], l['kernel_tflops'], l['kernel_ms_per_step'])
P
