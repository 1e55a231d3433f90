#!/bin/bash
# round 6: the 512x128 single-W-buffer tap-reuse tile against the 384x128 tile of rounds 2-5 (AETHER_CONV3_384=1), same build
set -x
mkdir -p gpurun_out/r06a
python -m pytest tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06a/vae_tests.log
python tools/gpu_conv_probe.py > gpurun_out/r06a/conv_probe_512.jsonl 2>&1
AETHER_CONV3_384=1 python tools/gpu_conv_probe.py > gpurun_out/r06a/conv_probe_384.jsonl 2>&1
for lanes in 2 1; do
  python tools/gpu_vae_bench.py --reps 5 --lanes $lanes --out gpurun_out/r06a/vae_512_l$lanes.json > gpurun_out/r06a/vae_512_l$lanes.log 2>&1
  AETHER_CONV3_384=1 python tools/gpu_vae_bench.py --reps 5 --lanes $lanes --out gpurun_out/r06a/vae_384_l$lanes.json > gpurun_out/r06a/vae_384_l$lanes.log 2>&1
done
tail -3 gpurun_out/r06a/vae_tests.log; cat gpurun_out/r06a/conv_probe_512.jsonl gpurun_out/r06a/conv_probe_384.jsonl | cut -c1-250; grep -h "seconds" gpurun_out/r06a/vae_*_l*.log | cut -c1-160
