#!/bin/bash
# VAE correctness (tests/test_vae_gpu.py) + timing with one and two lanes: the loop used for every VAE launch-plan change of round 5.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-vae_check}; mkdir -p $O
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
for lanes in 2 1 2; do timeout 200 python tools/gpu_vae_bench.py --lanes $lanes --out $O/vae_lanes$lanes.json > $O/vae_lanes$lanes.log 2>&1; grep -h seconds $O/vae_lanes$lanes.log | cut -c1-125; done
