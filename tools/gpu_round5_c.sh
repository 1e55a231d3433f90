#!/bin/bash
# Round-5 second development lease: reconstruction fixtures under device semantics, the VAE / guided / windows tests, GroupNorm non-temporal A/B,
# VAE timing (two lanes, one lane), the windows leg.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05c
mkdir -p $O
AETHER_ORACLE_BUDGET_S=900 timeout 1000 python tools/make_fullsize_golden_gpu.py recon4 recon10 recon50 > $O/recon_device.log 2>&1; tail -4 $O/recon_device.log | cut -c1-200
cp gpurun_out/fixtures/fullsize_recon*_device.npz tests/golden/ 2>/dev/null
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_fullsize_guided_gpu.py tests/test_blend_cpu.py tests/test_drivers_gpu.py tests/test_demo_gpu.py -m gpu -q -s --maxfail=8 -p no:cacheprovider --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -h "\[fullsize\]\|passed\|failed\|^FAILED\|^E  " $O/pytest.log | cut -c1-1600 | tail -30
timeout 200 python tools/gpu_gn_probe.py > $O/gn_probe_default.log 2>&1
AETHER_GN_NT=1 timeout 200 python tools/gpu_gn_probe.py > $O/gn_probe_nt.log 2>&1
for f in default nt; do echo "== $f"; grep shape $O/gn_probe_$f.log | cut -c1-330; done
timeout 200 python tools/gpu_vae_bench.py --out $O/vae_two_lanes.json > $O/vae_two_lanes.log 2>&1
AETHER_GN_NT=1 timeout 200 python tools/gpu_vae_bench.py --out $O/vae_two_lanes_nt.json > $O/vae_two_lanes_nt.log 2>&1
timeout 200 python tools/gpu_vae_bench.py --lanes 1 --out $O/vae_one_lane.json > $O/vae_one_lane.log 2>&1
grep -h seconds $O/vae_*.log | cut -c1-170
timeout 400 python bench.py --windows > $O/windows.json 2> $O/windows.err; python -c "
import json,sys
for ln in open('$O/windows.json'):
    if ln.startswith('{'): print(json.dumps(json.loads(ln)['windows'])[:600])"
