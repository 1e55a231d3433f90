#!/bin/bash
# Round-5 development lease: the whole -m gpu suite (with the prints of the full-size parity tests), VAE A/B of this round's GroupNorm changes,
# ONE-lane per-direction VAE kernel traces, GroupNorm probe, the windows leg, and the end-to-end reference-dtype calibration of the 50-step fixtures.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05b
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s --maxfail=6 -p no:cacheprovider --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -h "\[fullsize\]\|\[attention\]\|passed\|failed\|Error\|FAILED" $O/pytest.log | cut -c1-1500 | tail -40
timeout 200 python tools/gpu_vae_bench.py --out $O/vae_fused.json > $O/vae_fused.log 2>&1
AETHER_VAE_GN_TWO_LAUNCH=1 timeout 200 python tools/gpu_vae_bench.py --out $O/vae_two_launch.json > $O/vae_two_launch.log 2>&1
AETHER_GN_APPLY_FIXED_BLOCK=1 timeout 200 python tools/gpu_vae_bench.py --out $O/vae_fixed_block.json > $O/vae_fixed_block.log 2>&1
timeout 200 python tools/gpu_vae_bench.py --lanes 1 --out $O/vae_one_lane.json > $O/vae_one_lane.log 2>&1
grep -h seconds $O/vae_*.log | cut -c1-200
for d in encode decode; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vae1_$d/trace -- python tools/gpu_vae_bench.py --lanes 1 --only $d --reps 2 --out $O/vae1_$d.json > $O/vae1_$d.log 2>&1
  python tools/summarize_rocprof.py $O/vae1_$d $O/vae1_${d}_summary > /dev/null 2>&1
done
timeout 200 python tools/gpu_gn_probe.py > $O/gn_probe.log 2>&1; tail -8 $O/gn_probe.log | cut -c1-400
timeout 400 python bench.py --windows > $O/windows.json 2> $O/windows.err; tail -c 1200 $O/windows.json
timeout 900 python tools/make_fullsize_golden_gpu.py calib_full > $O/calib_full.log 2>&1; grep calib_full $O/calib_full.log | cut -c1-300
