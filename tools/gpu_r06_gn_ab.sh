#!/bin/bash
mkdir -p gpurun_out/r06g
for rep in 1 2; do for lanes in 2 1; do
  python tools/gpu_vae_bench.py --reps 5 --lanes $lanes --out gpurun_out/r06g/vae_bwd_l${lanes}_$rep.json > gpurun_out/r06g/vae_bwd_l${lanes}_$rep.log 2>&1
  AETHER_GN_FORWARD=1 python tools/gpu_vae_bench.py --reps 5 --lanes $lanes --out gpurun_out/r06g/vae_fwd_l${lanes}_$rep.json > gpurun_out/r06g/vae_fwd_l${lanes}_$rep.log 2>&1
done; done
for f in gpurun_out/r06g/vae_*.log; do echo $f; grep -h seconds $f | cut -c1-90; done
