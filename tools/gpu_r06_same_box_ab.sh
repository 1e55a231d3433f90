#!/bin/bash
# Same-box A/B of the round-5 library (tools/ab/libaether_hip_r05.so, built from commit 5428a84; same C ABI) against the round-6 library.
# Build the old library first, in the build container (the .so is git-ignored and travels with the gpurun snapshot):
#   git worktree add /tmp/r05 5428a84 && (cd /tmp/r05 && python -m aether_amd.build --force) && mkdir -p tools/ab && cp /tmp/r05/aether_amd/csrc/libaether_hip.so tools/ab/libaether_hip_r05.so
# Then:
# DiT step (bench.py --no-clip --no-extra-legs) and the VAE (tools/gpu_vae_bench.py, two lanes), interleaved twice.
OUT=gpurun_out/r06ab
mkdir -p $OUT
cp aether_amd/csrc/libaether_hip.so $OUT/libaether_hip_r06.so
for rep in 1 2; do for tag in r05 r06; do
  if [ $tag = r05 ]; then cp tools/ab/libaether_hip_r05.so aether_amd/csrc/libaether_hip.so; else cp $OUT/libaether_hip_r06.so aether_amd/csrc/libaether_hip.so; fi
  touch aether_amd/csrc/libaether_hip.so
  python bench.py --steps 10 --warmup 3 --no-clip --no-cpu-baseline --no-extra-legs > $OUT/step_${tag}_$rep.json 2> $OUT/step_${tag}_$rep.err
  python tools/gpu_vae_bench.py --reps 5 --lanes 2 --out $OUT/vae_${tag}_$rep.json > $OUT/vae_${tag}_$rep.log 2>&1
done; done
cp $OUT/libaether_hip_r06.so aether_amd/csrc/libaether_hip.so; rm $OUT/libaether_hip_r06.so
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06ab/step_*.json')):
    l=json.loads([x for x in open(f) if x.startswith('{')][-1]); print(f.split('/')[-1], round(l['value'],3), round(l['ms_per_step'],1), l['kernel_tflops'])
for f in sorted(glob.glob('gpurun_out/r06ab/vae_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], {k:round(v['seconds'],4) for k,v in d.items() if isinstance(v,dict)})
P
