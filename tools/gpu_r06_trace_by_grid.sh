#!/bin/bash
# one-lane encode / decode kernel traces, summarized per (kernel, workgroups per launch): which SHAPES the time goes to
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
for dir in encode decode; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06b/trace_$dir -- python tools/gpu_vae_bench.py --lanes 1 --reps 1 --only $dir --out gpurun_out/r06b/bench_$dir.json > gpurun_out/r06b/trace_$dir.log 2>&1
  python tools/summarize_trace_by_grid.py gpurun_out/r06b/trace_$dir gpurun_out/r06b/by_grid_$dir.md > /dev/null
  find gpurun_out/r06b/trace_$dir -name "*.csv" -size +20M -delete
done
head -60 gpurun_out/r06b/by_grid_encode.md; head -70 gpurun_out/r06b/by_grid_decode.md
