#!/bin/bash
# rocprofv3 kernel trace of the full-size VAE encode + decode (tools/gpu_vae_bench.py: 1 warm-up + 2 timed passes each).
# Output: gpurun_out/<name>/trace; summary JSON/MD next to it (copy into profiles/).
set -u
NAME=${1:-prof_vae}
OUT=gpurun_out/$NAME
export TMPDIR=/tmp
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/gpu_vae_bench.py > $OUT/trace.log 2>&1
grep -h seconds $OUT/trace.log | cut -c1-160
python tools/summarize_rocprof.py $OUT gpurun_out/${NAME}_summary
