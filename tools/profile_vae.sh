#!/bin/bash
# rocprofv3 evidence for the VAE (run on the MI355X box through gpurun): ONE-lane launch plan (kernels of a call do not overlap, so durations and counters
# attribute per kernel), full-size encode + decode (tools/gpu_vae_bench.py).
#   1. kernel trace + stats;  2. PMC counters in SEPARATE passes (gpurun refuses --pmc combined with trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Output: gpurun_out/<name>/{trace,pmc_*}; summary JSON/MD next to it (copy into profiles/).
set -u
NAME=${1:-prof_vae}
OUT=gpurun_out/$NAME
export TMPDIR=/tmp
mkdir -p $OUT
CMD="python tools/gpu_vae_bench.py --lanes 1 --reps 1 --out $OUT/bench.json"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $CMD > $OUT/pmc_l2.log 2>&1
grep -h seconds $OUT/trace.log | cut -c1-160
python tools/summarize_rocprof.py $OUT gpurun_out/${NAME}_summary
