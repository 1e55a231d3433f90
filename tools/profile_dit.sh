#!/bin/bash
# rocprofv3 evidence for bench.py's numbers (run on the MI355X box through gpurun):
#   1. kernel trace + stats of the same command bench.py's roofline comes from (2 timed steps);
#   2. PMC counters in SEPARATE passes (gpurun refuses --pmc combined with trace domains; FETCH_SIZE and WRITE_SIZE do not
#      fit one pass: /opt/skills/guides/MI355X_MICROARCH.md §rocprofv3 PMC slots), one forward each.
# Output: gpurun_out/<name>/{trace,pmc_*}; summarise with  python tools/summarize_rocprof.py gpurun_out/<name> profiles/<prefix>
set -u
NAME=${1:-prof_dit}
OUT=gpurun_out/$NAME
export TMPDIR=/tmp
BENCH="python bench.py --no-clip --no-cpu-baseline --no-extra-legs"
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 2 --warmup 1 > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- $BENCH --steps 1 --warmup 0 > $OUT/pmc_l2.log 2>&1
grep -h '"metric"' $OUT/trace.log | cut -c1-400
python tools/summarize_rocprof.py $OUT gpurun_out/${NAME}_summary
