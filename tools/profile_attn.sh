#!/bin/bash
# PMC anatomy of the attention kernel variants (separate passes; SQ has 8 slots per pass).
set -u
OUT=gpurun_out/${1:-prof_attn}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
for FL in ${FLAGS_LIST:-1 33}; do
  python tools/gpu_attn_probe.py --flags $FL > $OUT/plain_$FL.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmcA_$FL -- python tools/gpu_attn_probe.py --flags $FL --iters 2 > $OUT/pmcA_$FL.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmcB_$FL -- python tools/gpu_attn_probe.py --flags $FL --iters 2 > $OUT/pmcB_$FL.log 2>&1
done
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/plain_*.log $OUT/summary.txt
