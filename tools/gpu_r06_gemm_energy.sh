#!/bin/bash
# GEMM main-loop variants — us, cycles / XCD, effective GHz, LDS / VMEM instruction counts
export TMPDIR=/tmp
OUT=gpurun_out/r06f
mkdir -p $OUT
python tools/gpu_gemm_variants.py --out $OUT/gemm_variants.json > $OUT/variants.log 2>&1
for v in shipped ksps2 snake nt_outer noprio vendor; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$v -- python tools/gpu_gemm_variants.py --pmc-run $v > $OUT/trace_$v.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_$v -- python tools/gpu_gemm_variants.py --pmc-run $v > $OUT/pmc_$v.log 2>&1
done
cat $OUT/variants.log | grep -v Warning
python tools/summarize_gemm_energy.py $OUT $OUT/gemm_energy.txt
find $OUT -name "*.csv" -size +5M -delete
