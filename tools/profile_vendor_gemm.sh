#!/bin/bash
# Reference point only: what the vendor GEMM behind torch.nn.functional.linear looks like at the DiT shapes (kernel name = tile /
# wave layout, registers, LDS, cycles, MFMA busy, LDS activity, L2 hit rate, fabric reads), next to aether_gemm_bf16 in the same
# process.  Passes: trace | pmcA | pmcB (or one of them: $2).
set -u
OUT=gpurun_out/${1:-prof_vendor_gemm}
WHAT=${2:-all}
export TMPDIR=/tmp
mkdir -p $OUT
[ $WHAT = all -o $WHAT = trace ] && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/gpu_vendor_gemm_probe.py > $OUT/trace.log 2>&1
[ $WHAT = all -o $WHAT = pmcA ] && timeout 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/pmcA -- python tools/gpu_vendor_gemm_probe.py > $OUT/pmcA.log 2>&1
[ $WHAT = all -o $WHAT = pmcB ] && timeout 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmcB -- python tools/gpu_vendor_gemm_probe.py > $OUT/pmcB.log 2>&1
python tools/summarize_vendor_gemm.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-300
