#!/bin/bash
set -u
OUT=gpurun_out/${1:-prof_gemm_pmc}
export TMPDIR=/tmp
mkdir -p $OUT
python tools/gpu_gemm_probe.py > $OUT/plain.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmcA -- python tools/gpu_gemm_probe.py --iters 2 > $OUT/pmcA.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmcB -- python tools/gpu_gemm_probe.py --iters 2 > $OUT/pmcB.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcC -- python tools/gpu_gemm_probe.py --iters 2 > $OUT/pmcC.log 2>&1
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
grep flags $OUT/plain.log; cat $OUT/summary.txt
