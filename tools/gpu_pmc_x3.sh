#!/bin/bash
mkdir -p gpurun_out/pmcx3
R=$GRAFT_REPO_ROOT
export SRADSGAN_CONV_MATH=bf16x3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $R/gpurun_out/pmcx3/p1 -o p1 --output-format csv -- python3 $R/tools/prof_conv_x3.py 5 > $R/gpurun_out/pmcx3/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcx3/p2 -o p2 --output-format csv -- python3 $R/tools/prof_conv_x3.py 5 > $R/gpurun_out/pmcx3/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_CVT SQ_WAVES -d $R/gpurun_out/pmcx3/p3 -o p3 --output-format csv -- python3 $R/tools/prof_conv_x3.py 5 > $R/gpurun_out/pmcx3/p3.log 2>&1
cd $R
for p in p1 p2 p3; do f=$(find gpurun_out/pmcx3/$p -name "*counter_collection.csv" | head -1); python tools/pmc_summary.py $f; done
tail -2 gpurun_out/pmcx3/p1.log
