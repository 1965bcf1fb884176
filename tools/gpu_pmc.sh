#!/bin/bash
# PMC passes on the dominant conv kernels (separate runs, no trace domains mixed in)
mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $R/gpurun_out/pmc/p1 -o p1 --output-format csv -- python3 $R/tools/prof_conv.py 32 5 > $R/gpurun_out/pmc/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc/p2 -o p2 --output-format csv -- python3 $R/tools/prof_conv.py 32 5 > $R/gpurun_out/pmc/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc/p3 -o p3 --output-format csv -- python3 $R/tools/prof_conv.py 32 5 > $R/gpurun_out/pmc/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc/p4 -o p4 --output-format csv -- python3 $R/tools/prof_conv.py 32 5 > $R/gpurun_out/pmc/p4.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmc/kt -o kt --output-format csv -- python3 $R/tools/prof_conv.py 32 20 > $R/gpurun_out/pmc/kt.log 2>&1
cd $R; find gpurun_out/pmc -name "*.csv" | head -20; tail -3 gpurun_out/pmc/p1.log
