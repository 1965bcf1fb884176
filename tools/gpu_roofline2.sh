#!/bin/bash
# Roofline evidence for bench.py's two dominant kernels, both conv arithmetic modes:
# kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of `bench.py --roofline-only`.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/roof2
cd /tmp && export TMPDIR=/tmp
for M in ${MODES:-fp32 bf16x3 half}; do
  O=$R/gpurun_out/roof2/$M
  mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --roofline-only --conv-math $M > $O/kt.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/bench.py --roofline-only --no-sustained --conv-math $M > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/bench.py --roofline-only --no-sustained --conv-math $M > $O/write.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/mfma -o p --output-format csv -- python3 $R/bench.py --roofline-only --no-sustained --conv-math $M > $O/mfma.log 2>&1
  tail -1 $O/kt.log
done
cd $R
python tools/roofline_summary.py gpurun_out/roof2
