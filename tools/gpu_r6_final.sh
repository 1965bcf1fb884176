#!/bin/bash
# Round 6: full GPU suite, then the evidence run (tools/gpu_evidence.sh).
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/evidence; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $R/gpurun_out/evidence/pytest_gpu.log 2>&1; tail -3 $R/gpurun_out/evidence/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_evidence.sh
