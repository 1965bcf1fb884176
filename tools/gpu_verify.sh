#!/bin/bash
# Round-end verification: the whole GPU suite, smoke(), and the chain sweep line in split-bf16 arithmetic.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/verify
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py --workload chain --conv-math bf16x3 2>/dev/null | tail -1 > $O/chain_bf16x3.json
cut -c1-400 $O/chain_bf16x3.json
