#!/bin/bash
# Round 6, second call: graph tests after the plane-pool capture fix, the near-tie parity test, the rest of the suite, chain sweep checks.
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6b; mkdir -p $E; cd $R
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -s > $E/pytest_graph.log 2>&1; tail -4 $E/pytest_graph.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "near_tie" > $E/pytest_neartie.log 2>&1; grep "near-tie" $E/pytest_neartie.log; tail -3 $E/pytest_neartie.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_graph_gpu.py > $E/pytest_rest.log 2>&1; tail -6 $E/pytest_rest.log
timeout 600 python bench.py --workload chain --conv-math bf16x3 2>&1 | tail -1 > $E/chain_bf16x3.json; python -c "
import json; d=json.load(open('$E/chain_bf16x3.json')); print(d['value'], {k:(v['img_per_s'],v['peak_mem_gb']) for k,v in d['per_scale'].items()})"
timeout 600 python bench.py --workload chain --conv-math bf16x3 --scales 4 2>&1 | tail -1 > $E/chain_x4_alone.json; python -c "
import json; d=json.load(open('$E/chain_x4_alone.json')); print(d['value'], {k:(v['img_per_s'],v['peak_mem_gb']) for k,v in d['per_scale'].items()})"
timeout 600 python bench.py --workload chain --conv-math bf16x3 --scales 4 --spinup-steps 0 2>&1 | tail -1 > $E/chain_x4_alone_nospin.json; python -c "
import json; d=json.load(open('$E/chain_x4_alone_nospin.json')); print(d['value'], {k:(v['img_per_s'],v['peak_mem_gb']) for k,v in d['per_scale'].items()})"
