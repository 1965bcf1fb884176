#!/bin/bash
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4j
mkdir -p $E
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_configs_gpu.py -x -q -m gpu -k "inference or tail or rab or resgroup or generator_small or psnr or evaluator" > $E/tests.log 2>&1; tail -5 $E/tests.log
SRHIP_TAIL_EVAL=0 timeout 600 python bench.py --workload infer 2>&1 | tail -1 > $E/infer_old.json; cut -c1-300 $E/infer_old.json
timeout 600 python bench.py --workload infer 2>&1 | tail -1 > $E/infer_new.json; cut -c1-300 $E/infer_new.json
SRHIP_TAIL_EVAL=0 timeout 600 python bench.py --workload infer 2>&1 | tail -1 > $E/infer_old2.json; cut -c1-120 $E/infer_old2.json
timeout 600 python bench.py --workload infer 2>&1 | tail -1 > $E/infer_new2.json; cut -c1-120 $E/infer_new2.json
