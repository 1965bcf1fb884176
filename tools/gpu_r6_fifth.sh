#!/bin/bash
# experiment (third lever): the RAB weight gradients on the 4-wave flat kernel at <= 1.5 blocks per CU, so that the
# main stream's conv kernels keep two blocks per CU beside it instead of one
R=$GRAFT_REPO_ROOT; E=$R/gpurun_out/r6e; mkdir -p $E; cd $R
B="python bench.py --steps 30 --warmup 5 --step-only"
V="SRHIP_X_PP=0 SRHIP_DU_PP=0 SRHIP_WGRAD_PP_CONVERT=0 SRHIP_FLAT_F32_K8=0"
for i in 1 2; do
  timeout 300 $B 2>&1 | tail -1 | cut -c1-150
  for fb in 192 384 768; do
    echo "4-wave flat kernel, block target $fb:"; env $V SRHIP_FLAT_BLOCKS=$fb timeout 300 $B 2>&1 | tail -1 | cut -c1-150
  done
  echo "8-wave kernel with in-kernel split (x, du fp32):"; env SRHIP_X_PP=0 SRHIP_DU_PP=0 SRHIP_WGRAD_PP_CONVERT=0 timeout 300 $B 2>&1 | tail -1 | cut -c1-150
done
env $V SRHIP_FLAT_BLOCKS=384 timeout 300 python tools/step_timeline.py 2>&1 | tail -12
