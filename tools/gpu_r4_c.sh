#!/bin/bash
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/r4e
mkdir -p $E
cd $R
VARS=0,128,256,0,128,256 timeout 600 python tools/ablate_pers.py > $E/ablate.log 2>&1; cat $E/ablate.log
