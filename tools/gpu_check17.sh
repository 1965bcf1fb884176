cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gp -o gp -- python3 $GRAFT_REPO_ROOT/tools/prof_gp.py > $GRAFT_REPO_ROOT/gpurun_out/prof_gp.log 2>&1
