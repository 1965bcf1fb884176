cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -o st -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-line --no-sustained --spinup-steps 0 --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_step -name "*.db" | head -1); python tools/rocpd_stats.py $f 40
