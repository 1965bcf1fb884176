# Per-kernel table of the step with all work on ONE stream: every duration is the kernel's own (no other stream shares the CUs).
cd /tmp && export TMPDIR=/tmp
export SRHIP_OVERLAP_WGRAD=0 SRHIP_OVERLAP_D=0
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_serial -o st -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-line --no-sustained --spinup-steps 0 --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_serial -name "*.db" | head -1); python tools/rocpd_stats.py $f 70 | tee gpurun_out/prof_serial/table.txt
