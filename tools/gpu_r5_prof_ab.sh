# Round 5: per-kernel table of the training step with the RAB's 256-channel tensors as padded planes (SRHIP_PP_RAB=1) and as fp32 (=0)
cd /tmp && export TMPDIR=/tmp
for pp in 1 0; do
  SRHIP_PP_RAB=$pp rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_step_pp$pp -o st -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32-line --no-sustained --spinup-steps 0 --steps 6 --warmup 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for pp in 1 0; do echo "== SRHIP_PP_RAB=$pp"; f=$(find gpurun_out/prof_step_pp$pp -name "*.db" | head -1); python tools/rocpd_stats.py $f 28; done
