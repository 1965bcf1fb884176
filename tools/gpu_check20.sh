echo "=== gpu tests, bf16x3 ==="
SRADSGAN_CONV_MATH=bf16x3 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "=== bench bf16x3 ==="
SRADSGAN_CONV_MATH=bf16x3 python bench.py --no-cpu-baseline 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
SRADSGAN_CONV_MATH=bf16x3 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o x3 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && ls gpurun_out/prof_x3 | head; f=$(find gpurun_out/prof_x3 -name "*.db" | head -1); python tools/rocpd_stats.py $f 2>&1 | head -48
