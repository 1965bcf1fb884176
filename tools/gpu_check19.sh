python tools/sweep_math.py 2>&1 | grep -v Warn | tail -40
echo "=== gpu tests, bf16x3 ==="
SRADSGAN_CONV_MATH=bf16x3 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== bench bf16x3 ==="
SRADSGAN_CONV_MATH=bf16x3 python bench.py --no-cpu-baseline 2>&1 | tail -2
