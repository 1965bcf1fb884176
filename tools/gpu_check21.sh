python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --no-cpu-baseline 2>&1 | tail -1
python bench.py --no-cpu-baseline --conv-math fp32 2>&1 | tail -1
