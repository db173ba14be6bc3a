cd /root/repo
echo "== determinism split6 B=256"; PUZZLE_MI355_MATH=split6 python tools/step_determinism.py 2>&1 | grep -v "^\[Puzzle" | grep -v differing | head -3
echo "== determinism f32 B=256"; python tools/step_determinism.py 2>&1 | grep -v "^\[Puzzle" | grep -v differing | head -3
echo "== gpu tests split6"; PUZZLE_MI355_MATH=split6 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== gpu tests f32"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
PUZZLE_MI355_MATH=split6 python bench.py --no-cpu-baseline --no-extras 2>/dev/null > gpurun_out/bench_split6.json
python bench.py --no-cpu-baseline --no-extras 2>/dev/null > gpurun_out/bench_f32.json
