cd /root/repo
timeout 250 tools/probes/pk_forms_probe > gpurun_out/pk_forms_probe.txt 2>&1
timeout 200 tools/probes/bn_vs_mfma > gpurun_out/bn_vs_mfma.txt 2>&1
timeout 120 tools/probes/split_probe > gpurun_out/split_probe.txt 2>&1
for m in f32 split6 split9; do echo "== gpu tests $m"; PUZZLE_MI355_MATH=$m timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3; done
for m in split6 f32; do echo "== determinism $m"; PUZZLE_MI355_MATH=$m python tools/step_determinism.py 2>&1 | grep -v "^\[Puzzle" | grep -v differing | head -2; done
for m in f32 split6; do PUZZLE_MI355_MATH=$m python bench.py --no-cpu-baseline --no-extras 2>/dev/null > gpurun_out/bench_$m.json; done
