#!/bin/bash
# Round 6, first GPU call after a stretch without a device: what was written blind is checked first and on its own (fast, fails
# early), then the A/B of the opt-in backward-statistics epilogue, then everything the round's profiles/ entries are made of
# (tools/refresh_profiles.sh). Outputs under gpurun_out/ (scratch): copy what is to be judged with tools/collect_profiles.sh r06.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "from puzzlelib_amd import lib; print('build', lib.buildId())" > gpurun_out/r06_new_tests.txt 2>&1
# 1. the kernels and entries nobody has run on a device yet
PUZZLE_MI355_UNVERIFIED=1 timeout 1200 python -m pytest tests/test_gpu_6_fulltensor.py tests/test_gpu_2_boundary.py tests/test_gpu_8_multigpu.py tests/test_gpu_0_ops.py \
	-q -k "statistics or fp16 or out_of_scope or rehearsal or runGrid or (fresh and (r3ds1 or r33))" >> gpurun_out/r06_new_tests.txt 2>&1
echo "new tests exit $?" >> gpurun_out/r06_new_tests.txt
# 2. the new tapes (boundary tests, handlers, nets)
timeout 1500 python -m pytest tests/test_gpu_7_reftests.py -q -k "Boundary or Hip or Handlers or Nets or Sequential or Embedder or CTC or Cast or Module or Pad2D or Slice or ConvertToGraph" \
	> gpurun_out/r06_new_tapes.txt 2>&1
echo "new tapes exit $?" >> gpurun_out/r06_new_tapes.txt
# 3. A/B: BatchNorm-backward statistics from the producing backward-data epilogue (opt-in), two runs each, interleaved
for i in 1 2; do
	python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | tail -1 > gpurun_out/r06_ab_default_$i.json
	PUZZLE_MI355_DGRAD_STATS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | tail -1 > gpurun_out/r06_ab_dgradstats_$i.json
done
# 4. F(2x2, 5x5) on the Winograd kernel (opt-in): parity, then config 3 with and without it
PUZZLE_MI355_WINO5=1 timeout 600 python -m pytest tests/test_gpu_0_ops.py tests/test_gpu_5_nets.py -q -k "5x5 or nin" > gpurun_out/r06_wino5_tests.txt 2>&1
echo "wino5 tests exit $?" >> gpurun_out/r06_wino5_tests.txt
for i in 1 2; do
	python tools/nin_step.py 300 2> /dev/null | tail -1 > gpurun_out/r06_nin_default_$i.txt
	PUZZLE_MI355_WINO5=1 python tools/nin_step.py 300 2> /dev/null | tail -1 > gpurun_out/r06_nin_wino5_$i.txt
done
tail -2 gpurun_out/r06_wino5_tests.txt; tail -n 1 gpurun_out/r06_nin_default_*.txt gpurun_out/r06_nin_wino5_*.txt
python - <<'PY' > gpurun_out/r06_dgradstats_step_ab.txt 2>&1
import json
for tag in ("default", "dgradstats"):
	rows = [json.load(open("gpurun_out/r06_ab_%s_%d.json" % (tag, i))) for i in (1, 2)]
	print("%-11s ms/step %s  img/s %s  bn_bwd_gate %s  dgrad_bnstats %s" % (
		tag, ["%.2f" % r["ms_per_step"] for r in rows], ["%.0f" % r["value"] for r in rows],
		rows[0]["backend_fusion_counts_total"].get("bn_bwd_gate"), rows[0]["backend_fusion_counts_total"].get("dgrad_bnstats")))
PY
cat gpurun_out/r06_dgradstats_step_ab.txt
tail -4 gpurun_out/r06_new_tests.txt; tail -3 gpurun_out/r06_new_tapes.txt
