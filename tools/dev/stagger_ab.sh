#!/bin/bash
# development aid: staggered start of the implicit GEMM's first round (PUZZLE_MI355_IG_STAGGER = sleep quanta per slot) on the
# short-reduction 1x1 layers, steady state (300 launches per pass)
cd "$(dirname "$0")/../.."
for st in "$@"; do
	echo "== PUZZLE_MI355_IG_STAGGER=$st"
	for i in 1 3 4 7 9 12; do
		PUZZLE_MI355_IG_STAGGER=$st PUZZLE_MI355_IG_STAGGER_K=${STK:-256} python tools/conv_census.py --reps 300 --passes fwd,dgrad --only $i 2>&1 | grep 1x1 | cut -c1-78
	done
done
