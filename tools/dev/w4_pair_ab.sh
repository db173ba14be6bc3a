#!/bin/bash
# development aid: F(4x4) Winograd kernel with one / two channel blocks per workgroup on the four 3x3 layers, steady state
cd "$(dirname "$0")/../.."
for v in "$@"; do
	echo "== PUZZLE_MI355_WINO4_PAIR=$v"
	for i in 2 6 11 16; do
		PUZZLE_MI355_WINO4_PAIR=$v python tools/conv_census.py --reps 300 --passes fwd,dgrad --only $i 2>&1 | grep 3x3 | cut -c1-78
	done
done
