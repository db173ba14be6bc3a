#!/bin/bash
# development aid: A/B of the unpipelined implicit-GEMM form on the census's 1x1 layers (variant libraries from conv_var.sh)
cd "$(dirname "$0")/../.."
for spec in "$@"; do
	lib=${spec%%:*}; np=${spec##*:}
	echo "== lib $lib PUZZLE_MI355_IG_NP=$np"
	PUZZLE_MI355_IG_NP=$np PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$lib.so timeout 600 python tools/conv_census.py --reps 10 --passes fwd,dgrad 2>&1 | grep -E "1x1|per step" | cut -c1-100
done
