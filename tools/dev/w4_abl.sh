#!/bin/bash
# development aid: builds ablated variants of wino4.hip (W4_ABL bits) into gpurun_out-independent scratch libraries
# usage (here): bash tools/dev/w4_abl.sh build "0 1 2 4 8 3";  (GPU): bash tools/dev/w4_abl.sh run "0 1 2 4 8 3"
cd "$(dirname "$0")/../.."
mode=$1; shift
vars=${1:-"0 1 2 4 8"}
if [ "$mode" = build ]; then
	for v in $vars; do tools/variant_build.sh w4abl_$v --rig -DW4_ABL=$v || exit 1; done
else
	for v in $vars; do
		echo "== W4_ABL=$v"
		PUZZLE_MI355_LIB=$PWD/puzzlelib_amd/variants/lib_w4abl_$v.so timeout 300 python tools/dev/w4_time.py 2>&1 | grep -v Warn
	done
fi
