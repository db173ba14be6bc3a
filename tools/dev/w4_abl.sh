#!/bin/bash
# development aid: builds ablated variants of wino4.hip (W4_ABL bits) into gpurun_out-independent scratch libraries
# usage (here): bash tools/dev/w4_abl.sh build "0 1 2 4 8 3";  (GPU): bash tools/dev/w4_abl.sh run "0 1 2 4 8 3"
cd "$(dirname "$0")/../.."
mode=$1; shift
vars=${1:-"0 1 2 4 8"}
if [ "$mode" = build ]; then
	mkdir -p tools/dev/abl
	for v in $vars; do
		hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DW4_ABL=$v -c puzzlelib_amd/csrc/wino4.hip -o tools/dev/abl/wino4_$v.o || exit 1
		objs=$(ls puzzlelib_amd/csrc/build/*.o | grep -v wino4.o)
		hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/abl/lib_$v.so $objs tools/dev/abl/wino4_$v.o -ldl || exit 1
	done
else
	for v in $vars; do
		echo "== W4_ABL=$v"
		PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so timeout 300 python tools/dev/w4_time.py 2>&1 | grep -v Warn
	done
fi
