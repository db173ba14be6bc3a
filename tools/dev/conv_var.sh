#!/bin/bash
# development aid: variant builds of conv.hip with extra -D flags: bash tools/dev/conv_var.sh build name "-DFLAG"; run name... (times the
# 1x1 layers of the census: tools/conv_census.py --only N)
cd "$(dirname "$0")/../.."
mode=$1; shift
if [ "$mode" = build ]; then
	mkdir -p tools/dev/abl
	name=$1; shift
	hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c puzzlelib_amd/csrc/conv.hip -o tools/dev/abl/conv_$name.o || exit 1
	objs=$(ls puzzlelib_amd/csrc/build/*.o | grep -v "/conv.o")
	hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/abl/lib_$name.so $objs tools/dev/abl/conv_$name.o -ldl || exit 1
else
	for v in "$@"; do
		echo "== $v"
		PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so timeout 600 python tools/conv_census.py --reps 10 2>&1 | grep "1x1" | cut -c1-110
	done
fi
