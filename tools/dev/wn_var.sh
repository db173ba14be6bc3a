#!/bin/bash
# development aid: variant builds of wino.hip with extra -D flags: bash tools/dev/wn_var.sh build name "-DFLAG"; run name... (W4ARGS passed to w4_time.py)
cd "$(dirname "$0")/../.."
mode=$1; shift
if [ "$mode" = build ]; then
	mkdir -p tools/dev/abl
	name=$1; shift
	hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c puzzlelib_amd/csrc/wino.hip -o tools/dev/abl/wino_$name.o || exit 1
	objs=$(ls puzzlelib_amd/csrc/build/*.o | grep -v "/wino.o")
	hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/abl/lib_$name.so $objs tools/dev/abl/wino_$name.o -ldl || exit 1
else
	for v in "$@"; do
		echo "== $v"
		PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so timeout 300 python tools/dev/w4_time.py $W4ARGS 2>&1 | grep -v Warn
	done
fi
