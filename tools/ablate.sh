#!/bin/bash
# Builds variant libraries of the convolution kernels (ablation macros PZ_ABL / PZ_LB in csrc/conv.hip) into
# puzzlelib_amd/variants/ and, on a GPU box, times them on selected census layers. Kernel-tuning aid only.
set -e
cd "$(dirname "$0")/.."
mkdir -p puzzlelib_amd/variants
build() {   # name, extra flags
	local name=$1; shift
	hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c puzzlelib_amd/csrc/conv.hip -o puzzlelib_amd/variants/conv_$name.o
	hipcc --offload-arch=gfx950 -shared -fPIC -o puzzlelib_amd/variants/lib_$name.so puzzlelib_amd/variants/conv_$name.o \
		$(ls puzzlelib_amd/csrc/build/*.o | grep -v conv.o) -ldl
}
PASS=${PASS:-fwd}
if [ "$1" = "build" ]; then
	[ -n "$ONLY" ] && { build $ONLY $FLAGS; exit 0; }
	build base
	build noload -DPZ_ABL=9
	build noload_nobar -DPZ_ABL=27
	build loadonly -DPZ_ABL=32
	build storeonly -DPZ_ABL=64
	build nearloads -DPZ_ABL=128
	ls -la puzzlelib_amd/variants/*.so
else
	for v in ${VARIANTS:-base noload noload_nobar loadonly storeonly nearloads}; do
		for layer in 2 6 12; do
			echo "== $v layer $layer"
			PUZZLE_MI355_LIB=$PWD/puzzlelib_amd/variants/lib_$v.so python tools/conv_census.py --only $layer --passes $PASS --reps 5 | sed -n 2p
		done
	done
fi
