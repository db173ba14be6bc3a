#!/bin/bash
# Variant builds of csrc/wino.hip (WN_ABL ablation bits / tuning macros) into puzzlelib_amd/variants/ and, on a GPU box,
# their timings on the 3x3 census layers through tools/wino_check.py. Kernel-tuning aid only.
set -e
cd "$(dirname "$0")/.."
mkdir -p puzzlelib_amd/variants
build() {   # name, extra flags: a scratch copy of csrc/ with the measurement rig applied (tools/variant_build.sh)
	local name=$1; shift
	tools/variant_build.sh w_$name --rig "$@"
}
if [ "$1" = "build" ]; then
	shift
	if [ -n "$1" ]; then name=$1; shift; build $name "$@"; exit 0; fi
	build base
	build noload -DWN_ABL=1
	build nostore -DWN_ABL=3
	build nomfma -DWN_ABL=4
	build noepi -DWN_ABL=8
	build mfmaonly -DWN_ABL=11
else
	for v in ${VARIANTS:-base noload nostore nomfma noepi mfmaonly}; do
		for layer in ${LAYERS:-4 5 7}; do
			echo "== $v layer $layer: $(PUZZLE_MI355_LIB=$PWD/puzzlelib_amd/variants/lib_w_$v.so python tools/wino_check.py --only $layer | sed 's/.*| fwd/fwd/')"
		done
	done
fi
