#!/bin/bash
# Builds variant libraries of the convolution kernels (ablation macros PZ_ABL / PZ_LB in csrc/conv.hip) into
# puzzlelib_amd/variants/ and, on a GPU box, times them on selected census layers. Kernel-tuning aid only.
set -e
cd "$(dirname "$0")/.."
mkdir -p puzzlelib_amd/variants
build() {   # name, extra flags: a scratch copy of csrc/ with the measurement rig applied (tools/variant_build.sh)
	local name=$1; shift
	tools/variant_build.sh $name --rig "$@"
}
PASS=${PASS:-fwd}
if [ "$1" = "build" ]; then
	[ -n "$ONLY" ] && { build $ONLY $FLAGS; exit 0; }
	build base
	build wg_nomfma -DPZ_ABL=512
	build wg_onestore -DPZ_ABL=1024
	build wg_nosplit -DPZ_ABL=2048
	build noload -DPZ_ABL=1
	build nosplit -DPZ_ABL=8
	build onestore -DPZ_ABL=16
	build nosplit_onestore -DPZ_ABL=24
	build nofilter -DPZ_ABL=32
	build noslp -fno-slp-vectorize
	ls -la puzzlelib_amd/variants/*.so
else
	for v in ${VARIANTS:-base noload nosplit onestore nosplit_onestore nofilter noslp}; do
		for layer in ${LAYERS:-3 12 14}; do
			echo "== $v layer $layer"
			PUZZLE_MI355_LIB=$PWD/puzzlelib_amd/variants/lib_$v.so python tools/conv_census.py --only $layer --passes $PASS --reps 5 | sed -n 2p
		done
	done
fi
