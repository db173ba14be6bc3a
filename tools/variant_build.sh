#!/bin/bash
# Builds a VARIANT of the native library from a scratch copy of puzzlelib_amd/csrc — never from the shipped sources in place:
#     tools/variant_build.sh NAME [--rig] [--patch FILE]... [compiler flags, e.g. -DPZ_WG_RUNS=4]
# -> puzzlelib_amd/variants/lib_NAME.so     (use with PUZZLE_MI355_LIB=$PWD/puzzlelib_amd/variants/lib_NAME.so)
# --rig applies tools/dev/measurement_rig.patch first (the timing-only ablation switches PZ_ABL / WN_ABL / W4_ABL, PZ_IG_PRIO,
# PZ_EPI_AUX, W4_DUMMY_VALU ... that used to live in the kernel sources). The variant's pz_build_id() hashes its own sources
# and flags and pz_build_flags() carries the flags, so bench.py / smoke() refuse it unless PUZZLE_MI355_ALLOW_VARIANT=1.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
V=puzzlelib_amd/variants
D=$V/src_$name
rm -rf "$D"; mkdir -p "$D/csrc"
cp puzzlelib_amd/csrc/*.hip puzzlelib_amd/csrc/*.cpp puzzlelib_amd/csrc/*.h puzzlelib_amd/csrc/Makefile "$D/csrc/"
mkdir -p "$D/include" "$V/include"      # common.h includes ../../include/puzzle_mi355.h relative to csrc/
cp include/puzzle_mi355.h "$V/include/"
extra=()
while [ $# -gt 0 ]; do
	case "$1" in
		--rig) patch -s -p3 -d "$D/csrc" < tools/dev/measurement_rig.patch; shift;;
		--patch) patch -s -p3 -d "$D/csrc" < "$2"; shift 2;;
		*) extra+=("$1"); shift;;
	esac
done
make -s -C "$D/csrc" -j8 EXTRA="${extra[*]}" OUT="../../lib_$name.so" INC=../../include
ls -la "$V/lib_$name.so"
