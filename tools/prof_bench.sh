#!/bin/bash
# rocprofv3 kernel statistics of the bench command (side measurements off): gpurun_out/prof_<tag>/ and a per-step summary
# CSV under gpurun_out/<tag>_kernel_stats.csv (copy it to profiles/ to have it judged).
#   tools/prof_bench.sh <tag> [extra env assignments, e.g. PUZZLE_MI355_LAZY_OFF=sidestream]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=${1:-prof}
shift || true
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err || true
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$STATS" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
python - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 16.0      # 3 warm-up + 10 timed + 3 roofline steps... approximate: normalise by launches of Adam below
adam = [r for r in rows if "OpAdam" in r["Name"]]
steps = float(adam[0]["Calls"]) if adam else steps
tot = 0.0
print("%-86s %7s %9s %9s" % ("kernel", "calls/s", "avg us", "ms/step"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
	ms = float(r["TotalDurationNs"]) / 1e6 / steps
	tot += ms
	name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:86]
	print("%-86s %7.1f %9.1f %9.3f" % (name, float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms))
print("sum of all kernels per step: %.2f ms over %d steps" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps, steps))
PY
