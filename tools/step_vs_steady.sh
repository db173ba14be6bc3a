#!/bin/bash
# The same launches inside the training step and alone in steady state: rocprofv3 kernel traces of (a) bench.py and (b) the census's
# pointwise layers with --reps 300, joined on (kernel name, grid size). Output: gpurun_out/step_vs_steady.txt
#   tools/step_vs_steady.sh
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/svs
rm -rf $OUT && mkdir -p $OUT/a $OUT/b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/a -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $OUT/a.log 2>&1
for i in 1 3 4 7 9 12 14 17 19; do
	rocprofv3 --kernel-trace --output-format csv -d $OUT/b/$i -- python $ROOT/tools/conv_census.py --reps 300 --passes fwd,dgrad,wgrad --only $i > $OUT/b/$i.log 2>&1
done
python - $OUT <<'PY' > $ROOT/gpurun_out/step_vs_steady.txt
import csv, glob, re, sys, collections
out = sys.argv[1]
def load(pattern):
	acc = collections.defaultdict(list)
	for f in glob.glob(pattern, recursive=True):
		for r in csv.DictReader(open(f)):
			name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
			if "igemm_conv_kernel" not in name and "wgrad_conv_kernel" not in name: continue
			grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
			acc[(name[:70], grid, int(r.get("Grid_Size_Z", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
	return acc
a = load(out + "/a/**/*kernel_trace.csv")
b = load(out + "/b/**/*kernel_trace.csv")
print("%-72s %9s | %6s %9s | %6s %9s | %6s" % ("kernel", "grid", "n step", "us step", "n alone", "us alone", "ratio"))
ta = tb = 0.0
for key in sorted(a, key=lambda k: -sum(a[k])):
	va = a[key]
	# drop the warm-up third
	va = va[len(va) // 3:]
	ma = sum(va) / len(va)
	if key in b:
		vb = b[key][len(b[key]) // 3:]
		mb = sum(vb) / len(vb)
		ta += ma * len(va); tb += mb * len(va)
		print("%-72s %9d | %6d %9.1f | %6d %9.1f | %6.3f" % (key[0], key[1], len(va), ma, len(vb), mb, ma / mb))
	else:
		print("%-72s %9d | %6d %9.1f | %6s %9s |" % (key[0], key[1], len(va), ma, "-", "-"))
if tb: print("matched launches: in the step %.1f ms, alone %.1f ms, ratio %.3f" % (ta / 1e3, tb / 1e3, ta / tb))
PY
find $OUT -name "*.csv" -delete
