"""Sums rocprofv3 counter_collection.csv files per kernel: prints and writes <dir>/summary.json
{pass: {kernel: {"dispatches": n, "<counter>": total}}}."""
import csv, glob, json, os, re, sys, collections

root = sys.argv[1]
summary = {}
for sub in sorted(os.listdir(root)):
	files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
	if not files:
		continue
	acc = collections.defaultdict(lambda: collections.defaultdict(float))
	seen = collections.defaultdict(set)
	for f in files:
		for r in csv.DictReader(open(f)):
			k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
			k = re.sub(r"\((pz_|float|unsigned|int|HIP|EltArgs|IgemmArgs|WgradArgs|PackArgs|WinoArgs|WinoWgradArgs|WinoFilterArgs|BnGeom|PoolGeom).*", "", k).strip()
			acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
			seen[k].add(r["Dispatch_Id"])
	summary[sub] = {k: dict(v, dispatches=len(seen[k])) for k, v in acc.items()}
	print("==", sub)
	for k, v in sorted(summary[sub].items(), key=lambda kv: -max(x for n, x in kv[1].items() if n != "dispatches"))[:14]:
		print("   %-70s %s" % (k[:70], {n: int(x) for n, x in v.items()}))
json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1)
