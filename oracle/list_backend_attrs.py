"""
TEST INFRASTRUCTURE (build container only) — lists every attribute PuzzleLib's dispatch surface reads from the backend
object (`backend.<name>` in Backend/*.py and Backend/Kernels/*.py, plus the methods it calls on `.blas`, `.dnn`,
`.matmod`, `.costmod`, `.memoryPool`, `.GPUArray`) and writes them to tests/golden/backend_attrs.json. The GPU test
tests/test_gpu_2_boundary.py asserts that the MI355X backend object offers each of them.
"""
import json, os, re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "backend_attrs.json")

files = [os.path.join(REF, "Backend", f) for f in os.listdir(os.path.join(REF, "Backend")) if f.endswith(".py")]
files += [os.path.join(REF, "Backend", "Kernels", f) for f in os.listdir(os.path.join(REF, "Backend", "Kernels"))
		  if f.endswith(".py")]

attrs, sub = set(), {"blas": set(), "dnn": set(), "matmod": set(), "costmod": set(), "memoryPool": set(), "GPUArray": set()}

for path in files:
	text = open(path).read()
	# only the GPU (initCuda/initHip/initGPU) code paths talk to a backend object
	for name in re.findall(r"\bbackend\.([A-Za-z_]\w*)", text):
		attrs.add(name)
	for obj in sub:
		for name in re.findall(r"\b%s\.([A-Za-z_]\w*)" % obj, text):
			sub[obj].add(name)

# names that only the CUDA branch (initCuda) uses
cuda_only = {"mapLRN", "mapLRNBackward", "crossMapLRN", "crossMapLRNBackward", "spatialTf", "spatialTfBackward"}
sub["dnn"] -= cuda_only

json.dump({"backend": sorted(attrs), **{k: sorted(v) for k, v in sub.items()}}, open(OUT, "w"), indent=1)
print(OUT, len(attrs), {k: len(v) for k, v in sub.items()})
