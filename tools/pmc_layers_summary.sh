#!/bin/bash
# MFMA-pipe utilisation and shader clock of representative ResNet-50 convolution launches (batch 256) from the SQ / GRBM
# counters (tools/pmc_layer.sh, one rocprofv3 pass per counter group): prints, per main kernel, MFMA instructions per
# dispatch, matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x active cycles), VALU per MFMA.
# Usage: tools/pmc_layers_summary.sh > gpurun_out/pmc_layers.txt
cd "$(dirname "$0")/.."
for spec in "3 fwd" "7 fwd" "12 fwd" "14 fwd" "14 dgrad" "12 wgrad" "2 fwd" "6 fwd" "11 fwd" "16 fwd" "6 wgrad"; do
	set -- $spec
	PMC_GROUPS="1 2 3" bash tools/pmc_layer.sh $1 $2 s3_l$1_$2 > gpurun_out/pmc_s3_l$1_$2.txt 2>&1
	python - "$1" "$2" gpurun_out/pmc_s3_l$1_$2.txt <<'PY'
import sys, re
layer, which, path = sys.argv[1:4]
kern, cur = {}, None
for line in open(path):
    if line.startswith("== "):
        cur = line[3:].strip(); kern[cur] = {}
    elif cur and line.startswith("   "):
        p = line.split()
        kern[cur][p[0]] = (float(p[1]), int(p[2].strip("(")))
main = max((k for k in kern if "SQ_INSTS_MFMA" in kern[k] and kern[k]["SQ_INSTS_MFMA"][0] > 0), key=lambda k: kern[k]["SQ_INSTS_MFMA"][0], default=None)
if main is None:
    print("layer %s %s: no MFMA kernel found" % (layer, which)); sys.exit(0)
d = kern[main]
n = d["SQ_INSTS_MFMA"][1]
busy, wave_c = d["SQ_VALU_MFMA_BUSY_CYCLES"][0], d["SQ_BUSY_CYCLES"][0]
gui = d.get("GRBM_GUI_ACTIVE", (0, 1))[0] / max(d.get("GRBM_GUI_ACTIVE", (0, 1))[1], 1)
# SQ_VALU_MFMA_BUSY_CYCLES = 64 per v_mfma_f32_32x32x2_f32, summed over the chip's 1 024 SIMDs; GRBM_GUI_ACTIVE is summed
# over the 8 XCDs: busy fraction = busy per dispatch / (1 024 SIMDs x active cycles) — the convention of profiles/r01_pmc_*
frac = (busy / n) / (gui / 8.0 * 1024) if gui else float("nan")
print("census layer %2s %-5s %-58s dispatches %2d  MFMA/dispatch %11.0f  GUI_ACTIVE/dispatch %10.0f  MFMA-busy fraction %.3f  VALU insts/MFMA %.2f" % (
    layer, which, main[:58], n, d["SQ_INSTS_MFMA"][0] / n, gui, frac,
    (d.get("SQ_ACTIVE_INST_VALU", (0, 1))[0] / 4.0) / max(d["SQ_INSTS_MFMA"][0], 1)))
PY
done
