#!/bin/bash
# Collects SQ / TCC counters for one census layer (tools/conv_census.py --only N --passes P), one rocprofv3 pass per
# counter group, into gpurun_out/pmc_<tag>/; prints per-kernel sums. Usage: tools/pmc_layer.sh <layer> <pass> <tag>
# PMC_CMD="python tools/wino_check.py --only 5 --reps 2" replaces the profiled command; PMC_GROUPS="1 2" picks counter groups.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
LAYER=${1:-6}; PASS=${2:-fwd}; TAG=${3:-l${LAYER}_${PASS}}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_MFMA"
G3="GRBM_GUI_ACTIVE FETCH_SIZE"
G4="WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
CMD=${PMC_CMD:-python $ROOT/tools/conv_census.py --only $LAYER --passes $PASS --reps 2}
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
	i=$((i+1))
	case " ${PMC_GROUPS:-1 2 3 4} " in *" $i "*) ;; *) continue ;; esac
	(cd $ROOT && rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- $CMD > $OUT/g$i.log 2>&1) || true
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print("==", k)
    for c in sorted(d): print("   %-32s %18.0f  (%d dispatches)" % (c, d[c], cnt[(k, c)]))
PY
