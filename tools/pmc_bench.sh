#!/bin/bash
# HBM traffic of the bench's kernels from the memory-side L2 counters (one rocprofv3 pass per counter, as
# MI355X_MICROARCH.md prescribes), plus the calibration of those counters on known byte counts.
# Writes gpurun_out/pmc_bench/{fetch,write,cal_fetch,cal_write}/ and prints per-kernel totals.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- python $ROOT/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -- python $ROOT/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/write.log 2>&1 || true
find $OUT -name "*kernel_trace.csv" -delete
python $ROOT/tools/pmc_summary.py $OUT
python $ROOT/tools/pmc_traffic.py $OUT $ROOT/gpurun_out/hbm_traffic.json
