#!/bin/bash
# Everything the round's profiles/ entries are made of, in one GPU call (≈6 minutes): the full -m gpu suite, the rocprofv3
# kernel statistics of the bench command, the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes, calibrated),
# the bench line itself, the two censuses and the NiN step timeline. Copy what is to be judged into profiles/ afterwards
# (tools/collect_profiles.sh <round tag>).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "from puzzlelib_amd import lib; print('build', lib.buildId())" > gpurun_out/gpu_tests.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q >> gpurun_out/gpu_tests.txt 2>&1; echo "pytest exit $?" >> gpurun_out/gpu_tests.txt
bash tools/prof_bench.sh one_stream > gpurun_out/summary_one_stream.txt 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
# steady state: 300 launches per pass (10-launch bursts after a synchronisation read 13-20 % low)
python tools/conv_census.py --reps 300 > gpurun_out/census_resnet50.txt 2>&1
python tools/conv_census.py --nin --reps 300 > gpurun_out/census_nin.txt 2>&1
bash tools/step_vs_steady.sh > /dev/null 2>&1
bash tools/nin_trace.sh > gpurun_out/nin_step_trace.txt 2>&1
tail -3 gpurun_out/gpu_tests.txt; tail -c 1500 gpurun_out/bench_final.json
