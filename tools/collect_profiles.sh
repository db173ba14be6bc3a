#!/bin/bash
# gpurun_out/ (scratch) -> profiles/<tag>_* (tracked): tools/collect_profiles.sh r03
cd "$(dirname "$0")/.."
T=${1:-r03}
cp gpurun_out/gpu_tests.txt profiles/${T}_gpu_tests.txt
cp gpurun_out/bench_final.json profiles/${T}_bench_resnet50_b256.json
cp gpurun_out/one_stream_kernel_stats.csv profiles/${T}_bench_resnet50_b256_kernel_stats.csv
cp gpurun_out/summary_one_stream.txt profiles/${T}_kernel_summary.txt
cp gpurun_out/hbm_traffic.json profiles/${T}_hbm_traffic.json
cp gpurun_out/pmc_bench/summary.json profiles/${T}_pmc_fetch_write_summary.json
cp gpurun_out/census_resnet50.txt profiles/${T}_conv_census_resnet50_b256.txt
cp gpurun_out/census_nin.txt profiles/${T}_conv_census_nin_b128.txt
cp gpurun_out/nin_step_trace.txt profiles/${T}_nin_step_trace.txt
ls -la profiles | grep ${T}_
