cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.txt 2>&1; echo "pytest exit $?" >> gpurun_out/gpu_tests.txt
bash tools/prof_bench.sh two_streams > gpurun_out/summary_two_streams.txt 2>&1
bash tools/prof_bench.sh one_stream PUZZLE_MI355_LAZY_OFF=sidestream > gpurun_out/summary_one_stream.txt 2>&1
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.txt 2>&1
PUZZLE_MI355_MATH=split6 bash tools/prof_bench.sh split6 > gpurun_out/summary_split6.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -3 gpurun_out/gpu_tests.txt; cat gpurun_out/bench_final.json
