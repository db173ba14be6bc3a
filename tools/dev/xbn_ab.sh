python -m pytest tests/test_gpu_3_fusion.py tests/test_gpu_4_fullsize.py tests/test_gpu_5_nets.py tests/test_gpu_6_fulltensor.py -x -q > gpurun_out/r05_xbn_tests.txt 2>&1; tail -5 gpurun_out/r05_xbn_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
PUZZLE_MI355_LAZY_OFF=xbn python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xbn off', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xbn on ', d['ms_per_step'], d['roofline']['frac'], d.get('fusion_counts',{}).get('conv_xbn'))"
done
