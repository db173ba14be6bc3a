cd /root/repo
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f img/s  issue %.2f' % (d['ms_per_step'], d['value'], d['timed_region_host']['issue_ms_per_step']))"; }
for rep in 1 2; do
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_SDMA=0
done
echo "== NiN"; python tools/nin_step.py 300 2>/dev/null | tail -1; HIP_FORCE_DEV_KERNARG=1 python tools/nin_step.py 300 2>/dev/null | tail -1
