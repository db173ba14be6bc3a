cd /root/repo
for t in 4 2; do echo "== tile $t"; timeout 600 python tools/wino_check.py --tile $t --reps 10 2>&1 | grep -v Warn; done
