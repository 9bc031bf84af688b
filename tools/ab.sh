# usage (GPU box): bash tools/ab.sh [bench args...]   -- A/B of graphtrans_amd/libgt_old.so (another build) vs the current library,
# alternating runs on the same box (clean lines: no kernel brackets, no CPU leg)
for v in old new old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  python bench.py --steps 100 --warmup 10 --no-kernel-timing --no-cpu-baseline --no-extra --report /tmp/ab_report.json "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step_idle_device'))"
done
