# usage (GPU box): [AB_LIBS="name=path ..."] bash tools/ab.sh [bench args...]   -- alternating clean bench runs of several builds of the
# library on the same box (default: old = graphtrans_amd/libgt_old.so against new = the in-tree library)
LIBS=${AB_LIBS:-"old=$PWD/graphtrans_amd/libgt_old.so new="}
for round in 1 2 3; do
  for kv in $LIBS; do
    v=${kv%%=*}; p=${kv#*=}
    unset GT_LIB_PATH; [ -n "$p" ] && export GT_LIB_PATH=$p
    python bench.py --steps 100 --warmup 10 --no-kernel-timing --no-cpu-baseline --no-extra --report /tmp/ab_report.json "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step_idle_device'))"
  done
done
