# usage (GPU box): bash tools/ab_env.sh VAR=VALUE [bench args...]   -- A/B of one environment knob (off = the knob set, on = unset)
KV=$1; shift
for v in off on off on; do
  unset ${KV%%=*}; [ $v = off ] && export $KV
  python bench.py --steps 80 --warmup 10 --no-kernel-timing --no-cpu-baseline --no-extra "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
