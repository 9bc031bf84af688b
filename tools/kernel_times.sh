# usage (GPU box): bash tools/kernel_times.sh <pattern> [bench args]  -- in-step HIP-event kernel times (bench.py's `kernels`) matching a regex
P=$1; shift
python bench.py --steps 48 --warmup 10 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python -c "
import sys, json, re
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'])
for k, v in sorted(d['kernels'].items()):
    if re.search(r'$P', k): print('  %-46s %8.2f us x %5.1f/step' % (k, v['avg_us'], v['calls'] / 2.0))
"
