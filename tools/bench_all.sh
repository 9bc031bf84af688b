# usage (GPU box): bash tools/bench_all.sh <tag>  -- clean bench lines (no kernel timing, no CPU baseline) of the five workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r04g}; mkdir -p $O
for w in code2 molpcba nci1 code2-pna er; do
  S=100; [ $w = er ] && S=20
  python bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/clean_$w.json 2> $O/clean_$w.err
  python -c "
import json,sys
d=json.loads(open('$O/clean_$w.json').read().strip().splitlines()[-1])
print('$w', d['value'], 'graphs/s', d['ms_per_step'], 'ms/step host', d['host_enqueue_ms_per_step'])" || tail -3 $O/clean_$w.err
done
