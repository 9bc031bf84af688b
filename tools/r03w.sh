cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03w
for rep in 1 2; do
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  GT_W1_LN=$1 GT_W1_LNB=$2 python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03w/bench_code2_ln$1_lnb$2_$rep.json 2>/dev/null
done
done
python bench.py --workload code2 --no-cpu-baseline --no-extra --steps 64 > gpurun_out/r03w/bench_code2_kt.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03w/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'))
        if 'kernels' in d:
            for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1].get('total_ms',0)):
                print(f"   {k:34s} calls={v.get('calls'):4} avg_us={v.get('avg_us'):8.2f} total_ms={v.get('total_ms'):7.3f} frac={v.get('frac')}")
    except Exception as e: print(f, 'ERR', e)
PY
