cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03y
python -m pytest tests/test_hip_linear1.py -x -q -k weight_gradient 2>&1 | tail -2
for v in 0 128 256 512 1024; do
  if [ $v = 0 ]; then export GT_W1_DW=0; else export GT_W1_DW=1 GT_W1_DW_BLOCKS=$v; fi
  python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03y/bench_code2_dwb${v}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03y/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
