cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03x
python -m pytest tests/test_hip_linear1.py tests/test_hip_linear.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py -x -q > gpurun_out/r03x/pytest.txt 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r03x/pytest.txt | tail -5
for rep in 1 2; do
for v in 1 0; do
  GT_W1_DW=$v python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03x/bench_code2_dw${v}_$rep.json 2>/dev/null
done
done
python bench.py --workload code2 --no-cpu-baseline --no-extra --steps 64 > gpurun_out/r03x/bench_code2_kt.json 2>/dev/null
python bench.py --workload molpcba --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03x/bench_molpcba.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03x/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'))
        if 'kernels' in d:
            for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1].get('total_ms',0)):
                print(f"   {k:34s} calls={v.get('calls'):4} avg_us={v.get('avg_us'):8.2f} total_ms={v.get('total_ms'):7.3f} frac={v.get('frac')}")
    except Exception as e: print(f, 'ERR', e)
PY
