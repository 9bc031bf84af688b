cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03v
python -m pytest tests/test_hip_linear1.py tests/test_hip_attention.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py -x -q > gpurun_out/r03v/pytest.txt 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r03v/pytest.txt | tail -5
for w in code2 molpcba; do
  python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03v/bench_${w}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03v/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
