cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03zz
python -m pytest tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py tests/test_hip_dp.py -x -q > gpurun_out/r03zz/pytest.txt 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r03zz/pytest.txt | tail -5
for rep in 1 2; do
for v in 1 0; do
  GT_PREP_OVERLAP=$v python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03zz/bench_code2_prep${v}_$rep.json 2>/dev/null
done
done
for v in 1 0; do
GT_PREP_OVERLAP=$v python bench.py --workload molpcba --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03zz/bench_molpcba_prep$v.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03zz/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
