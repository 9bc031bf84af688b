cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03r
python -m pytest tests/test_hip_linear1.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py -x -q > gpurun_out/r03r/pytest.txt 2>&1
tail -8 gpurun_out/r03r/pytest.txt
for w in code2 molpcba; do
  python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03r/bench_${w}_w1.json 2>/dev/null
  GT_BF16_GEMM=tiled python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > gpurun_out/r03r/bench_${w}_tiled.json 2>/dev/null
done
python bench.py --workload er --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r03r/bench_er_w1.json 2>/dev/null
GT_BF16_GEMM=tiled python bench.py --workload er --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-kernel-timing > gpurun_out/r03r/bench_er_tiled.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03r/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
