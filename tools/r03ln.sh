cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03ln
python -m pytest tests/test_hip_norm.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for rep in 1 2; do for v in 1 0; do
GT_LN_BWD_D128=$v python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('d128=$v', d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'))"
done; done
