cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ai; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_norm.py tests/test_hip_bn_sync.py -x -q > $O/pytest_norm.txt 2>&1; tail -3 $O/pytest_norm.txt
timeout 1200 python -m pytest tests/test_hip_engine.py -x -q > $O/pytest_engine.txt 2>&1; tail -3 $O/pytest_engine.txt
run() { tag=$1; w=$2; shift; shift; env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2; do
run off code2 GT_BN_ONE=0
run on128 code2 GT_BN_ONE=1
run on64 code2 GT_BN_ONE_PART=64
run on256 code2 GT_BN_ONE_PART=256
done
for rep in 1 2; do
run off molpcba GT_BN_ONE=0
run on128 molpcba GT_BN_ONE=1
done
