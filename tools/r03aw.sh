cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03aw; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_norm.py tests/test_hip_bn_sync.py tests/test_hip_engine.py tests/test_hip_parity.py -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
run() { tag=$1; w=$2; shift; shift; timeout 300 env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2 3; do
run fin code2 A=1
run old code2 GT_BN_FIN=0
done
for rep in 1 2; do
run fin molpcba A=1
run old molpcba GT_BN_FIN=0
done
