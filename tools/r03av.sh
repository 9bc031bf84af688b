cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03av; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_norm.py tests/test_hip_engine.py tests/test_hip_parity.py -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
GT_CHECK_NOSYNC=1 GT_CHECK_ITERS=3000 GT_CHECK_MODE=mixed timeout 900 python tools/engine_check_full.py 2>&1 | grep "module path"
