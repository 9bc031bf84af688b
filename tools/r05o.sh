cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
timeout 2400 python -m pytest tests/test_hip_configs.py -q -s -k "precision or bench_precision" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
grep -n "logits:\|bench sample\|closest to their bound\|passed\|failed\|rc" $O/tests.log | cut -c1-400
