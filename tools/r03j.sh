cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_dp.py tests/test_hip_norm.py tests/test_hip_engine.py -q -x > $O/pytest.txt 2>&1
echo "rc $?" >> $O/pytest.txt
tail -30 $O/pytest.txt
