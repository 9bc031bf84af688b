cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_bn_sync.py -q -s > $O/pytest.txt 2>&1
echo "rc $?" >> $O/pytest.txt
timeout 600 python -m pytest "tests/test_hip_dp.py" -q > $O/pytest_dp.txt 2>&1
grep -E "passed|failed|Error|assert " $O/pytest.txt $O/pytest_dp.txt | head -40
