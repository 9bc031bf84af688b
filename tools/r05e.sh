cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
timeout 900 python -m pytest tests/test_hip_norm.py -x -q > $O/test_norm.log 2>&1; echo "tests rc $?" >> $O/test_norm.log
tail -n 25 $O/test_norm.log
timeout 300 python tools/bn_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bn_bench.txt
