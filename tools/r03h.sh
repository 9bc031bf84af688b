cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
