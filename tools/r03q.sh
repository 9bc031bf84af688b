cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03q
python -m pytest tests/test_hip_linear1.py -x -q -s > gpurun_out/r03q/pytest_lin1.txt 2>&1
tail -30 gpurun_out/r03q/pytest_lin1.txt
python tools/gemm1_bench.py 32000 > gpurun_out/r03q/gemm1_32k.txt 2>&1
GT_W1_MAX_NTW=4 python tools/gemm1_bench.py 32000 > gpurun_out/r03q/gemm1_32k_ntw4.txt 2>&1
python tools/gemm1_bench.py 131072 > gpurun_out/r03q/gemm1_131k.txt 2>&1
cat gpurun_out/r03q/gemm1_*.txt
