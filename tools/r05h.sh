cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
timeout 900 python -m pytest tests/test_hip_linear3r.py tests/test_hip_linear3x.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -n 6 $O/tests.log
GT_LIN3R=1 timeout 300 python tools/gemm3r_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm3r_bench.txt
echo "--- GT_LIN3R_DW_PC=0 (one wave per SIMD)"
GT_LIN3R=1 GT_LIN3R_DW_PC=0 timeout 300 python tools/gemm3r_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm3r_bench_pc0.txt
echo "--- k_lin3 / k_lin3_dw"
GT_LIN3R=0 GT_LIN3R_DW=0 timeout 300 python tools/gemm3r_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm3_bench_old.txt
