cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
timeout 600 python -m pytest tests/test_hip_linear3r.py -x -q > $O/test_lin3r.log 2>&1; echo "tests rc $?" >> $O/test_lin3r.log
tail -n 4 $O/test_lin3r.log
for m in 0 1 2 3 8; do ./tools/gemm3_probe_r$m; done 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
timeout 300 python tools/gemm3r_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm3r_bench.txt
