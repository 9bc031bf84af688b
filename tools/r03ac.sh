cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ac; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_linear1.py -x -q -k "weight_gradient" > $O/pytest_dw.txt 2>&1; tail -5 $O/pytest_dw.txt
for b in 128 192 256 384; do echo "== GT_DW16_BLOCKS=$b"; GT_DW16_BLOCKS=$b timeout 300 python tools/dw16_bench.py; done 2>&1 | grep -v amdgpu.ids | tee $O/dw16_bench.txt
echo "== old kernel"; GT_DW16=0 timeout 300 python tools/dw16_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/dw16_bench.txt
