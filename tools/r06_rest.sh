# usage (GPU box): bash tools/r06_rest.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06f}; mkdir -p $O
python -m pytest tests/test_hip_linear_bn_slab.py tests/test_hip_options.py -m gpu -q > $O/pytest_slab.txt 2>&1; echo "slab rc $?"; tail -6 $O/pytest_slab.txt
python -m pytest tests -m gpu -q -s -k "not test_hip_linear_bn_slab and not test_hip_options" --deselect tests/test_hip_aggregate.py --deselect tests/test_hip_attention.py --deselect tests/test_hip_bn_sync.py > $O/pytest_gpu.txt 2>&1; echo "rest rc $?"; grep -E "passed|failed" $O/pytest_gpu.txt | tail -3; grep -E "^\[molpcba fp32\]|^\[code2 fp32\]|FAILED" $O/pytest_gpu.txt | head -20
