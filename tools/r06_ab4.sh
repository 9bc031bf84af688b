cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06j}; mkdir -p $O
python -m pytest tests/test_hip_norm.py tests/test_hip_layers.py tests/test_hip_fp32_accuracy.py -m gpu -q -x > $O/pytest_part.txt 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_part.txt
export AB_LIBS="fold=$PWD/graphtrans_amd/libgt_fold.so new="
{ echo "== code2 b256 fp32"; bash tools/ab.sh --mode fp32; } 2>&1 | tee $O/ab_ln_bwd_fp32.txt
