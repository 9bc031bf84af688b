cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06i}; mkdir -p $O
export AB_LIBS="old=$PWD/graphtrans_amd/libgt_old.so fold=$PWD/graphtrans_amd/libgt_fold.so new="
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "segment or pool or vn or virtual" > $O/pytest_part.txt 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_part.txt
{ echo "== code2 b256"; bash tools/ab.sh; echo "== code2 b32"; bash tools/ab.sh --batch 32; } 2>&1 | tee $O/ab_fold_segsum.txt
