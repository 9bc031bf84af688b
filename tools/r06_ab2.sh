# usage (GPU box): bash tools/r06_ab2.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06h}; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_hip_options.py tests/test_hip_engine.py -m gpu -q -x > $O/pytest_part.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_part.txt
{ echo "== code2 b256"; bash tools/ab.sh; echo "== code2 b32"; bash tools/ab.sh --batch 32; echo "== molpcba"; bash tools/ab.sh --workload molpcba; } 2>&1 | tee $O/ab_segsum.txt
