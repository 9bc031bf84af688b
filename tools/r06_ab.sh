# usage (GPU box): bash tools/r06_ab.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06g}; mkdir -p $O
python -m pytest tests/test_hip_linear_bn_slab.py tests/test_hip_options.py -m gpu -q > $O/pytest_slab.txt 2>&1; echo "slab rc $?"; tail -4 $O/pytest_slab.txt
{ echo "== code2 b256"; bash tools/ab.sh; echo "== code2 b32"; bash tools/ab.sh --batch 32; echo "== molpcba"; bash tools/ab.sh --workload molpcba; } 2>&1 | tee $O/ab_vn_chain.txt
bash tools/timeline_round.sh $1 code2 mixed > $O/timeline_round.log 2>&1; tail -3 $O/timeline_round.log
grep -E "k_slab|k_segsum|k_small|k_bn_small|k_bcast|launches per step|kernel time per step" $O/${1}_code2_mixed_summary.txt
