# LayerNorm backward with two trips of bf16 rows in flight: parity + same-box A/B (libgt_old.so = the previous norm.hip)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06ln; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_norm.py -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
bash tools/ab.sh --workload er --steps 20 2>&1 | tee $O/ab_er.txt
bash tools/ab.sh 2>&1 | tee $O/ab_code2.txt
