# k_lin1: prefetch loads pinned in front of the chunk's MFMAs (sched_barrier); A/B against the previous build (libgt_old.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06l1; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_linear1.py tests/test_hip_linear2.py -q -x > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for v in old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  echo "== $v"; python tools/gemm1_bench.py 32000 2>/dev/null | grep "M="
done 2>&1 | cut -c1-200 | tee $O/gemm1_ab.txt
unset GT_LIB_PATH
bash tools/ab.sh 2>&1 | tee $O/ab_code2.txt
bash tools/ab.sh --workload molpcba 2>&1 | tee $O/ab_molpcba.txt
