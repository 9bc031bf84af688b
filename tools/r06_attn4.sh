# attention: VGPR-form MFMA accumulators (launch bounds 2 waves per SIMD) on top of the prefetch fix; same-box A/B against the previous attention.hip
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06at; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_attention.py tests/test_hip_options.py -q -x > $O/pytest_attn4.txt 2>&1; tail -2 $O/pytest_attn4.txt
for v in old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  echo "== $v"; python tools/attn_bench.py 2>/dev/null | grep -E "kernels" | grep -E "longest first" | grep -E "Code2-like batch, dropout|ER|one of 1001"
done 2>&1 | tee $O/attn_ab4.txt
unset GT_LIB_PATH
bash tools/ab.sh 2>&1 | tee $O/ab4_code2.txt
bash tools/ab.sh --mode fp32 2>&1 | tee $O/ab4_code2_fp32.txt
bash tools/ab.sh --workload er --steps 20 2>&1 | tee $O/ab4_er.txt
